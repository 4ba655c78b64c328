// Cross-attention PROBABILITIES over a short, per-prompt key set (the DiT's text cross-attention: <= 128 keys after the zero-padding
// rows are merged into one key) - the first half of the "cached V.Wo^T" form of diffusers==0.33.1 WanTransformerBlock.attn2 (call site
// /root/reference/inference_t23d.py:94-103):
//
//        attn2(x) = softmax(q K^T) V Wo^T + bo  =  sum_h  P_h (V_h Wo_h^T)  + bo
//
// K and V depend on the PROMPT only, so the host builds (V_h Wo_h^T) once per prompt (vist3a_amd/wan/dit.py::_context); per step this
// kernel writes the normalised probabilities P[m][h * Lkp + j] (bf16) and ONE GEMM with K = H * Lkp (1152 at Wan-1.3B instead of 1536)
// replaces the P.V MFMAs plus the to_out projection.
//
// One workgroup = 4 waves on 128 upw consecutive queries of one (batch item, head); a wave owns 32-query units.  The head's Lk <= 128 keys are
// copied to LDS once per workgroup (the flash kernel's image: 256-byte rows, 16-byte chunks XOR-swizzled by row), next to two small tables
// built under those loads: the per-row score factor (scale x log2 e x the query's RMS factor) and the per-key additive term (merged-padding
// bias x log2 e, -1e30 beyond the last key) - so the unit itself has no global load but its Q fragments.  S^T = K.Q^T on
// v_mfma_f32_32x32x16_bf16 over ceil(Lk / 32) sub-tiles with the flash kernel's permuted key -> row map, so that lane (query = l31, hi) ends
// up with 16 CONSECUTIVE keys per 32-key sub-tile: the softmax is a per-lane loop plus one cross-half exchange, and a lane stores its
// probabilities as 32 contiguous bytes.  (Round 4: 17.4 -> 14.0 us at the production shape against the first form - 64-key tiles, the row
// statistics and the bias loaded by every lane inside the softmax.)
// Rounding contract (oracle/wan_dit.py `ctx_vo`): scores fp32 from bf16 q, k (+ fp32 key bias), p = 2^(t - max t), t = s * scale * log2 e
// (* the query's RMS factor) + bias * log2 e, in fp32; l = fp32 sum in key order per lane then across the two lane halves,
// P = bf16(p * (1 / l)) (one IEEE division per query).
// q_row_sumsq: q arrives UN-normalised (the to_q projection's own bf16 output) together with the partial sums of squares of its rows that
// the projection's epilogue emitted (v3a_gemm_args.row_sumsq); the per-row factor rsqrt(mean(q^2) + eps) of diffusers' RMSNorm across
// heads multiplies the scores here, its per-column weight is folded into the cached keys by the host: the normalisation pass over q
// (one launch, 50 MB of traffic per block and step) disappears.
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

struct XaP {
  const char* q; const char* k; char* p;
  const float* kbias;
  long q_bs, k_bs, p_bs;   // elements per batch item
  int ldq, ldk, ldp;
  int H, Nq, Nk, Lkp, kbias_stride, kbias_first;
  float scale_log2e;
  const float* qsq;        // optional [B * Nq][qparts]: partial sums of squares of the (un-normalised) q rows
  int qparts;
  float q_eps, inv_dim;
  int upw;                 // 32-query units every wave walks (the workgroup covers 4 * upw * 32 consecutive queries of one head)
};

constexpr int D = 128, KROWB = D * 2, NW = 4;
constexpr float LOG2E = 1.4426950408889634f;

template <int NU>   // 32-key sub-tiles: Nk <= Lkp <= 32 NU
__global__ __launch_bounds__(256, 2) void xattn_probs_kernel(const XaP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [32 NU keys][256 B] | row factors [4 upw 32] | key table [32 NU]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int nsb = (p.Nq + 31) >> 5;
  const int SB = NW * p.upw;
  const int wpb = (nsb + SB - 1) / SB;                 // workgroups per (batch item, head)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);    // consecutive ids = the query ranges of one (batch, head): its keys stay in one L2
  const int bh = bid / wpb, sb0 = (bid % wpb) * SB;
  const int b = bh / p.H, h = bh % p.H;
  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;
  float* cs = (float*)(smem + NU * 32 * KROWB);
  float* bl = cs + SB * 32;

  // ---- keys -> LDS once per workgroup (rows clamped to the last key: the excess rows are masked through the key table) ----
#pragma unroll
  for (int j = 0; j < 2 * NU; ++j) {
    const int g = j * NW + wave;
    const int r = g * 4 + lane / 16, c = lane % 16;
    const int row = r < p.Nk ? r : p.Nk - 1;
    glds16(Kb + (size_t)row * p.ldk * 2 + (size_t)(c ^ (r & 15)) * 16, smem + g * 1024);
  }
  // ---- Q fragments (B operand) of a 32-query unit: lane (q = l31, hi) slot j <-> d = 16 ks + 8 hi + j ----
  auto load_q = [&](int sb, bf16x8 (&qf)[8]) {
    int qr = sb * 32 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  };
  bf16x8 qf[8];
  load_q(min(sb0 + wave, nsb - 1), qf);
  // ---- per-row score factor (softmax scale x log2 e x the query's RMS factor when q arrives un-normalised: parts added in index order)
  //      and the per-key additive table (merged-padding bias x log2 e; -1e30 beyond the last key), once per workgroup, under the loads ----
  for (int i = tid; i < SB * 32; i += 256) {
    float c = p.scale_log2e;
    if (p.qsq) {
      const float ss = sum_parts_in_order(p.qsq + ((size_t)b * p.Nq + min(sb0 * 32 + i, p.Nq - 1)) * p.qparts, p.qparts);
      c *= rsqrtf(ss * p.inv_dim + p.q_eps);
    }
    cs[i] = c;
  }
  if (tid < 32 * NU)
    bl[tid] = tid >= p.Nk ? -1e30f : ((p.kbias && tid >= p.kbias_first) ? p.kbias[(size_t)b * p.kbias_stride + tid] * LOG2E : 0.f);
  // K fragment offsets: MFMA row l31 <-> key pi(l31) of the 32-key sub-tile
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kfo[ks] = pi * KROWB + (((2 * ks + hi) ^ (pi & 15)) << 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- the wave's units: sub-blocks sb0 + wave, + 4, + 8, ...; the next unit's Q is in flight under this unit's MFMAs / softmax / stores ----
  for (int j = 0; j < p.upw; ++j) {
    const int sb = sb0 + wave + NW * j;
    if (sb >= nsb) break;
    const bool more = j + 1 < p.upw && sb + NW < nsb;
    bf16x8 qn[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qn[ks] = qf[ks];
    if (more) load_q(sb + NW, qn);
    // S^T = K . Q^T : lane (q = l31, hi) holds keys 32 u + 16 hi + r
    f32x16 s[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(smem + u * 32 * KROWB + kfo[ks]);
        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[u], 0, 0, 0);
      }
    }
    // t = s c + table (bias, key-tail mask), maximum
    const float c = cs[(wave + NW * j) * 32 + l31];
    float mx = -1e30f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float* bk = bl + 32 * u + 16 * hi;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 t4 = *(const f32x4*)(bk + 4 * r4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = __fadd_rn(__fmul_rn(s[u][4 * r4 + e], c), t4[e]);
          s[u][4 * r4 + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[u][r] - mx);
        s[u][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 32, 64);
    const float rl = 1.0f / l;
    // P = bf16(p / l): 16 consecutive keys per lane and sub-tile = two 16-byte stores
    const int qr = sb * 32 + l31;
    if (qr < p.Nq) {
      char* prow = p.p + ((size_t)b * p.p_bs + (size_t)qr * p.ldp + (size_t)h * p.Lkp) * 2;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int k0 = 32 * u + 16 * hi;
        if (k0 < p.Lkp) {   // (Lkp % 16 == 0: a 16-key group is inside the padded row or outside it)
          u32x4 w0, w1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            w0[e] = pack_bf16x2(s[u][2 * e] * rl, s[u][2 * e + 1] * rl);
            w1[e] = pack_bf16x2(s[u][8 + 2 * e] * rl, s[u][8 + 2 * e + 1] * rl);
          }
          *(u32x4*)(prow + k0 * 2) = w0;
          *(u32x4*)(prow + k0 * 2 + 16) = w1;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = qn[ks];
  }
}

}  // namespace

extern "C" int v3a_xattn_probs_bf16(const v3a_xattn_probs_args* a, void* stream) {
  if (!a || !a->q || !a->k || !a->p) return V3A_ERR_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Nq <= 0 || a->Nk <= 0 || a->D != 128) return V3A_ERR_SHAPE;
  if (a->Nk > 128 || a->Lkp < a->Nk || a->Lkp % 16 || a->Lkp > 128) return V3A_ERR_SHAPE;
  if (a->ldq % 8 || a->ldk % 8 || a->ldp % 8 || a->q_batch_stride % 8 || a->k_batch_stride % 8 || a->p_batch_stride % 8) return V3A_ERR_SHAPE;
  if (a->ldp < a->H * a->Lkp) return V3A_ERR_SHAPE;
  if (a->key_bias && a->key_bias_stride < a->Nk) return V3A_ERR_SHAPE;
  XaP p = {};
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.p = (char*)a->p; p.kbias = a->key_bias;
  p.q_bs = a->q_batch_stride; p.k_bs = a->k_batch_stride; p.p_bs = a->p_batch_stride;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldp = a->ldp;
  p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk; p.Lkp = a->Lkp;
  p.kbias_stride = a->key_bias_stride; p.kbias_first = a->key_bias_first > 0 ? a->key_bias_first : 0;
  p.scale_log2e = a->scale * 1.4426950408889634f;
  if (a->q_row_sumsq) {
    if (a->q_sumsq_parts <= 0 || a->q_sumsq_parts % 4) return V3A_ERR_SHAPE;
    // the statistics are indexed by the DENSE row number b * Nq + m of the projection that produced q
    if (a->B > 1 && a->q_batch_stride != (long)a->Nq * a->ldq) return V3A_ERR_SHAPE;
    p.qsq = a->q_row_sumsq; p.qparts = a->q_sumsq_parts; p.q_eps = a->q_eps; p.inv_dim = 1.0f / (float)(a->H * 128);
  }
  const int nu = (a->Lkp + 31) / 32;   // sub-tiles over the PADDED row: every column of [0, Lkp) is written (zeros beyond the last key)
  const int nsb = (a->Nq + 31) / 32;
  // units per wave: 1 up to 2048 workgroups (the production launch, 2 x 12 heads x 4096 queries = 768 workgroups = 3 per CU, measured 14.0 us
  // against 15.4 / 15.7 at 2 / 4 units per wave, whose 384 / 192 workgroups load the CUs unevenly); beyond that a wave walks several
  // units with the next unit's Q in flight under the current one and a head's keys are read once per 128 upw queries
  int upw = (int)(((long)a->B * a->H * nsb + 4 * 2048 - 1) / (4 * 2048));
  upw = upw < 1 ? 1 : (upw > 8 ? 8 : upw);
  while (upw > 1 && (nsb + 4 * upw - 1) / (4 * upw) == (nsb + 4 * (upw - 1) - 1) / (4 * (upw - 1))) --upw;   // no longer walk than the split needs
  p.upw = upw;
  const long wgs = (long)a->B * a->H * ((nsb + 4 * upw - 1) / (4 * upw));
  if (wgs > 0x7fffffffL) return V3A_ERR_SHAPE;
  const int lds = nu * 32 * KROWB + 4 * upw * 32 * 4 + nu * 32 * 4;
  switch (nu) {
    case 1: hipLaunchKernelGGL(xattn_probs_kernel<1>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p); break;
    case 2: hipLaunchKernelGGL(xattn_probs_kernel<2>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p); break;
    case 3: hipLaunchKernelGGL(xattn_probs_kernel<3>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(xattn_probs_kernel<4>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p); break;
  }
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
