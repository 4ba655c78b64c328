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
// One workgroup = 128 queries (4 waves x 32) of one (batch item, head).  All Lk <= 128 keys of the head sit in LDS (two 64-key tiles in
// the flash kernel's image: 256-byte rows, 16-byte chunks XOR-swizzled by row); S^T = K.Q^T on v_mfma_f32_32x32x16_bf16 with the
// flash kernel's permuted key -> row map, so that lane (query = l31, hi) ends up with 16 CONSECUTIVE keys per 32-key sub-tile: the
// softmax is a per-lane loop plus one cross-half exchange, and a lane stores its probabilities as 32 contiguous bytes.
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
};

constexpr int D = 128, KV = 64, KROWB = D * 2, KTILE = KV * KROWB, NW = 4;
constexpr int KINS = KTILE / 1024 / NW;   // 4 DMA instructions per wave per key tile

template <int NT>   // key tiles (1: Lk <= 64, 2: Lk <= 128)
__global__ __launch_bounds__(256, 4) void xattn_probs_kernel(const XaP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int nqb = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // consecutive ids = the query blocks of one (batch, head): its keys stay in one L2
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * (NW * 32) + wave * 32;
  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;

  // ---- keys -> LDS (rows clamped to the last key: the excess rows are masked below) ----
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < KINS; ++j) {
      const int g = j * NW + wave;
      const int r = g * 4 + lane / 16, c = lane % 16;
      int row = t * KV + r;
      row = row < p.Nk ? row : p.Nk - 1;
      glds16(Kb + (size_t)row * p.ldk * 2 + (size_t)(c ^ (r & 15)) * 16, smem + t * KTILE + g * 1024);
    }
  // ---- Q fragments (B operand): lane (q = l31, hi) slot j <-> d = 16 ks + 8 hi + j ----
  bf16x8 qf[8];
  {
    int qr = q0 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  }
  // K fragment offsets: MFMA row l31 <-> key pi(l31) of the 32-key sub-tile
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kfo[ks] = pi * KROWB + (((2 * ks + hi) ^ (pi & 15)) << 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- S^T = K . Q^T : lane (q = l31, hi) holds keys 64 t + 32 u + 16 hi + r ----
  f32x16 s[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(smem + t * KTILE + u * 32 * KROWB + kfo[ks]);
        s[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t][u], 0, 0, 0);
      }
    }
  // ---- the query's RMS factor (q not normalised yet): parts added in index order ----
  float c = p.scale_log2e;
  if (p.qsq) {
    const float ss = sum_parts_in_order(p.qsq + ((size_t)b * p.Nq + min(q0 + l31, p.Nq - 1)) * p.qparts, p.qparts);
    c *= rsqrtf(ss * p.inv_dim + p.q_eps);
  }
  // ---- t = s c + bias log2 e (merged padding key), key-tail mask, maximum ----
  float mx = -1e30f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k0 = 64 * t + 32 * u + 16 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + r;
        float v = s[t][u][r] * c;
        if (p.kbias && key >= p.kbias_first && key < p.Nk) v += p.kbias[(size_t)b * p.kbias_stride + key] * 1.4426950408889634f;
        v = key < p.Nk ? v : -1e30f;
        s[t][u][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][u][r] - mx);
        s[t][u][r] = e;
        l += e;
      }
  l += __shfl_xor(l, 32, 64);
  const float rl = 1.0f / l;
  // ---- P = bf16(p / l): 16 consecutive keys per lane and sub-tile = two 16-byte stores ----
  const int qr = q0 + l31;
  if (qr < p.Nq) {
    char* prow = p.p + ((size_t)b * p.p_bs + (size_t)qr * p.ldp + (size_t)h * p.Lkp) * 2;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k0 = 64 * t + 32 * u + 16 * hi;
        if (k0 < p.Lkp) {   // (Lkp % 16 == 0: a 16-key group is inside the padded row or outside it)
          u32x4 w0, w1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            w0[e] = pack_bf16x2(s[t][u][2 * e] * rl, s[t][u][2 * e + 1] * rl);
            w1[e] = pack_bf16x2(s[t][u][8 + 2 * e] * rl, s[t][u][8 + 2 * e + 1] * rl);
          }
          *(u32x4*)(prow + k0 * 2) = w0;
          *(u32x4*)(prow + k0 * 2 + 16) = w1;
        }
      }
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
    p.qsq = a->q_row_sumsq; p.qparts = a->q_sumsq_parts; p.q_eps = a->q_eps; p.inv_dim = 1.0f / (float)(a->H * 128);
  }
  const int nt = a->Nk > 64 ? 2 : 1;
  const long wgs = (long)a->B * a->H * ((a->Nq + 127) / 128);
  if (wgs > 0x7fffffffL) return V3A_ERR_SHAPE;
  if (nt == 1) hipLaunchKernelGGL(xattn_probs_kernel<1>, dim3((unsigned)wgs), dim3(256), KTILE, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(xattn_probs_kernel<2>, dim3((unsigned)wgs), dim3(256), 2 * KTILE, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
