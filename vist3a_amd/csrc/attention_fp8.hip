// fp8 (OCP e4m3) flash attention forward for gfx950 on the block-scaled matrix instruction
// v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, twice the bf16 MFMA rate) - BASELINE config #4
// ("Wan-14B ... fp8 MFMA attention"; model selected at /root/reference/utils/argument.py:399, /root/reference/inference_t23d.py:73).
// The reference has no fp8 path: this is an MI355X-side precision option of the self-attention launch, gated by the host, whose
// arithmetic is pinned by an e4m3-emulating mode of the oracle (oracle/wan_dit.py, `attention_fp8_emulated`).
//
// Arithmetic:  S = (q8 . k8^T) * scale  (fp32 accumulation of e4m3 products; q8 / k8 / v8 are the RNE e4m3 roundings of the
// bf16 operands divided by their per-tensor scales), online softmax in fp32 over 64-key tiles, P rounded to e4m3 AFTER a 2^8
// pre-scale (p in (0, 1] would lose everything below 2^-9 otherwise; the hardware block scale 2^-8 undoes it inside the MFMA),
// O = sum_tiles P8 . v8 / l with l accumulated from the unrounded p in fp32.
//
// Layout: Q, K [batch][token][head*128 + d] BYTES, V^T [head*128 + d][batch*stride + key] BYTES (what v3a_quantize_fp8 writes
// from the bf16 tensors of the bf16 path).  Per workgroup 4 waves x 32 queries, 64-key tiles, the swapped S^T = K . Q^T form
// of attention.hip with the same key permutation, so a lane ends the S phase holding 2 x 16 consecutive keys of one query:
// exactly the 32 k-values of its half of the PV instruction's B operand, fed straight from registers.
// (Within one lane-half the instruction pairs byte b of A with byte b of B, so any k order is fine as long as both agree.)
// K tile = 64 keys x 128 B and V^T tile = 128 rows x 64 B: 8 KB each, K double- / V^T single-buffered = 24 KB per workgroup,
// half the LDS-DMA instructions and half the LDS reads of the bf16 kernel, a quarter of its MFMA instructions.
#include "common.h"
#include "../../include/vist3a_hip.h"

// defined in attention.hip: merges key-split partial softmaxes (out = out_mul * sum_s w_s O_s / sum_s w_s l_s) into bf16 rows
int v3a_attn_combine_launch(const float* ws_o, const float* ws_ml, void* o, long o_bs, int ldo, int B, int Nq, int H, int D, int S,
                            float scale_log2e, float out_mul, void* stream);

namespace {


struct Attn8P {
  const char* q; const char* k; const char* vt; char* o;
  long q_bs, k_bs, vt_bs, o_bs;   // batch strides: q/k/vt in BYTES (= elements), o in bf16 elements
  int ldq, ldk, ldvt, ldo;
  int H, Nq, Nk;
  float scale_log2e;              // softmax scale * q_scale * k_scale * log2(e)
  float out_scale;                // v_scale
  int kv_seg;                     // > 0: keys live in per-rank slabs of kv_seg keys (see attention.hip), k_seg / vt_seg BYTES apart
  long k_seg, vt_seg;
  int kv_split, B;                // > 1: key tiles divided among kv_split workgroups, partials merged by v3a_attn_combine_launch
  float* ws_o; float* ws_ml;
};

constexpr int D = 128, NW = 4, KV = 64;
constexpr int KTILE = KV * D;       // 8 KB: 64 keys x 128 B
constexpr int VTILE = D * KV;       // 8 KB: 128 rows x 64 B
constexpr int P_SHIFT = 8;          // P is rounded to e4m3 as p * 2^8; the B-operand block scale is 2^-8
constexpr int SCALE_ONE = 0x7f7f7f7f, SCALE_P = ((127 - P_SHIFT) * 0x01010101);

__device__ __forceinline__ int pack_fp8x4(float a, float b, float c, float d) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, lo, true);
}

__global__ __launch_bounds__(NW * 64, 3) void attn_fwd_fp8_kernel(const Attn8P p) {
  constexpr int OPITCH = D * 2 + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int nqb = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int S = p.kv_split > 1 ? p.kv_split : 1;
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid0 % S, bid = bid0 / S;
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * (NW * 32) + wave * 32;

  const char* Qb = p.q + (size_t)b * p.q_bs + (size_t)h * D;
  const char* Kb = p.k + (size_t)b * p.k_bs + (size_t)h * D;
  const char* Vb = p.vt + (size_t)h * D * p.ldvt + (size_t)b * p.vt_bs;

  // ---- Q (B operand of S^T): lane (q = l31, hi), k-step s: the 32 bytes d = 64 s + 32 hi .. + 31 ----
  i32x8 qf[2];
  {
    int qr = q0 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq + hi * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4 a = *(const u32x4*)(qp + s * 64), c = *(const u32x4*)(qp + s * 64 + 16);
      qf[s][0] = a[0]; qf[s][1] = a[1]; qf[s][2] = a[2]; qf[s][3] = a[3];
      qf[s][4] = c[0]; qf[s][5] = c[1]; qf[s][6] = c[2]; qf[s][7] = c[3];
    }
  }
  // ---- DMA sources: K piece = 8 rows x 128 B, V^T piece = 16 rows x 64 B; two of each per wave and tile ----
  const char* kp[2];
  const char* vp[2];
  int krow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = j * NW + wave;
    const int r = g * 8 + (lane >> 3);                 // key row inside the tile
    krow[j] = r;
    kp[j] = Kb + (size_t)((lane & 7) ^ ((r >> 1) & 7)) * 16;   // LDS chunk c' of row r holds chunk c' ^ ((r >> 1) & 7)
    const int d = g * 16 + (lane >> 2);                // V^T row
    vp[j] = Vb + (size_t)d * p.ldvt + (size_t)((lane & 3) ^ ((d >> 2) & 3)) * 16;
  }
  auto stage_k = [&](int s, int kt) {
    // (slabs: a 64-key tile lies inside one segment; its base moves by k_seg - kv_seg * ldk bytes per segment crossed)
    const size_t so = p.kv_seg > 0 ? (size_t)((kt * KV) / p.kv_seg) * (size_t)(p.k_seg - (long)p.kv_seg * p.ldk) : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = kt * KV + krow[j];
      row = row < p.Nk ? row : p.Nk - 1;
      glds16(kp[j] + (size_t)row * p.ldk + so, smem + s * KTILE + (j * NW + wave) * 1024);
    }
  };
  auto stage_v = [&](int kt) {
    const size_t so = p.kv_seg > 0 ? (size_t)((kt * KV) / p.kv_seg) * (size_t)(p.vt_seg - p.kv_seg) : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(vp[j] + (size_t)kt * KV + so, smem + 2 * KTILE + (j * NW + wave) * 1024);
  };
  // ---- fragment offsets ----
  // K (A operand): MFMA row i = l31 <-> key pi(i) of the 32-key sub-tile; k-step s: chunks 4 s + 2 hi, + 1
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  const int kf = (pi >> 1) & 7;   // same for pi + 32
  int kfo[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) kfo[s][e] = pi * 128 + (((4 * s + 2 * hi + e) ^ kf) << 4);
  // V^T (A operand of PV): row d = l31 + 32 dt; bytes 0-15 <-> keys 16 hi + r, bytes 16-31 <-> keys 32 + 16 hi + r
  int vfo[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) vfo[e] = l31 * 64 + (((2 * e + hi) ^ ((l31 >> 2) & 3)) << 4);   // (d >> 2) & 3 is the same for d + 32 dt

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale_log2e;
  const int nkt = (p.Nk + KV - 1) / KV;
  const int kt0 = (int)((long)split * nkt / S), kt1 = (int)((long)(split + 1) * nkt / S);
  stage_k(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    stage_v(kt);
    if (kt + 1 < kt1) stage_k(cur ^ 1, kt + 1);
    const char* sK = smem + cur * KTILE;
    const char* sV = smem + 2 * KTILE;
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 a = *(const u32x4*)(sK + t * 32 * 128 + kfo[ks][0]), bq = *(const u32x4*)(sK + t * 32 * 128 + kfo[ks][1]);
        i32x8 kfr;
        kfr[0] = a[0]; kfr[1] = a[1]; kfr[2] = a[2]; kfr[3] = a[3]; kfr[4] = bq[0]; kfr[5] = bq[1]; kfr[6] = bq[2]; kfr[7] = bq[3];
        s[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kfr, qf[ks], s[t], 0, 0, 0, SCALE_ONE, 0, SCALE_ONE);
      }
    }
    // lane (q = l31, hi) holds keys kt*64 + 32 t + 16 hi + r
    if (kt == nkt - 1 && (p.Nk & (KV - 1))) {
      const int kb = kt * KV + 16 * hi;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb + 32 * t + r >= p.Nk) s[t][r] = -1e30f;
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c - (float)P_SHIFT;   // p * 2^8 = exp2(s c - m c + 8)
    m_run = m_new;
    float psum = 0.f;
    i32x8 pf;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(s[t][r] * c - mc);
        psum += pv[r];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) pf[4 * t + e] = pack_fp8x4(pv[4 * e], pv[4 * e + 1], pv[4 * e + 2], pv[4 * e + 3]);
    }
    l_run = l_run * alpha + psum;   // in units of 2^8
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // V^T pieces were issued before the next K pieces
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 a = *(const u32x4*)(sV + i * 32 * 64 + vfo[0]), bq = *(const u32x4*)(sV + i * 32 * 64 + vfo[1]);
      i32x8 vfr;
      vfr[0] = a[0]; vfr[1] = a[1]; vfr[2] = a[2]; vfr[3] = a[3]; vfr[4] = bq[0]; vfr[5] = bq[1]; vfr[6] = bq[2]; vfr[7] = bq[3];
      oacc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vfr, pf, oacc[i], 0, 0, 0, SCALE_ONE, 0, SCALE_P);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // see attention.hip: no V^T read may still be in flight
    __builtin_amdgcn_s_barrier();
  }

  // ---- finish ----
  l_run += __shfl_xor(l_run, 32, 64);
  if (S > 1) {   // unnormalised partial + (reference, sum in units of 2^8): merged by attn_combine_kernel with out_mul = 256 v_scale
    const int qr = q0 + l31;
    if (qr < p.Nq) {
      const size_t row = ((size_t)split * p.B + b) * p.Nq + qr;
      float* po = p.ws_o + (row * p.H + h) * D;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = oacc[i][g * 4 + e];
          *(f32x4*)(po + i * 32 + g * 8 + hi * 4) = v;
        }
      if (hi == 0) {
        float* pm = p.ws_ml + (row * p.H + h) * 2;
        pm[0] = m_run; pm[1] = l_run;
      }
    }
    return;
  }
  const float inv = p.out_scale * 256.0f / l_run;   // l is in units of 2^8 while the block scale already removed P's 2^8
  char* reg = smem + wave * (32 * OPITCH);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = i * 32 + g * 8 + hi * 4;
      u32x2 pk;
      pk[0] = pack_bf16x2(oacc[i][g * 4 + 0] * inv, oacc[i][g * 4 + 1] * inv);
      pk[1] = pack_bf16x2(oacc[i][g * 4 + 2] * inv, oacc[i][g * 4 + 3] * inv);
      *(u32x2*)(reg + l31 * OPITCH + d * 2) = pk;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  constexpr int CH = D / 8;
  char* Ob = p.o + ((size_t)b * p.o_bs + (size_t)h * D) * 2;
#pragma unroll
  for (int it = 0; it < 32 * CH / 64; ++it) {
    const int idx = it * 64 + lane;
    const int ql = idx / CH, ch = idx % CH;
    const u32x2 lo = *(const u32x2*)(reg + ql * OPITCH + ch * 16);
    const u32x2 hi2 = *(const u32x2*)(reg + ql * OPITCH + ch * 16 + 8);
    const int qr = q0 + ql;
    if (qr < p.Nq) {
      u32x4 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = hi2[0]; v[3] = hi2[1];
      *(u32x4*)(Ob + ((size_t)qr * p.ldo) * 2 + ch * 16) = v;
    }
  }
}

// bf16 -> e4m3 (RNE, clamped to +-448 so that nothing becomes NaN), y = fp8(x / scale); 16 elements per thread
struct QuantP { const char* x; char* y; long rows; int cols, ldx, ldy; float inv_scale; };
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const QuantP p) {
  const int cpr = p.cols / 16;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows * cpr) return;
  const long r = i / cpr;
  const int c = (int)(i % cpr) * 16;
  const char* src = p.x + ((size_t)r * p.ldx + c) * 2;
  float f[16];
  unpack_bf16x8(*(const u32x4*)src, f);
  unpack_bf16x8(*(const u32x4*)(src + 16), f + 8);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(f[4 * e + k] * p.inv_scale, -448.f), 448.f);
    o[e] = (unsigned)pack_fp8x4(v[0], v[1], v[2], v[3]);
  }
  *(u32x4*)(p.y + (size_t)r * p.ldy + c) = o;
}

// one wave per row: amax, then the scaled e4m3 image.  The row is held in registers between the two steps when it fits (cols <= 8192).
struct QuantRowsP { const char* x; char* y; float* scale; long rows; int cols, ldx, ldy; };
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const QuantRowsP p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const char* src = p.x + (size_t)r * p.ldx * 2;
  const int nch = p.cols / 8;   // 16-byte chunks of 8 bf16
  constexpr int HOLD = 16;      // chunks per lane kept in registers
  u32x4 held[HOLD];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < HOLD; ++i) {
    const int c = i * 64 + lane;
    held[i] = c < nch ? *(const u32x4*)(src + (size_t)c * 16) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < HOLD; ++i) {
    float f[8];
    unpack_bf16x8(held[i], f);
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
  }
  for (int c = HOLD * 64 + lane; c < nch; c += 64) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(src + (size_t)c * 16), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = fmaxf(amax, 1e-12f) / 448.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) p.scale[r] = sc;
  char* dst = p.y + (size_t)r * p.ldy;
  auto emit = [&](const u32x4 v, int c) {
    float f[8];
    unpack_bf16x8(v, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = fminf(fmaxf(f[k] * inv, -448.f), 448.f);
    u32x2 o;
    o[0] = (unsigned)pack_fp8x4(f[0], f[1], f[2], f[3]);
    o[1] = (unsigned)pack_fp8x4(f[4], f[5], f[6], f[7]);
    *(u32x2*)(dst + (size_t)c * 8) = o;
  };
#pragma unroll
  for (int i = 0; i < HOLD; ++i) {
    const int c = i * 64 + lane;
    if (c < nch) emit(held[i], c);
  }
  for (int c = HOLD * 64 + lane; c < nch; c += 64) emit(*(const u32x4*)(src + (size_t)c * 16), c);
}

}  // namespace

extern "C" int v3a_quantize_fp8_rows(const void* x, void* y, float* scale, long rows, int cols, int ldx, int ldy, void* stream) {
  if (!x || !y || !scale) return V3A_ERR_ARG;
  if (rows <= 0 || cols <= 0 || cols % 8 || ldx % 8 || ldy % 8 || ldx < cols || ldy < cols) return V3A_ERR_SHAPE;
  hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     QuantRowsP{(const char*)x, (char*)y, scale, rows, cols, ldx, ldy});
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_quantize_fp8(const void* x, void* y, long rows, int cols, int ldx, int ldy, float scale, void* stream) {
  if (!x || !y) return V3A_ERR_ARG;
  if (rows <= 0 || cols <= 0 || cols % 16 || ldx % 8 || ldy % 16 || ldx < cols || ldy < cols || !(scale > 0.f)) return V3A_ERR_SHAPE;
  const long n = rows * (cols / 16);
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     QuantP{(const char*)x, (char*)y, rows, cols, ldx, ldy, 1.0f / scale});
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_attention_fwd_fp8(const v3a_attn_fp8_args* a, void* stream) {
  if (!a || !a->q || !a->k || !a->vt || !a->o) return V3A_ERR_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Nq <= 0 || a->Nk <= 0 || a->D != 128) return V3A_ERR_SHAPE;
  if (a->ldq % 16 || a->ldk % 16 || a->ldvt % 16 || a->ldo % 8) return V3A_ERR_SHAPE;
  if (a->q_batch_stride % 16 || a->k_batch_stride % 16 || a->vt_batch_stride % 16 || a->o_batch_stride % 8) return V3A_ERR_SHAPE;
  if (!(a->q_scale > 0.f) || !(a->k_scale > 0.f) || !(a->v_scale > 0.f)) return V3A_ERR_ARG;
  Attn8P p = {};
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.vt = (const char*)a->vt; p.o = (char*)a->o;
  p.q_bs = a->q_batch_stride; p.k_bs = a->k_batch_stride; p.vt_bs = a->vt_batch_stride; p.o_bs = a->o_batch_stride;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldvt = a->ldvt; p.ldo = a->ldo;
  p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.scale_log2e = a->scale * a->q_scale * a->k_scale * 1.4426950408889634f;
  p.out_scale = a->v_scale;
  p.B = a->B; p.kv_split = a->kv_split > 1 ? a->kv_split : 1;
  p.kv_seg = a->kv_seg; p.k_seg = a->k_seg_stride; p.vt_seg = a->vt_seg_stride;
  if (a->kv_seg < 0 || a->kv_split < 0) return V3A_ERR_ARG;
  if (a->kv_seg > 0 && (a->kv_seg % 64 || a->Nk % a->kv_seg || a->k_seg_stride % 16 || a->vt_seg_stride % 16)) return V3A_ERR_SHAPE;
  if (p.kv_split > 1) {
    if (!a->workspace || p.kv_split > (a->Nk + 63) / 64) return V3A_ERR_ARG;
    p.ws_o = (float*)a->workspace;
    p.ws_ml = p.ws_o + (size_t)p.kv_split * a->B * a->Nq * a->H * D;
  }
  constexpr int OBYTES = NW * 32 * (D * 2 + 8);
  constexpr int LDS = (2 * KTILE + VTILE > OBYTES) ? 2 * KTILE + VTILE : OBYTES;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_fwd_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return V3A_ERR_LAUNCH;
    attr = true;
  }
  const int nqb = (a->Nq + NW * 32 - 1) / (NW * 32);
  hipLaunchKernelGGL(attn_fwd_fp8_kernel, dim3((unsigned)(nqb * a->B * a->H * p.kv_split)), dim3(NW * 64), LDS, (hipStream_t)stream, p);
  if (hipGetLastError() != hipSuccess) return V3A_ERR_LAUNCH;
  if (p.kv_split > 1)
    return v3a_attn_combine_launch(p.ws_o, p.ws_ml, a->o, a->o_batch_stride, a->ldo, a->B, a->Nq, a->H, D, p.kv_split, p.scale_log2e,
                                   256.0f * a->v_scale, stream);
  return V3A_OK;
}
