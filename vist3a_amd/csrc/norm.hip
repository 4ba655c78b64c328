// Row-wise normalisation kernels (HBM-bound): one 64-lane wave owns one token row, 16-byte loads,
// the row stays in registers between the statistics pass and the apply pass (one read, one write).
//   v3a_layernorm      : FP32LayerNorm (+ optional affine) (+ optional AdaLN  *(1+scale)+shift )
//                        diffusers==0.33.1 WanTransformerBlock norm1/norm2/norm3/norm_out (SURVEY §8 A5,A9);
//                        vggt/layers/block.py:41-47 norm1/norm2 (R4,R6,R7); attention.py q_norm/k_norm via hd rows.
//   v3a_rmsnorm_rope   : RMSNorm over the full width ("rms_norm_across_heads") + optional complex RoPE on
//                        adjacent pairs per head — WanAttnProcessor2_0 norm_q/norm_k + apply_rotary_emb (A6,A7).
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

struct LnP {
  const char* x; char* y;
  const float* w; const float* b;        // affine, may be null
  const float* scale; const float* shift;  // modulation [nb][mstride], may be null
  int M, d, ldx, ldy;
  int rpb, mstride;
  float eps;
  int x_f32, y_f32;
  int ig, is, io, og, os, oo;   // row remaps: in row = m + (m/ig)*is + io (ig>0), out row likewise
  int rms;                      // 1: no mean subtraction
  float* y_scale;               // non-null: e4m3 output with per-row dynamic scale
  int stage;                    // 1: the workgroup's rows share their per-column parameters (AdaLN scale | shift of one batch item, else the
                                //    affine weight | bias): staged once per workgroup in LDS (2 x ceil(d / 256) KiB) by LDS-DMA
};

// Per-column parameters of a row-normalisation workgroup (4 waves = 4 rows) -> LDS: two arrays of d floats as whole 1-KiB DMA pieces (lanes
// past d re-read the array's last float4; a null second array repeats the first and is never read).  Issued BEFORE the row loads: the wave's
// own wait for its row covers its pieces (in-order vmcnt), the workgroup barrier the other waves'.  Read back after the reductions at LDS
// latency - loaded from L2 there (twelve 16-byte loads per lane in three dependent groups, there being no registers to hold them across the
// reductions at 8 waves per SIMD) they cost the AdaLN LayerNorm 1.4 of its 10.3 us.
__device__ __forceinline__ void stage_params(const float* a0, const float* a1, int d, char* lds, int wave, int lane, int narr = 2) {
  const int np = (d + 255) >> 8;
  for (int pc = wave; pc < narr * np; pc += 4) {
    const int arr = pc >= np, q = pc - arr * np;
    const float* src = (arr && a1) ? a1 : a0;
    glds16(src + min(q * 256 + lane * 4, d - 4), lds + (size_t)pc * 1024);
  }
}

// F8OUT: e4m3 output with per-row scales (a separate instantiation: its extra live registers would cost the bf16 kernel a wave per
// SIMD, and 8192 rows are exactly 8 waves per SIMD - one round)
template <int CPL, bool F8OUT = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnP p) {
  extern __shared__ __attribute__((aligned(16))) char lnsm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rowu = blockIdx.x * 4 + wave;
  if (!p.stage && rowu >= p.M) return;            // (staging workgroups keep every wave for the barrier: a wave past M recomputes row M - 1)
  const bool live_row = rowu < p.M;
  const int row0 = live_row ? rowu : p.M - 1;
  const size_t row = p.ig > 0 ? (size_t)row0 + (size_t)(row0 / p.ig) * p.is + p.io : (size_t)row0;
  const size_t orow = p.og > 0 ? (size_t)row0 + (size_t)(row0 / p.og) * p.os + p.oo : (size_t)row0;
  const int nch = p.d >> 3;
  if (p.stage) {   // staged arrays: AdaLN (scale, shift) when modulated, else the affine (weight, bias)
    const size_t mo = (size_t)((blockIdx.x * 4) / p.rpb) * p.mstride;
    if (p.scale) stage_params(p.scale + mo, p.shift + mo, p.d, lnsm, wave, lane);
    else stage_params(p.w, p.b, p.d, lnsm, wave, lane);
  }
  float v[CPL][8];
  float sum = 0.f;
  // All of the row's loads are issued back to back, branch-free (chunks past the row end read the last chunk and are zeroed):
  // under a per-lane `if (c < nch)` hipcc wraps every load in exec-mask control flow and waits for each one before issuing the next -
  // three serial memory round trips per row at d = 1536.
  if (p.x_f32) {   // (wave-uniform)
    f32x4 ra[CPL], rb[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const float* xp = (const float*)p.x + row * p.ldx + min(lane + i * 64, nch - 1) * 8;
      ra[i] = *(const f32x4*)xp; rb[i] = *(const f32x4*)(xp + 4);
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = ra[i][e]; v[i][4 + e] = rb[i][e]; }
  } else {
    u32x4 raw[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) raw[i] = *(const u32x4*)(p.x + (row * p.ldx + min(lane + i * 64, nch - 1) * 8) * 2);
#pragma unroll
    for (int i = 0; i < CPL; ++i) unpack_bf16x8(raw[i], v[i]);
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const bool live = lane + i * 64 < nch;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[i][e] = live ? v[i][e] : 0.f; sum += v[i][e]; }
  }
  const float mean = p.rms ? 0.f : wave_sum(sum) / (float)p.d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[i][e] - mean; sq += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.d + p.eps);
  const size_t moff = (size_t)(row0 / p.rpb) * p.mstride;
  if (p.stage) {
    __syncthreads();                              // every wave's parameter pieces have landed (its own: in order before its row)
    if (!live_row) return;
  }
  const bool lds_mod = p.stage && p.scale, lds_aff = p.stage && !p.scale;
  const int apitch = ((p.d + 255) >> 8) << 8;     // floats between the two staged arrays
  // 8 consecutive parameters of chunk c: from the staged copy (arr = 0 / 1) or from global memory
  auto ld8 = [&](bool staged, int arr, const float* g, int c, f32x4& a, f32x4& b) {
    if (staged) {   // (explicit LDS address space: through a generic pointer hipcc emits flat loads for BOTH branches)
      const LDS_AS f32x4* l = (const LDS_AS f32x4*)(lnsm + (size_t)(arr * apitch + c * 8) * 4);
      a = l[0]; b = l[1];
    } else {
      a = *(const f32x4*)(g + c * 8); b = *(const f32x4*)(g + c * 8 + 4);
    }
  };
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
    if (p.w) {
      f32x4 w0, w1;
      ld8(lds_aff, 0, p.w, c, w0, w1);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] *= w0[e]; o[4 + e] *= w1[e]; }
      if (p.b) {
        f32x4 b0, b1;
        ld8(lds_aff, 1, p.b, c, b0, b1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] += b0[e]; o[4 + e] += b1[e]; }
      }
    }
    if (p.scale) {
      f32x4 s0, s1, h0, h1;
      ld8(lds_mod, 0, p.scale + moff, c, s0, s1);
      ld8(lds_mod, 1, p.shift + moff, c, h0, h1);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = o[e] * (1.f + s0[e]) + h0[e]; o[4 + e] = o[4 + e] * (1.f + s1[e]) + h1[e]; }
    }
    if constexpr (F8OUT) {   // keep the bf16-rounded result in registers; the row maximum decides the e4m3 scale below
      unpack_bf16x8(pack_bf16x8(o), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
    } else if (p.y_f32) {
      float* yp = (float*)p.y + orow * p.ldy + c * 8;
      f32x4 a, bq;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = o[e]; bq[e] = o[4 + e]; }
      *(f32x4*)yp = a;
      *(f32x4*)(yp + 4) = bq;
    } else {
      *(u32x4*)(p.y + (orow * p.ldy + c * 8) * 2) = pack_bf16x8(o);
    }
  }
  if constexpr (F8OUT) {   // same arithmetic as quantize_fp8_rows_kernel (attention_fp8.hip) on the bf16 row
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc = fmaxf(amax, 1e-12f) / 448.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) p.y_scale[orow] = sc;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + i * 64;
      if (c >= nch) continue;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(v[i][e] * inv, -448.f), 448.f);
      u32x2 q;
      q[0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false), true);
      q[1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false), true);
      *(u32x2*)(p.y + orow * p.ldy + c * 8) = q;
    }
  }
}

struct RmsP {
  const char* x; char* y;
  const float* w;       // [d]
  const float* w2;      // second column block's weight (blockIdx.y = 1), or null
  const float* rope;    // [ntok][hd/2][2] (cos, sin) or null
  int M, d, ldx, ldy, hd, tokens_per_batch;
  float eps;
  float f8_inv_scale;   // > 0: y is e4m3 BYTES [M][ldy], value = e4m3(clamp(bf16(result) * f8_inv_scale)) (operands of the fp8 attention)
  int stage;            // 1: the weight is staged once per workgroup in LDS (stage_params)
};

template <int CPL, bool F8OUT = false>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(const RmsP pin) {
  RmsP p = pin;
  if (blockIdx.y) { p.x += (size_t)p.d * 2; p.y += (size_t)p.d * (F8OUT ? 1 : 2); p.w = p.w2; }   // the k half of a fused q|k projection
  extern __shared__ __attribute__((aligned(16))) char lnsm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rowu = blockIdx.x * 4 + wave;
  if (!p.stage && rowu >= p.M) return;
  const bool live_row = rowu < p.M;
  const int row = live_row ? rowu : p.M - 1;
  const int nch = p.d >> 3;
  if (p.stage) stage_params(p.w, nullptr, p.d, lnsm, wave, lane, 1);
  // the rotary factors of a lane's chunks: chunk c = lane + 64 i sits at position c % (hd / 8) of its head, which is the same for every i
  // when hd / 8 divides 64 (hd = 128: 16) - one pair of loads per lane, issued with the row's loads, instead of one pair per chunk behind
  // the reduction
  const int tok = row % p.tokens_per_batch;
  const int cph = p.hd >> 3;  // chunks per head
  const bool rope_once = p.rope && (64 % cph) == 0;
  f32x4 rr0 = {0.f, 0.f, 0.f, 0.f}, rr1 = rr0;
  if (rope_once) {
    const float* rp = p.rope + ((size_t)tok * (p.hd >> 1) + (size_t)(lane % cph) * 4) * 2;
    rr0 = *(const f32x4*)rp; rr1 = *(const f32x4*)(rp + 4);
  }
  float v[CPL][8];
  float sq = 0.f;
  {   // branch-free, all loads in flight together (see layernorm_kernel)
    u32x4 raw[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) raw[i] = *(const u32x4*)(p.x + ((size_t)row * p.ldx + min(lane + i * 64, nch - 1) * 8) * 2);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      unpack_bf16x8(raw[i], v[i]);
      const bool live = lane + i * 64 < nch;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = live ? v[i][e] : 0.f; sq += v[i][e] * v[i][e]; }
    }
  }
  const float rs = rsqrtf(wave_sum(sq) / (float)p.d + p.eps);
  if (p.stage) {
    __syncthreads();                              // every wave's weight pieces have landed
    if (!live_row) return;
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    float o[8];
    {
      f32x4 w0, w1;
      if (p.stage) {
        const LDS_AS f32x4* l = (const LDS_AS f32x4*)(lnsm + (size_t)c * 32);
        w0 = l[0]; w1 = l[1];
      } else {
        w0 = *(const f32x4*)(p.w + c * 8); w1 = *(const f32x4*)(p.w + c * 8 + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = v[i][e] * rs * w0[e]; o[4 + e] = v[i][4 + e] * rs * w1[e]; }
    }
    if (p.rope) {
      f32x4 r0 = rr0, r1 = rr1;
      if (!rope_once) {
        const float* rp = p.rope + ((size_t)tok * (p.hd >> 1) + (size_t)(c % cph) * 4) * 2;
        r0 = *(const f32x4*)rp; r1 = *(const f32x4*)(rp + 4);
      }
      const float cs[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = o[2 * e], x1 = o[2 * e + 1], co = cs[2 * e], si = cs[2 * e + 1];
        o[2 * e] = x0 * co - x1 * si;
        o[2 * e + 1] = x0 * si + x1 * co;
      }
    }
    if constexpr (F8OUT) {   // same two roundings as rmsnorm_rope -> bf16 -> v3a_quantize_fp8
      float f[8];
      unpack_bf16x8(pack_bf16x8(o), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * p.f8_inv_scale, -448.f), 448.f);
      u32x2 q;
      q[0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false), true);
      q[1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false), true);
      *(u32x2*)(p.y + (size_t)row * p.ldy + c * 8) = q;
    } else {
      *(u32x4*)(p.y + ((size_t)row * p.ldy + c * 8) * 2) = pack_bf16x8(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Channel norm over short rows (channels-last pixels): LPR lanes (power of two) share one row, CPL
// 16-byte chunks each; 64/LPR rows per wave.  mode 0: RMSNorm  x*rsqrt(mean(x^2)+eps)*w
//                                             mode 1: F.normalize semantics  x/max(||x||,1e-12)*sqrt(d)*w (+b)
// optional SiLU.  WanRMS_norm + nonlinearity: /root/reference/utils/wan_utils.py:150-184, :370-372.
struct RowNormP {
  const char* x; char* y;
  const float* w; const float* b;
  long M;
  int d, ldx, ldy, lpr, mode, act;
  float eps;
};

template <int CPL>
__global__ __launch_bounds__(256) void rownorm_kernel(const RowNormP p) {
  const int lane = threadIdx.x & 63;
  const int lpr = p.lpr, rpw = 64 / lpr;
  const int j = lane & (lpr - 1), r = lane / lpr;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + r;
  const int nch = p.d >> 3;
  const bool live = row < p.M;
  float v[CPL][8];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = j + i * lpr;
    if (live && c < nch) {
      const u32x4 raw = *(const u32x4*)(p.x + ((size_t)row * p.ldx + c * 8) * 2);
      unpack_bf16x8(raw, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sq += v[i][e] * v[i][e];
    }
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  float rs;
  if (p.mode == 0) rs = rsqrtf(sq / (float)p.d + p.eps);
  else rs = sqrtf((float)p.d) / fmaxf(sqrtf(sq), 1e-12f);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = j + i * lpr;
    if (c >= nch) continue;
    float o[8];
    const f32x4 w0 = *(const f32x4*)(p.w + c * 8), w1 = *(const f32x4*)(p.w + c * 8 + 4);
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (p.b) { b0 = *(const f32x4*)(p.b + c * 8); b1 = *(const f32x4*)(p.b + c * 8 + 4); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[i][e] * rs * (e < 4 ? w0[e & 3] : w1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
      if (p.act == V3A_ACT_SILU) t = silu(t);
      o[e] = t;
    }
    *(u32x4*)(p.y + ((size_t)row * p.ldy + c * 8) * 2) = pack_bf16x8(o);
  }
}

// ---------------------------------------------------------------------------------------------
// Row softmax  P = softmax(scale * S), S f32 [M][N] -> P bf16 [M][N]; one wave per row, online
// max/sum pass then a normalise pass (second read is L2 resident).  Used by the single-head
// C=384 VAE mid-block attention (/root/reference/utils/wan_utils.py:428-475) whose head width
// exceeds the flash kernel's register budget.
struct SoftmaxP { const float* s; char* p; int M, N, lds, ldp; float scale_log2e; };

__global__ __launch_bounds__(256) void softmax_rows_kernel(const SoftmaxP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const float* sr = p.s + (size_t)row * p.lds;
  const int nv = p.N >> 2;
  float m = -1e30f, l = 0.f;
  for (int i = lane; i < nv; i += 64) {
    const f32x4 x = *(const f32x4*)(sr + i * 4);
    const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])) * p.scale_log2e;
    const float mn = fmaxf(m, mx);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += __builtin_amdgcn_exp2f(x[e] * p.scale_log2e - mn);
    l = l * __builtin_amdgcn_exp2f(m - mn) + acc;
    m = mn;
  }
  const float mw = wave_max(m);
  l = wave_sum(l * __builtin_amdgcn_exp2f(m - mw));
  const float inv = 1.f / l;
  char* pr = p.p + (size_t)row * p.ldp * 2;
  for (int i = lane; i < nv; i += 64) {
    const f32x4 x = *(const f32x4*)(sr + i * 4);
    u32x2 o;
    o[0] = pack_bf16x2(__builtin_amdgcn_exp2f(x[0] * p.scale_log2e - mw) * inv, __builtin_amdgcn_exp2f(x[1] * p.scale_log2e - mw) * inv);
    o[1] = pack_bf16x2(__builtin_amdgcn_exp2f(x[2] * p.scale_log2e - mw) * inv, __builtin_amdgcn_exp2f(x[3] * p.scale_log2e - mw) * inv);
    *(u32x2*)(pr + i * 8) = o;
  }
}

}  // namespace

#define DISPATCH_CPL(KERNEL, P, d, grid, lds, stream)                                             \
  do {                                                                                       \
    const int cpl_ = ((d) / 8 + 63) / 64;                                                    \
    if (cpl_ <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), lds, (hipStream_t)stream, P); \
    else if (cpl_ <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), lds, (hipStream_t)stream, P); \
    else if (cpl_ <= 3) hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), lds, (hipStream_t)stream, P); \
    else if (cpl_ <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), lds, (hipStream_t)stream, P); \
    else if (cpl_ <= 10) hipLaunchKernelGGL(KERNEL<10>, grid, dim3(256), lds, (hipStream_t)stream, P); \
    else return V3A_ERR_SHAPE;                                                               \
  } while (0)

extern "C" int v3a_layernorm(const v3a_layernorm_args* a, void* stream) {
  if (!a || !a->x || !a->y) return V3A_ERR_ARG;
  if (a->M <= 0 || a->d <= 0 || a->d % 8 || a->ldx % 8 || a->ldy % 8) return V3A_ERR_SHAPE;
  if ((a->scale == nullptr) != (a->shift == nullptr)) return V3A_ERR_ARG;
  LnP p = {};
  p.x = (const char*)a->x; p.y = (char*)a->y; p.w = a->weight; p.b = a->bias;
  p.scale = a->scale; p.shift = a->shift;
  p.M = a->M; p.d = a->d; p.ldx = a->ldx; p.ldy = a->ldy;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : a->M; p.mstride = a->mod_stride;
  p.eps = a->eps; p.x_f32 = a->x_is_f32; p.y_f32 = a->y_is_f32;
  p.ig = a->in_row_group; p.is = a->in_row_skip; p.io = a->in_row_off;
  p.og = a->out_row_group; p.os = a->out_row_skip; p.oo = a->out_row_off;
  p.rms = a->rms;
  p.y_scale = a->y_fp8_scale;
  if (p.y_scale && p.y_f32) return V3A_ERR_ARG;
  const dim3 grid((a->M + 3) / 4);
  // per-column parameters staged in LDS when the 4 rows of a workgroup share them and the two arrays fit 16 KB (8 workgroups per CU stay resident)
  const int np = (a->d + 255) / 256;
  p.stage = ((p.scale && p.shift && p.rpb % 4 == 0) || (!p.scale && p.w)) && a->d >= 4 && 2 * np * 1024 <= 16384 && a->M >= 64;
  const size_t lds = p.stage ? (size_t)2 * np * 1024 : 0;
  if (p.y_scale) {
    const int cpl = (a->d / 8 + 63) / 64;
    if (cpl <= 3) hipLaunchKernelGGL((layernorm_kernel<3, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else if (cpl <= 10) hipLaunchKernelGGL((layernorm_kernel<10, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else return V3A_ERR_SHAPE;
    return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
  }
  DISPATCH_CPL(layernorm_kernel, p, a->d, grid, lds, stream);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_rmsnorm_rope(const v3a_rmsnorm_rope_args* a, void* stream) {
  if (!a || !a->x || !a->y || !a->weight) return V3A_ERR_ARG;
  if (a->M <= 0 || a->d <= 0 || a->d % 8 || a->ldx % 8 || a->ldy % 8) return V3A_ERR_SHAPE;
  if (a->rope && (a->head_dim <= 0 || a->head_dim % 8 || a->d % a->head_dim)) return V3A_ERR_SHAPE;
  RmsP p = {};
  p.x = (const char*)a->x; p.y = (char*)a->y; p.w = a->weight; p.rope = a->rope;
  p.M = a->M; p.d = a->d; p.ldx = a->ldx; p.ldy = a->ldy;
  p.hd = a->head_dim > 0 ? a->head_dim : a->d;
  p.tokens_per_batch = a->tokens_per_batch > 0 ? a->tokens_per_batch : a->M;
  p.eps = a->eps;
  p.w2 = a->weight2;
  p.f8_inv_scale = a->y_fp8_scale > 0.f ? 1.0f / a->y_fp8_scale : 0.f;
  if (a->y_fp8_scale < 0.f || (a->y_fp8_scale > 0.f && a->y == a->x)) return V3A_ERR_ARG;   // (bytes cannot overwrite the bf16 input in place)
  const dim3 grid((a->M + 3) / 4, a->weight2 ? 2 : 1);
  const int np = (a->d + 255) / 256;
  p.stage = np * 1024 <= 16384 && a->M >= 64;   // the weight through LDS (<= 16 KB per workgroup: 8 workgroups per CU stay resident)
  const size_t lds = p.stage ? (size_t)np * 1024 : 0;
  if (p.f8_inv_scale > 0.f) {   // (its own instantiations: the extra live registers would cost the bf16 kernels a wave per SIMD)
    const int cpl = (a->d / 8 + 63) / 64;
    if (cpl <= 3) hipLaunchKernelGGL((rmsnorm_rope_kernel<3, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else if (cpl <= 10) hipLaunchKernelGGL((rmsnorm_rope_kernel<10, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else return V3A_ERR_SHAPE;
    return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
  }
  DISPATCH_CPL(rmsnorm_rope_kernel, p, a->d, grid, lds, stream);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_rownorm_act(const v3a_rownorm_args* a, void* stream) {
  if (!a || !a->x || !a->y || !a->weight) return V3A_ERR_ARG;
  if (a->M <= 0 || a->d <= 0 || a->d % 8 || a->ldx % 8 || a->ldy % 8) return V3A_ERR_SHAPE;
  const int nch = a->d / 8;
  // choose chunks-per-lane / lanes-per-row (power of two) with the least idle lanes
  int best_cpl = 0, best_lpr = 0, best_waste = 1 << 30;
  for (int cpl = 1; cpl <= 4; ++cpl) {
    int need = (nch + cpl - 1) / cpl, lpr = 1;
    while (lpr < need) lpr <<= 1;
    if (lpr > 64) continue;
    const int waste = lpr * cpl - nch;
    if (waste < best_waste) { best_waste = waste; best_cpl = cpl; best_lpr = lpr; }
  }
  if (!best_cpl) return V3A_ERR_SHAPE;
  RowNormP p = {};
  p.x = (const char*)a->x; p.y = (char*)a->y; p.w = a->weight; p.b = a->bias;
  p.M = a->M; p.d = a->d; p.ldx = a->ldx; p.ldy = a->ldy; p.lpr = best_lpr; p.mode = a->mode; p.act = a->act;
  p.eps = a->eps;
  const long rpb = 4L * (64 / best_lpr);
  const dim3 grid((unsigned)((a->M + rpb - 1) / rpb));
  switch (best_cpl) {
    case 1: hipLaunchKernelGGL(rownorm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
    case 2: hipLaunchKernelGGL(rownorm_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
    case 3: hipLaunchKernelGGL(rownorm_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(rownorm_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
  }
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_softmax_rows(const float* s, void* p, int M, int N, int lds, int ldp, float scale, void* stream) {
  if (!s || !p) return V3A_ERR_ARG;
  if (M <= 0 || N <= 0 || N % 4 || lds % 4 || ldp % 4) return V3A_ERR_SHAPE;
  SoftmaxP sp;
  sp.s = s; sp.p = (char*)p; sp.M = M; sp.N = N; sp.lds = lds; sp.ldp = ldp;
  sp.scale_log2e = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, sp);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
