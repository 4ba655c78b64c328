// Shared device helpers for the gfx950 (CDNA4) kernels of the VIST3A hot path.
// Wave = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;         // 32 e4m3 = one 32x32x64 f8f6f4 MFMA A/B fragment (8 VGPR)

#define V3A_OK 0
#define V3A_ERR_ARG (-1)
#define V3A_ERR_SHAPE (-2)
#define V3A_ERR_LAUNCH (-3)
#define V3A_ERR_WORKSPACE (-4)

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: gfx950 does this in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); identical to torch's .to(bfloat16).  The (__bf16) cast is what makes hipcc emit it.
typedef __attribute__((ext_vector_type(2))) __bf16 v3a_bf16x2;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  v3a_bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }
__device__ __forceinline__ void unpack_bf16x8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return v;
}

// async global -> LDS copy of 16 B per lane; LDS destination is wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// GELU, tanh approximation (torch F.gelu(approximate="tanh")):  0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + 2^(-2u log2 e)),
// u = sqrt(2/pi) (x + 0.044715 x^3).  One v_exp_f32 + one v_rcp_f32 (1 ulp) per element instead of an IEEE division: the value
// is rounded to bf16 right after, eight mantissa bits above the approximation error.  Saturates correctly (2^inf -> rcp(inf) = 0).
__device__ __forceinline__ float gelu_tanh(float x) {
  const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float t = x * (c0 + c1 * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 resolution of 1 + erf): one v_rcp_f32, one v_exp_f32 and five FMAs
// instead of the ~45-instruction libdevice erff - the exact-GELU epilogue of the reconstruction MLPs (13416 x 4096 outputs per
// launch) was VALU-bound on it.  The result is rounded to bf16 right after.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(1.0f - p * t * e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// Wave-wide sum / maximum, the same value in every lane.  DPP cross-lane operands instead of six __shfl_xor steps: hipcc turns a
// shuffle into ds_bpermute_b32 (address arithmetic + an LDS-crossbar round trip of ~100 cycles), and a row-normalisation wave has one or
// two of these reductions between its loads and its stores - ~1 us of a 10 us launch.  Order of the sum: lanes of a quad, quads of a
// half row, half rows, rows 0+1 / 2+3, halves - fixed, so the result is deterministic (it is not the butterfly's association).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(0.f, v);          // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(0.f, v);          // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(0.f, v);         // row_half_mirror
  v += dpp_f32<0x140>(0.f, v);         // row_mirror: every lane of a 16-lane row holds the row's sum
  v += dpp_f32<0x142, 0xA>(0.f, v);    // row_bcast:15 into rows 1 and 3
  v += dpp_f32<0x143, 0xC>(0.f, v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v, v));
  v = fmaxf(v, dpp_f32<0x4E>(v, v));
  v = fmaxf(v, dpp_f32<0x141>(v, v));
  v = fmaxf(v, dpp_f32<0x140>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum of `parts` (% 4 == 0) consecutive floats in index order with the loads in flight TOGETHER: the partial sums of squares a GEMM epilogue
// emitted for one row (v3a_gemm_args.row_sumsq).  A plain `for (i < parts)` loop waits for every 16-byte load in turn - a dozen serial L2
// round trips in the prologue of every attention workgroup (measured: +18 us on the 185 us self-attention launch).
__device__ __forceinline__ float sum_parts_in_order(const float* sq, int parts) {
  float ss = 0.f;
  int i = 0;
  for (; i + 48 <= parts; i += 48) {   // 48 = 1536 / 32 (Wan-1.3B); 160 = 3 x 48 + 16 (Wan-14B)
    f32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = *(const f32x4*)(sq + i + 4 * j);
#pragma unroll
    for (int j = 0; j < 12; ++j) { ss += v[j][0]; ss += v[j][1]; ss += v[j][2]; ss += v[j][3]; }
  }
  for (; i + 16 <= parts; i += 16) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(sq + i + 4 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ss += v[j][0]; ss += v[j][1]; ss += v[j][2]; ss += v[j][3]; }
  }
  for (; i < parts; i += 4) {
    const f32x4 v = *(const f32x4*)(sq + i);
    ss += v[0]; ss += v[1]; ss += v[2]; ss += v[3];
  }
  return ss;
}

// Bijective XCD-aware remap of a linear workgroup id: hardware places block b on XCD b%8;
// give each XCD a contiguous chunk of the logical tile space so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
