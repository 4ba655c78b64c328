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
};

template <int CPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nch = p.d >> 3;
  float v[CPL][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      if (p.x_f32) {
        const float* xp = (const float*)p.x + (size_t)row * p.ldx + c * 8;
        const f32x4 a = *(const f32x4*)xp, bq = *(const f32x4*)(xp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = bq[e]; }
      } else {
        const u32x4 raw = *(const u32x4*)(p.x + ((size_t)row * p.ldx + c * 8) * 2);
        unpack_bf16x8(raw, v[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)p.d;
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
  const size_t moff = (size_t)(row / p.rpb) * p.mstride;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
    if (p.w) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * p.w[c * 8 + e] + (p.b ? p.b[c * 8 + e] : 0.f);
    }
    if (p.scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * (1.f + p.scale[moff + c * 8 + e]) + p.shift[moff + c * 8 + e];
    }
    if (p.y_f32) {
      float* yp = (float*)p.y + (size_t)row * p.ldy + c * 8;
      f32x4 a, bq;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = o[e]; bq[e] = o[4 + e]; }
      *(f32x4*)yp = a;
      *(f32x4*)(yp + 4) = bq;
    } else {
      *(u32x4*)(p.y + ((size_t)row * p.ldy + c * 8) * 2) = pack_bf16x8(o);
    }
  }
}

struct RmsP {
  const char* x; char* y;
  const float* w;       // [d]
  const float* rope;    // [ntok][hd/2][2] (cos, sin) or null
  int M, d, ldx, ldy, hd, tokens_per_batch;
  float eps;
};

template <int CPL>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(const RmsP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nch = p.d >> 3;
  float v[CPL][8];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const u32x4 raw = *(const u32x4*)(p.x + ((size_t)row * p.ldx + c * 8) * 2);
      unpack_bf16x8(raw, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sq += v[i][e] * v[i][e];
    }
  }
  const float rs = rsqrtf(wave_sum(sq) / (float)p.d + p.eps);
  const int tok = row % p.tokens_per_batch;
  const int cph = p.hd >> 3;  // chunks per head
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[i][e] * rs * p.w[c * 8 + e];
    if (p.rope) {
      const float* rp = p.rope + ((size_t)tok * (p.hd >> 1) + (size_t)(c % cph) * 4) * 2;
      const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
      const float cs[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = o[2 * e], x1 = o[2 * e + 1], co = cs[2 * e], si = cs[2 * e + 1];
        o[2 * e] = x0 * co - x1 * si;
        o[2 * e + 1] = x0 * si + x1 * co;
      }
    }
    *(u32x4*)(p.y + ((size_t)row * p.ldy + c * 8) * 2) = pack_bf16x8(o);
  }
}

}  // namespace

#define DISPATCH_CPL(KERNEL, P, d, grid, stream)                                             \
  do {                                                                                       \
    const int cpl_ = ((d) / 8 + 63) / 64;                                                    \
    if (cpl_ <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, (hipStream_t)stream, P); \
    else if (cpl_ <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, (hipStream_t)stream, P); \
    else if (cpl_ <= 3) hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, (hipStream_t)stream, P); \
    else if (cpl_ <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, (hipStream_t)stream, P); \
    else if (cpl_ <= 10) hipLaunchKernelGGL(KERNEL<10>, grid, dim3(256), 0, (hipStream_t)stream, P); \
    else return V3A_ERR_SHAPE;                                                               \
  } while (0)

extern "C" int v3a_layernorm(const v3a_layernorm_args* a, void* stream) {
  if (!a || !a->x || !a->y) return V3A_ERR_ARG;
  if (a->M <= 0 || a->d <= 0 || a->d % 8 || a->ldx % 8 || a->ldy % 8) return V3A_ERR_SHAPE;
  if ((a->scale == nullptr) != (a->shift == nullptr)) return V3A_ERR_ARG;
  LnP p;
  p.x = (const char*)a->x; p.y = (char*)a->y; p.w = a->weight; p.b = a->bias;
  p.scale = a->scale; p.shift = a->shift;
  p.M = a->M; p.d = a->d; p.ldx = a->ldx; p.ldy = a->ldy;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : a->M; p.mstride = a->mod_stride;
  p.eps = a->eps; p.x_f32 = a->x_is_f32; p.y_f32 = a->y_is_f32;
  const dim3 grid((a->M + 3) / 4);
  DISPATCH_CPL(layernorm_kernel, p, a->d, grid, stream);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_rmsnorm_rope(const v3a_rmsnorm_rope_args* a, void* stream) {
  if (!a || !a->x || !a->y || !a->weight) return V3A_ERR_ARG;
  if (a->M <= 0 || a->d <= 0 || a->d % 8 || a->ldx % 8 || a->ldy % 8) return V3A_ERR_SHAPE;
  if (a->rope && (a->head_dim <= 0 || a->head_dim % 8 || a->d % a->head_dim)) return V3A_ERR_SHAPE;
  RmsP p;
  p.x = (const char*)a->x; p.y = (char*)a->y; p.w = a->weight; p.rope = a->rope;
  p.M = a->M; p.d = a->d; p.ldx = a->ldx; p.ldy = a->ldy;
  p.hd = a->head_dim > 0 ? a->head_dim : a->d;
  p.tokens_per_batch = a->tokens_per_batch > 0 ? a->tokens_per_batch : a->M;
  p.eps = a->eps;
  const dim3 grid((a->M + 3) / 4);
  DISPATCH_CPL(rmsnorm_rope_kernel, p, a->d, grid, stream);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
