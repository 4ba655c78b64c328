// Row / pixel passes of the fp32-equivalent DPT heads (include/vist3a_hip.h: "bf16 pair" tensors, v3a_conv_split).  The reference runs
// the camera / depth / Gaussian heads with autocast off (/root/reference/models/anysplat_stitched.py:335); here an fp32 activation is
// carried between the split-bf16 convolutions as two bf16 planes (hi, lo), x = hi + lo, so that the convolutions read MFMA operands
// directly.  These kernels compute in fp32 and only split at the store.  All of them are HBM-bound single passes: 16-byte accesses,
// one read and one write of every element.
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
  hi = pack_bf16x8(v);
  float h[8], r[8];
  unpack_bf16x8(hi, h);
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e] - h[e];   // exact in fp32
  lo = pack_bf16x8(r);
}
__device__ __forceinline__ void load_pair8(const char* hi, const char* lo, size_t elem, float* v) {
  float a[8], b[8];
  unpack_bf16x8(*(const u32x4*)(hi + elem * 2), a);
  unpack_bf16x8(*(const u32x4*)(lo + elem * 2), b);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = a[e] + b[e];
}

// ---------------------------------------------------------------------------------------------------------------------
// f32 -> pair (inputs that arrive as fp32: the context image of the Gaussian head's input_merger)
__global__ __launch_bounds__(256) void split_f32_kernel(const float* x, char* hi, char* lo, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  u32x4 h, l;
  split8(v, h, l);
  *(u32x4*)(hi + i * 16) = h;
  *(u32x4*)(lo + i * 16) = l;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm of fp32 token rows -> pair (dpt_head.py:213-216: self.norm on the tapped aggregator tokens, fp32).  One wave per row, the row
// in registers between the two statistics passes (mean, then centred variance: the two-pass form of F.layer_norm) and the store.
struct LnPairP {
  const float* x; char* hi; char* lo; const float* w; const float* b;
  int M, d, ldx, ldy; float eps; int ig, is, io;
};
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_pair_kernel(const LnPairP p) {
  const int lane = threadIdx.x & 63;
  const int row0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row0 >= p.M) return;
  const size_t row = p.ig > 0 ? (size_t)row0 + (size_t)(row0 / p.ig) * p.is + p.io : (size_t)row0;
  const int nch = p.d >> 3;
  f32x4 ra[CPL], rb[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {   // branch-free: chunks past the row end re-read the last chunk and are zeroed
    const float* xp = p.x + row * p.ldx + min(lane + i * 64, nch - 1) * 8;
    ra[i] = *(const f32x4*)xp; rb[i] = *(const f32x4*)(xp + 4);
  }
  float v[CPL][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const bool live = lane + i * 64 < nch;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[i][e] = live ? ra[i][e] : 0.f; v[i][4 + e] = live ? rb[i][e] : 0.f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[i][e];
  }
  const float mean = wave_sum(sum) / (float)p.d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    if (lane + i * 64 < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[i][e] - mean; sq += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.d + p.eps);
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
    if (p.w) {
      const f32x4 w0 = *(const f32x4*)(p.w + c * 8), w1 = *(const f32x4*)(p.w + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] *= w0[e]; o[4 + e] *= w1[e]; }
    }
    if (p.b) {
      const f32x4 b0 = *(const f32x4*)(p.b + c * 8), b1 = *(const f32x4*)(p.b + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] += b0[e]; o[4 + e] += b1[e]; }
    }
    u32x4 h, l;
    split8(o, h, l);
    *(u32x4*)(p.hi + ((size_t)row0 * p.ldy + c * 8) * 2) = h;
    *(u32x4*)(p.lo + ((size_t)row0 * p.ldy + c * 8) * 2) = l;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bilinear resize of a channels-last pair (+ optional pair addend, + optional f32 table broadcast over T): custom_interpolate /
// F.interpolate(mode="bilinear", align_corners=True) of dpt_head.py:291-309,460-466 and vggt_dpt_gs_head.py:166, in fp32.
// The arithmetic is v3a_bilinear_cl's (same source coordinates, same blend order).
struct BilPairP {
  const char* xh; const char* xl; char* yh; char* yl; const char* ah; const char* al; const float* tab;
  int T, h, w, H, W, C, align;
};
__global__ __launch_bounds__(256) void bilinear_cl_pair_kernel(const BilPairP p) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int cch = p.C >> 3;
  const long pix = gid / cch;
  if (pix >= (long)p.T * p.H * p.W) return;
  const int c0 = (int)(gid - pix * cch) * 8;
  const int t = (int)(pix / ((long)p.H * p.W));
  const int rem = (int)(pix - (long)t * p.H * p.W);
  const int oy = rem / p.W, ox = rem - oy * p.W;
  float sy, sx;
  if (p.align) {
    sy = p.H > 1 ? (float)(p.h - 1) / (float)(p.H - 1) * (float)oy : 0.f;
    sx = p.W > 1 ? (float)(p.w - 1) / (float)(p.W - 1) * (float)ox : 0.f;
  } else {
    sy = fmaxf(((float)oy + 0.5f) * ((float)p.h / (float)p.H) - 0.5f, 0.f);
    sx = fmaxf(((float)ox + 0.5f) * ((float)p.w / (float)p.W) - 0.5f, 0.f);
  }
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const size_t fb = (size_t)t * p.h * p.w;
  float a[8], b[8], c[8], d[8], o[8];
  load_pair8(p.xh, p.xl, (fb + (size_t)y0 * p.w + x0) * p.C + c0, a);
  load_pair8(p.xh, p.xl, (fb + (size_t)y0 * p.w + x1) * p.C + c0, b);
  load_pair8(p.xh, p.xl, (fb + (size_t)y1 * p.w + x0) * p.C + c0, c);
  load_pair8(p.xh, p.xl, (fb + (size_t)y1 * p.w + x1) * p.C + c0, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
  if (p.ah) {
    float f[8];
    load_pair8(p.ah, p.al, (size_t)pix * p.C + c0, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += f[e];
  }
  if (p.tab) {
    const float* tp = p.tab + (size_t)rem * p.C + c0;
    const f32x4 t0 = *(const f32x4*)tp, t1 = *(const f32x4*)(tp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] += t0[e]; o[4 + e] += t1[e]; }
  }
  u32x4 h, l;
  split8(o, h, l);
  *(u32x4*)(p.yh + ((size_t)pix * p.C + c0) * 2) = h;
  *(u32x4*)(p.yl + ((size_t)pix * p.C + c0) * 2) = l;
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int v3a_split_f32(const float* x, void* hi, void* lo, long n, void* stream) {
  if (!x || !hi || !lo) return V3A_ERR_ARG;
  if (n <= 0 || n % 8) return V3A_ERR_SHAPE;
  hipLaunchKernelGGL(split_f32_kernel, dim3(nblk(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (char*)hi, (char*)lo, n / 8);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_layernorm_pair(const float* x, void* y_hi, void* y_lo, const float* weight, const float* bias, int M, int d, int ldx,
                                  int ldy, float eps, int in_row_group, int in_row_skip, int in_row_off, void* stream) {
  if (!x || !y_hi || !y_lo) return V3A_ERR_ARG;
  if (M <= 0 || d <= 0 || d % 8 || ldx % 4 || ldy % 8 || d > 4 * 512) return V3A_ERR_SHAPE;
  LnPairP p{x, (char*)y_hi, (char*)y_lo, weight, bias, M, d, ldx, ldy, eps, in_row_group, in_row_skip, in_row_off};
  const dim3 grid((M + 3) / 4);
  const int cpl = (d / 8 + 63) / 64;
  if (cpl <= 1) hipLaunchKernelGGL(layernorm_pair_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (cpl <= 2) hipLaunchKernelGGL(layernorm_pair_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(layernorm_pair_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" int v3a_bilinear_cl_pair(const void* x_hi, const void* x_lo, void* y_hi, void* y_lo, const void* add_hi, const void* add_lo,
                                    const float* table, int T, int h, int w, int H, int W, int C, int align_corners, void* stream) {
  if (!x_hi || !x_lo || !y_hi || !y_lo || ((add_hi == nullptr) != (add_lo == nullptr))) return V3A_ERR_ARG;
  if (T <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return V3A_ERR_SHAPE;
  BilPairP p{(const char*)x_hi, (const char*)x_lo, (char*)y_hi, (char*)y_lo, (const char*)add_hi, (const char*)add_lo, table, T, h, w, H, W, C, align_corners};
  hipLaunchKernelGGL(bilinear_cl_pair_kernel, dim3(nblk((long)T * H * W * (C >> 3))), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
