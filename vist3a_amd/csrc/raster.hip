// 3D-Gaussian rasteriser for gfx950 — the step right after the text->3DGS path.
// Replaces gsplat==1.4.0 `rasterization(..., render_mode="RGB+D", packed=False, rasterize_mode="classic", covars=...)` as
// called one camera at a time by /root/reference/third_party_model/anysplat/src/model/decoder/decoder_splatting_cuda.py:96-125.
//
//   v3a_gs_project    one wave per 64 Gaussians, ALL C cameras of the batch: the 64 x 300-B SH block is staged once into
//                     LDS with coalesced 16-B loads (it is 75 % of the input bytes) and mean/covariance are held in
//                     registers while the wave loops over cameras: world->camera, EWA projection of the 3x3 covariance,
//                     +eps2d blur, conic, 3-sigma radius, near/far/screen culling, SH colour for survivors.
//   v3a_gs_rasterize  tile counts -> rocPRIM scan -> (camera | tile | depth bits) keys -> rocPRIM stable radix sort ->
//                     tile ranges -> one 256-lane workgroup per 16x16 tile of every camera (C x tiles workgroups fill the
//                     256 CUs where one camera's ~800 tiles cannot): Gaussians staged through LDS 256 at a time, read back
//                     as wave-uniform 16-B broadcasts, front-to-back compositing with workgroup-wide early exit.
// The reference renders one camera per gsplat call (decoder_splatting_cuda.py:96); batching cameras is the same arithmetic
// per camera and is what keeps an MI355X busy.
//
// HBM-bound integer/byte work up to the blend; the blend is VALU-bound (one v_exp_f32 per pixel-Gaussian pair).  Index work
// (tile bounds, keys, order, ranges) is bit-exact against oracle/gsplat_raster.py; pixel values agree to fp32 tolerance.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

constexpr int TILE = 16;

struct ProjP {
  const float* means; const float* covars; const float* sh;
  const float* viewmat; const float* campos; const float* K;
  int sh_layout, sh_k, sh_degree;
  int U, C, W, H;
  float near_plane, far_plane, radius_clip, eps2d;
  int* radii; float* means2d; float* depths; float* conics; float* colors;
};

template <int DEG>
__device__ __forceinline__ void sh_eval(float x, float y, float z, float* b) {
  b[0] = 0.2820947917738781f;
  if constexpr (DEG >= 1) { b[1] = -0.48860251190292f * y; b[2] = 0.48860251190292f * z; b[3] = -0.48860251190292f * x; }
  if constexpr (DEG >= 2) {
    const float z2 = z * z, t0b = -1.092548430592079f * z, c1 = x * x - y * y, s1 = 2.f * x * y;
    const float p6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[4] = 0.5462742152960395f * s1; b[5] = t0b * y; b[6] = p6; b[7] = t0b * x; b[8] = 0.5462742152960395f * c1;
    if constexpr (DEG >= 3) {
      const float t0c = -2.285228997322329f * z2 + 0.4570457994644658f, t1b = 1.445305721320277f * z;
      const float c2 = x * c1 - y * s1, s2 = x * s1 + y * c1;
      const float p12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
      b[9] = -0.5900435899266435f * s2; b[10] = t1b * s1; b[11] = t0c * y; b[12] = p12; b[13] = t0c * x; b[14] = t1b * c1;
      b[15] = -0.5900435899266435f * c2;
      if constexpr (DEG >= 4) {
        const float t0d = z * (-4.683325804901025f * z2 + 2.007139630671868f), t1c = 3.31161143515146f * z2 - 0.47308734787878f;
        const float t2b = -1.770130769779931f * z, c3 = x * c2 - y * s2, s3 = x * s2 + y * c2;
        b[16] = 0.6258357354491763f * s3; b[17] = t2b * s2; b[18] = t1c * s1; b[19] = t0d * y;
        b[20] = 1.984313483298443f * z * p12 - 1.006230589874905f * p6;
        b[21] = t0d * x; b[22] = t1c * c1; b[23] = t2b * c2; b[24] = 0.6258357354491763f * c3;
      }
    }
  }
}

// One wave = 64 consecutive Gaussians.  LDS image of their SH block: rows of 3*sh_k floats, lane l reads row l (odd row
// pitch for the production sh_k = 25 -> conflict-free ds_read_b32).
template <int DEG>
__global__ __launch_bounds__(64) void gs_project_kernel(ProjP p) {
  extern __shared__ float s_sh[];
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const int lane = threadIdx.x;
  const int g0 = blockIdx.x * 64;
  const int g = g0 + lane;
  const int row = 3 * p.sh_k;
  const int nrow = min(64, p.U - g0);
  {  // coalesced stage of nrow*row floats
    const float* src = p.sh + (long)g0 * row;
    const int total = nrow * row;
    if ((((unsigned long long)(uintptr_t)src) & 15) == 0) {
      const int n4 = total >> 2;
      for (int i = lane; i < n4; i += 64) *(f32x4*)(s_sh + 4 * i) = *(const f32x4*)(src + 4 * i);
      for (int i = (n4 << 2) + lane; i < total; i += 64) s_sh[i] = src[i];
    } else {
      for (int i = lane; i < total; i += 64) s_sh[i] = src[i];
    }
  }
  __syncthreads();
  if (g >= p.U) return;
  const float mx = p.means[3L * g], my = p.means[3L * g + 1], mz = p.means[3L * g + 2];
  const float* cv = p.covars + 9L * g;  // upper triangle of the symmetric world covariance
  const float s00 = cv[0], s01 = cv[1], s02 = cv[2], s11 = cv[4], s12 = cv[5], s22 = cv[8];
  const float* s = s_sh + lane * row;
  const float W = (float)p.W, H = (float)p.H;
  for (int c = 0; c < p.C; ++c) {
    const float* V = p.viewmat + 16 * c;
    const float R00 = V[0], R01 = V[1], R02 = V[2], R10 = V[4], R11 = V[5], R12 = V[6], R20 = V[8], R21 = V[9], R22 = V[10];
    const float x = R00 * mx + R01 * my + R02 * mz + V[3];
    const float y = R10 * mx + R11 * my + R12 * mz + V[7];
    const float z = R20 * mx + R21 * my + R22 * mz + V[11];
    const long o = (long)c * p.U + g;
    int radius_i = 0;
    float m2x = 0.f, m2y = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    if (z >= p.near_plane && z <= p.far_plane) {
      // M = R * S
      const float a00 = R00 * s00 + R01 * s01 + R02 * s02, a01 = R00 * s01 + R01 * s11 + R02 * s12, a02 = R00 * s02 + R01 * s12 + R02 * s22;
      const float a10 = R10 * s00 + R11 * s01 + R12 * s02, a11 = R10 * s01 + R11 * s11 + R12 * s12, a12 = R10 * s02 + R11 * s12 + R12 * s22;
      const float a20 = R20 * s00 + R21 * s01 + R22 * s02, a21 = R20 * s01 + R21 * s11 + R22 * s12, a22 = R20 * s02 + R21 * s12 + R22 * s22;
      // Cc = M * R^T (symmetric)
      const float c00 = a00 * R00 + a01 * R01 + a02 * R02, c01 = a00 * R10 + a01 * R11 + a02 * R12, c02 = a00 * R20 + a01 * R21 + a02 * R22;
      const float c11 = a10 * R10 + a11 * R11 + a12 * R12, c12 = a10 * R20 + a11 * R21 + a12 * R22;
      const float c22 = a20 * R20 + a21 * R21 + a22 * R22;
      const float* Kc = p.K + 9 * c;
      const float fx = Kc[0], fy = Kc[4], cx = Kc[2], cy = Kc[5];
      const float tfx = 0.5f * W / fx, tfy = 0.5f * H / fy;
      const float lxp = (W - cx) / fx + 0.3f * tfx, lxn = cx / fx + 0.3f * tfx;
      const float lyp = (H - cy) / fy + 0.3f * tfy, lyn = cy / fy + 0.3f * tfy;
      const float rz = 1.f / z, rz2 = rz * rz;
      const float tx = z * fminf(lxp, fmaxf(-lxn, x * rz)), ty = z * fminf(lyp, fmaxf(-lyn, y * rz));
      const float j00 = fx * rz, j02 = -fx * tx * rz2, j11 = fy * rz, j12 = -fy * ty * rz2;
      // cov2d = J Cc J^T, J = [[j00,0,j02],[0,j11,j12]]
      const float t00 = j00 * c00 + j02 * c02, t01 = j00 * c01 + j02 * c12, t02 = j00 * c02 + j02 * c22;
      const float t11 = j11 * c11 + j12 * c12, t12 = j11 * c12 + j12 * c22;
      const float q00 = t00 * j00 + t02 * j02 + p.eps2d;
      const float q01 = t01 * j11 + t02 * j12;
      const float q11 = t11 * j11 + t12 * j12 + p.eps2d;
      const float det = q00 * q11 - q01 * q01;
      m2x = fx * x * rz + cx;
      m2y = fy * y * rz + cy;
      if (det > 0.f) {
        const float id = 1.f / det;
        ca = q11 * id; cb = -q01 * id; cc = q00 * id;
        const float mid = 0.5f * (q00 + q11);
        const float v1 = mid + sqrtf(fmaxf(0.01f, mid * mid - det));
        const float radius = ceilf(3.f * sqrtf(v1));
        const bool off = (m2x + radius <= 0.f) || (m2x - radius >= W) || (m2y + radius <= 0.f) || (m2y - radius >= H);
        if (radius > p.radius_clip && !off) radius_i = (int)radius;
      }
    }
    p.radii[o] = radius_i;
    p.depths[o] = z;
    p.means2d[2 * o] = m2x; p.means2d[2 * o + 1] = m2y;
    p.conics[3 * o] = ca; p.conics[3 * o + 1] = cb; p.conics[3 * o + 2] = cc;
    float r = 0.f, gg = 0.f, b = 0.f;
    if (radius_i > 0) {  // SH colour only for Gaussians that will be drawn (gsplat's `masks = radii > 0`)
      float dx = mx - p.campos[3 * c], dy = my - p.campos[3 * c + 1], dz = mz - p.campos[3 * c + 2];
      const float n = sqrtf(dx * dx + dy * dy + dz * dz);
      const float inv = 1.f / fmaxf(n, 1e-20f);
      float bas[NB];
      sh_eval<DEG>(dx * inv, dy * inv, dz * inv, bas);
      if (p.sh_layout == 0) {  // [U, K, 3]
#pragma unroll
        for (int k = 0; k < NB; ++k) { r += bas[k] * s[3 * k]; gg += bas[k] * s[3 * k + 1]; b += bas[k] * s[3 * k + 2]; }
      } else {  // [U, 3, K]
#pragma unroll
        for (int k = 0; k < NB; ++k) { r += bas[k] * s[k]; gg += bas[k] * s[p.sh_k + k]; b += bas[k] * s[2 * p.sh_k + k]; }
      }
      r = fmaxf(r + 0.5f, 0.f); gg = fmaxf(gg + 0.5f, 0.f); b = fmaxf(b + 0.5f, 0.f);
    }
    f32x4 col = {r, gg, b, z};
    *(f32x4*)(p.colors + 4 * o) = col;
  }
}

// ---------------------------------------------------------------------------------------------- binning
struct BinP {
  const int* radii; const float* means2d; const float* depths;
  long CU;  // C * U entries
  int U, tw, th;
  unsigned int* counts; unsigned int* incl;
  unsigned long long* keys; unsigned int* vals;
  unsigned int* offs;
  unsigned int n;
  int nt_all;  // C * tiles
};

__device__ __forceinline__ void tile_bounds(float mx, float my, int radius, int tw, int th, int& x0, int& x1, int& y0, int& y1) {
  const float r = (float)radius / (float)TILE, tx = mx / (float)TILE, ty = my / (float)TILE;
  // (uint32)floor(negative) saturates to 0 on the device gsplat runs on: clamp to [0, grid]
  x0 = (int)fminf(fmaxf(floorf(tx - r), 0.f), (float)tw); x1 = (int)fminf(fmaxf(ceilf(tx + r), 0.f), (float)tw);
  y0 = (int)fminf(fmaxf(floorf(ty - r), 0.f), (float)th); y1 = (int)fminf(fmaxf(ceilf(ty + r), 0.f), (float)th);
}

__global__ __launch_bounds__(256) void gs_count_kernel(BinP p) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.CU) return;
  unsigned int n = 0;
  const int r = p.radii[e];
  if (r > 0) {
    int x0, x1, y0, y1;
    tile_bounds(p.means2d[2 * e], p.means2d[2 * e + 1], r, p.tw, p.th, x0, x1, y0, y1);
    n = (unsigned int)((x1 - x0) * (y1 - y0));
  }
  p.counts[e] = n;
}

__global__ __launch_bounds__(256) void gs_emit_kernel(BinP p) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.CU) return;
  const int r = p.radii[e];
  if (r <= 0) return;
  int x0, x1, y0, y1;
  tile_bounds(p.means2d[2 * e], p.means2d[2 * e + 1], r, p.tw, p.th, x0, x1, y0, y1);
  unsigned int o = p.incl[e] - p.counts[e];
  const unsigned long long cam_tile0 = (unsigned long long)(e / p.U) * (unsigned long long)(p.tw * p.th);
  const unsigned long long d = (unsigned long long)__float_as_uint(p.depths[e]);
  for (int i = y0; i < y1; ++i)
    for (int j = x0; j < x1; ++j) {
      p.keys[o] = ((cam_tile0 + (unsigned long long)(i * p.tw + j)) << 32) | d;
      p.vals[o] = (unsigned int)e;
      ++o;
    }
}

// offs[t] = first sorted intersection whose (camera, tile) id >= t, offs[C*ntiles] = n
__global__ __launch_bounds__(256) void gs_ranges_kernel(BinP p) {
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int t = (int)(p.keys[i] >> 32);
  const int prev = (i == 0) ? -1 : (int)(p.keys[i - 1] >> 32);
  for (int k = prev + 1; k <= t; ++k) p.offs[k] = i;
  if (i == p.n - 1)
    for (int k = t + 1; k <= p.nt_all; ++k) p.offs[k] = p.n;
}

// ---------------------------------------------------------------------------------------------- compositing
struct BlendP {
  const float* means2d; const float* conics; const float* colors; const float* opac;
  const unsigned int* offs; const unsigned int* ids;
  const float* bg;
  float* out_color; float* out_depth; float* out_alpha;
  int U, W, H, tw, ntiles, clamp_rgb;
};

__global__ __launch_bounds__(256) void gs_blend_kernel(BlendP p) {
  __shared__ f32x4 s_g[256 * 3];  // {x, y, opacity, -} {conic a, b, c, -} {r, g, b, depth}
  const int tid = threadIdx.x;
  const int cam = blockIdx.x / p.ntiles, tile = blockIdx.x - cam * p.ntiles;
  const int px_i = (tile % p.tw) * TILE + (tid & 15), py_i = (tile / p.tw) * TILE + (tid >> 4);
  const bool inside = px_i < p.W && py_i < p.H;
  const float px = (float)px_i + 0.5f, py = (float)py_i + 0.5f;
  const unsigned int s = p.offs[blockIdx.x], e = p.offs[blockIdx.x + 1];
  float T = 1.f, r = 0.f, g = 0.f, b = 0.f, dsum = 0.f;
  bool done = !inside;
  for (unsigned int base = s; base < e; base += 256) {
    if (__syncthreads_and(done)) break;  // also the barrier that protects s_g from the previous round's readers
    const unsigned int n = min(256u, e - base);
    if ((unsigned int)tid < n) {
      const unsigned int id = p.ids[base + tid];  // camera-major entry index c*U + g
      const unsigned int gi = id - (unsigned int)cam * (unsigned int)p.U;
      f32x4 a = {p.means2d[2L * id], p.means2d[2L * id + 1], p.opac[gi], 0.f};
      f32x4 c = {p.conics[3L * id], p.conics[3L * id + 1], p.conics[3L * id + 2], 0.f};
      s_g[3 * tid] = a; s_g[3 * tid + 1] = c; s_g[3 * tid + 2] = *(const f32x4*)(p.colors + 4L * id);
    }
    __syncthreads();
    if (!done) {
      for (unsigned int k = 0; k < n; ++k) {
        const f32x4 a = s_g[3 * k], c = s_g[3 * k + 1];
        const float dx = a[0] - px, dy = a[1] - py;
        const float sigma = 0.5f * (c[0] * dx * dx + c[2] * dy * dy) + c[1] * dx * dy;
        const float alpha = fminf(0.999f, a[2] * __expf(-sigma));
        if (sigma < 0.f || alpha < 1.f / 255.f) continue;
        const float nT = T * (1.f - alpha);
        if (nT <= 1e-4f) { done = true; break; }
        const float vis = alpha * T;
        const f32x4 col = s_g[3 * k + 2];
        r += col[0] * vis; g += col[1] * vis; b += col[2] * vis; dsum += col[3] * vis;
        T = nT;
      }
    }
  }
  if (!inside) return;
  if (p.bg) { r += T * p.bg[0]; g += T * p.bg[1]; b += T * p.bg[2]; }
  if (p.clamp_rgb) { r = fminf(fmaxf(r, 0.f), 1.f); g = fminf(fmaxf(g, 0.f), 1.f); b = fminf(fmaxf(b, 0.f), 1.f); }
  const long pix = ((long)cam * p.H + py_i) * p.W + px_i;
  p.out_color[3 * pix] = r; p.out_color[3 * pix + 1] = g; p.out_color[3 * pix + 2] = b;
  p.out_depth[pix] = dsum;
  p.out_alpha[pix] = 1.f - T;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RLayout { size_t counts, incl, keys0, keys1, vals0, vals1, offs, tmp, tmp_bytes, total; };

RLayout rlayout(long CU, long nt_all, long cap) {
  RLayout l{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += al256(b); return o; };
  l.counts = take(4 * (size_t)CU); l.incl = take(4 * (size_t)CU);
  l.keys0 = take(8 * (size_t)cap); l.keys1 = take(8 * (size_t)cap);
  l.vals0 = take(4 * (size_t)cap); l.vals1 = take(4 * (size_t)cap);
  l.offs = take(4 * (size_t)(nt_all + 1));
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned int*)nullptr,
                                  (unsigned int*)nullptr, (size_t)cap, 0, 64);
  (void)rocprim::inclusive_scan(nullptr, t2, (unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)CU, rocprim::plus<unsigned int>());
  l.tmp_bytes = t1 > t2 ? t1 : t2;
  l.tmp = take(l.tmp_bytes);
  l.total = off;
  return l;
}

bool dims_ok(long U, int C, int width, int height, long max_isect) {
  if (U <= 0 || C <= 0 || width <= 0 || height <= 0 || max_isect <= 0 || max_isect >= (1L << 31)) return false;
  const long nt = (long)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  return U * (long)C < (1L << 32) && nt * C < (1L << 24);
}

template <int DEG>
void launch_project(const ProjP& p, hipStream_t stream) {
  const size_t lds = (size_t)64 * 3 * p.sh_k * sizeof(float);
  hipLaunchKernelGGL(gs_project_kernel<DEG>, dim3((unsigned)((p.U + 63) / 64)), dim3(64), lds, stream, p);
}

}  // namespace

extern "C" int v3a_gs_project(const v3a_gs_project_args* a, void* stream) {
  if (!a || !a->means || !a->covars || !a->sh || !a->viewmat || !a->campos || !a->K || !a->radii || !a->means2d || !a->depths ||
      !a->conics || !a->colors)
    return V3A_ERR_ARG;
  if (a->U < 0 || a->C <= 0 || a->width <= 0 || a->height <= 0 || a->U >= (1L << 31)) return V3A_ERR_SHAPE;
  if (a->sh_degree < 0 || a->sh_degree > 4 || a->sh_k < (a->sh_degree + 1) * (a->sh_degree + 1) || a->sh_k > 64) return V3A_ERR_SHAPE;
  if (a->sh_layout != 0 && a->sh_layout != 1) return V3A_ERR_ARG;
  if (a->U == 0) return V3A_OK;
  ProjP p = {a->means, a->covars, a->sh, a->viewmat, a->campos, a->K, a->sh_layout, a->sh_k, a->sh_degree, (int)a->U, a->C, a->width,
             a->height, a->near_plane, a->far_plane, a->radius_clip, a->eps2d, a->radii, a->means2d, a->depths, a->conics, a->colors};
  hipStream_t st = (hipStream_t)stream;
  switch (a->sh_degree) {
    case 0: launch_project<0>(p, st); break;
    case 1: launch_project<1>(p, st); break;
    case 2: launch_project<2>(p, st); break;
    case 3: launch_project<3>(p, st); break;
    default: launch_project<4>(p, st); break;
  }
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" long v3a_gs_rasterize_workspace_bytes(long U, int C, int width, int height, long max_isect) {
  if (!dims_ok(U, C, width, height, max_isect)) return V3A_ERR_SHAPE;
  const long ntiles = (long)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  return (long)rlayout(U * C, ntiles * C, max_isect).total;
}

extern "C" int v3a_gs_rasterize(const v3a_gs_rasterize_args* a, void* stream_) {
  if (!a || !a->radii || !a->means2d || !a->depths || !a->conics || !a->colors || !a->opacities || !a->out_color || !a->out_depth ||
      !a->out_alpha || !a->workspace || !a->n_isect)
    return V3A_ERR_ARG;
  if (!dims_ok(a->U, a->C, a->width, a->height, a->max_isect)) return V3A_ERR_SHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  const int tw = (a->width + TILE - 1) / TILE, th = (a->height + TILE - 1) / TILE, ntiles = tw * th;
  const long CU = a->U * a->C;
  const int nt_all = ntiles * a->C;
  const RLayout l = rlayout(CU, nt_all, a->max_isect);
  if ((size_t)a->workspace_bytes < l.total) return V3A_ERR_SHAPE;
  char* ws = (char*)a->workspace;
  BinP b = {};
  b.radii = a->radii; b.means2d = a->means2d; b.depths = a->depths;
  b.CU = CU; b.U = (int)a->U; b.tw = tw; b.th = th; b.nt_all = nt_all;
  b.counts = (unsigned int*)(ws + l.counts); b.incl = (unsigned int*)(ws + l.incl);
  b.keys = (unsigned long long*)(ws + l.keys0); b.vals = (unsigned int*)(ws + l.vals0);
  b.offs = (unsigned int*)(ws + l.offs);
  const unsigned gb = (unsigned)((CU + 255) / 256);
  hipLaunchKernelGGL(gs_count_kernel, dim3(gb), dim3(256), 0, stream, b);
  size_t tb = l.tmp_bytes;
  if (rocprim::inclusive_scan(ws + l.tmp, tb, b.counts, b.incl, (size_t)CU, rocprim::plus<unsigned int>(), stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  // the intersection count sizes the sort: one 4-byte read-back per camera batch (gsplat does the same `.item()` per camera)
  unsigned int n = 0;
  if (hipMemcpyAsync(&n, b.incl + (CU - 1), 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return V3A_ERR_LAUNCH;
  if (hipStreamSynchronize(stream) != hipSuccess) return V3A_ERR_LAUNCH;
  *a->n_isect = (long)n;
  if ((long)n > a->max_isect) return V3A_ERR_WORKSPACE;
  if (hipMemsetAsync(b.offs, 0, 4 * (size_t)(nt_all + 1), stream) != hipSuccess) return V3A_ERR_LAUNCH;
  const unsigned int* ids = b.vals;
  if (n > 0) {
    hipLaunchKernelGGL(gs_emit_kernel, dim3(gb), dim3(256), 0, stream, b);
    int tile_bits = 1;
    while ((1L << tile_bits) < nt_all) ++tile_bits;
    tb = l.tmp_bytes;
    unsigned long long* k1 = (unsigned long long*)(ws + l.keys1);
    unsigned int* v1 = (unsigned int*)(ws + l.vals1);
    if (rocprim::radix_sort_pairs(ws + l.tmp, tb, b.keys, k1, b.vals, v1, (size_t)n, 0, 32 + tile_bits, stream) != hipSuccess)
      return V3A_ERR_LAUNCH;
    b.keys = k1; b.vals = v1; b.n = n;
    ids = v1;
    hipLaunchKernelGGL(gs_ranges_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, b);
  }
  if (a->tile_offsets_out &&
      hipMemcpyAsync(a->tile_offsets_out, b.offs, 4 * (size_t)(nt_all + 1), hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  if (a->flatten_ids_out && n > 0 &&
      hipMemcpyAsync(a->flatten_ids_out, ids, 4 * (size_t)n, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  BlendP bp = {a->means2d, a->conics, a->colors, a->opacities, b.offs, ids, a->background, a->out_color, a->out_depth, a->out_alpha,
               (int)a->U, a->width, a->height, tw, ntiles, a->clamp_rgb};
  hipLaunchKernelGGL(gs_blend_kernel, dim3((unsigned)nt_all), dim3(256), 0, stream, bp);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
