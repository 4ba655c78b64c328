// HBM-bound glue kernels of the stitched reconstruction path (SURVEY.md §8a R1,R6,R10,R11,R12,R14,R15) and the
// latency-bound fp32 camera-head primitives (R9).  All are 16-byte vectorised, one pass over their data.
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// q/k LayerNorm(head_dim=64, affine, eps) + 2-D RoPE, in place on the fused [q|k] projection buffer.
// vggt/layers/attention.py:56-61 (q_norm, k_norm, rope) + rope.py:154-188.  8 lanes own one head of one token
// (8 x 8 elements); the rotate-half partner chunk (j <-> j+16 inside each 32-wide y/x half) lives in lane^2.
struct QkP {
  char* x; int M, ld, C;
  const float* qw; const float* qb; const float* kw; const float* kb;
  const float* cs;   // [maxpos][16][2] (cos, sin); row 0 = identity; may be null (no rope)
  int rows_per_frame, n_special, n_valid, wp;
  float eps;
};

__global__ __launch_bounds__(256) void qknorm_rope2d_kernel(const QkP p) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int cpr = p.C >> 2;  // chunks per row over q and k: 2*C/8
  const long row = gid / cpr;
  if (row >= p.M) return;
  const int ci = (int)(gid - row * cpr);
  const int which = ci >= (p.C >> 3);
  const int c = ci & 7;
  char* ptr = p.x + ((size_t)row * p.ld + (size_t)ci * 8) * 2;
  float v[8];
  unpack_bf16x8(*(const u32x4*)ptr, v);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float t = v[e] - mean; q += t * t; }
  q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
  const float rstd = rsqrtf(q * (1.f / 64.f) + p.eps);
  const float* w = which ? p.kw : p.qw;
  const float* b = which ? p.kb : p.qb;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * w[c * 8 + e] + b[c * 8 + e];
  if (p.cs) {
    const int pr = (int)(row % p.rows_per_frame);
    int pos = 0;
    if (pr >= p.n_special && pr < p.n_valid) {
      const int patch = pr - p.n_special;
      pos = ((c >> 2) == 0 ? patch / p.wp : patch % p.wp) + 1;
    }
    const float* t = p.cs + ((size_t)pos * 16 + (c & 1) * 8) * 2;
    const bool lo = (c & 2) == 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = __shfl_xor(v[e], 2, 64);
      const float co = t[2 * e], si = t[2 * e + 1];
      v[e] = lo ? v[e] * co - o * si : v[e] * co + o * si;
    }
  }
  *(u32x4*)ptr = pack_bf16x8(v);
}

// ------------------------------------------------------------------------------------------------
// VAE latent [C][Tl][H][W] f32 -> channels-last bf16 [4(Tl-1)+1][H][W][C] with the align_corners=True temporal
// lerp of stitched_model.py:92-107 (weights j/4; H,W untouched).
struct UpTP { const float* z; char* y; int C, Tl, HW; };
__global__ __launch_bounds__(256) void latent_upT_cl_kernel(const UpTP p) {
  const int T = (p.Tl - 1) * 4 + 1;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int cch = p.C >> 3;
  const long pix = gid / cch;
  if (pix >= (long)T * p.HW) return;
  const int c0 = (int)(gid - pix * cch) * 8;
  const int t = (int)(pix / p.HW), hw = (int)(pix - (long)t * p.HW);
  // torch: src = t * (Tl-1)/(T-1); i0 = floor(src); l1 = src - i0; out = (1-l1)*x0 + l1*x1
  const float scale = T > 1 ? (float)(p.Tl - 1) / (float)(T - 1) : 0.f;
  const float src = scale * (float)t;
  const int i0 = (int)src;
  const int i1 = i0 + (i0 < p.Tl - 1 ? 1 : 0);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = p.z[((size_t)(c0 + e) * p.Tl + i0) * p.HW + hw];
    const float b = p.z[((size_t)(c0 + e) * p.Tl + i1) * p.HW + hw];
    o[e] = l0 * a + l1 * b;
  }
  *(u32x4*)(p.y + ((size_t)pix * p.C + c0) * 2) = pack_bf16x8(o);
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize of a channels-last bf16 image stack [T][h][w][C] -> [T][H][W][C] (+ optional bf16 addend, + optional
// f32 table broadcast over T, + optional ReLU).  F.interpolate(mode="bilinear") with align_corners True
// (dpt_head.py:460-466, vggt_dpt_gs_head.py:166) or False (inference_t23d.py:118-123, T unchanged).
struct BilP {
  const char* x; char* y; const char* add; const float* tab;
  int T, h, w, H, W, C, align, relu, out_f32;
};
__global__ __launch_bounds__(256) void bilinear_cl_kernel(const BilP p) {
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
  const char* base = p.x + (size_t)t * p.h * p.w * p.C * 2 + c0 * 2;
  float a[8], b[8], c[8], d[8], o[8];
  unpack_bf16x8(*(const u32x4*)(base + ((size_t)y0 * p.w + x0) * p.C * 2), a);
  unpack_bf16x8(*(const u32x4*)(base + ((size_t)y0 * p.w + x1) * p.C * 2), b);
  unpack_bf16x8(*(const u32x4*)(base + ((size_t)y1 * p.w + x0) * p.C * 2), c);
  unpack_bf16x8(*(const u32x4*)(base + ((size_t)y1 * p.w + x1) * p.C * 2), d);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
  if (p.add) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(p.add + ((size_t)pix * p.C + c0) * 2), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += f[e];
  }
  if (p.tab) {
    const float* tp = p.tab + (size_t)rem * p.C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += tp[e];
  }
  if (p.relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
  }
  if (p.out_f32) {
    float* yp = (float*)p.y + (size_t)pix * p.C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) yp[e] = o[e];
  } else {
    *(u32x4*)(p.y + ((size_t)pix * p.C + c0) * 2) = pack_bf16x8(o);
  }
}

// ------------------------------------------------------------------------------------------------
// depth head activation (exp / 1+exp, head_act.py:61-112) fused with the unprojection to world points
// (geometry.py:10-58).  cam[frame] = {fx, fy, cx, cy, Rt[9] (= R^T row-major), tinv[3] (= -R^T t)}.
struct UnpP { const float* raw; int ld; const float* cam; float* depth; float* conf; float* pts; int S, H, W; };
__global__ __launch_bounds__(256) void depth_unproject_kernel(const UnpP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long hw = (long)p.H * p.W;
  if (i >= hw * p.S) return;
  const int f = (int)(i / hw);
  const int rem = (int)(i - f * hw);
  const int v = rem / p.W, u = rem - v * p.W;
  const float* c = p.cam + f * 16;
  const float d = expf(p.raw[(size_t)i * p.ld]);
  const float cf = 1.f + expf(p.raw[(size_t)i * p.ld + 1]);
  const float x = ((float)u - c[2]) * d / c[0];
  const float y = ((float)v - c[3]) * d / c[1];
  p.depth[i] = d;
  p.conf[i] = cf;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    p.pts[i * 3 + r] = ((c[4 + 3 * r] * x + c[5 + 3 * r] * y) + c[6 + 3 * r] * d) + c[13 + r] * 1.0f;
}

// ------------------------------------------------------------------------------------------------
// UnifiedGaussianAdapter.forward (gaussian_adapter.py:114-147) + opacity map (anysplat.py:225-238, exponent 2^x) +
// build_covariance (gaussians.py:33-44).  feats[u] = {density logit, 3 scale, 4 quat (xyzw), 3*dsh harmonics}.
struct AdP {
  const float* pts; const float* feats; int ldf; long U; int dsh; float op_exp;
  const float* shmask;
  float* means; float* cov; float* sh; float* opac; float* scales; float* rot;
};
// One 16-LANE GROUP per Gaussian: the 83-float input row and the 75-float SH output row are moved as coalesced 64-byte pieces
// (a wave's four Gaussians are adjacent rows, i.e. one contiguous 1.2 KB output span); the 8 leading scalars are broadcast inside
// the group and the tiny per-Gaussian algebra (opacity, softplus scales, quaternion -> R, covariance) is done redundantly by every
// lane.  The thread-per-Gaussian form read and wrote rows at a 300-byte stride per lane: 10.7 ms for 2.6 M Gaussians.
__global__ __launch_bounds__(256) void gaussian_adapter_kernel(const AdP p) {
  const int lane = threadIdx.x & 63, sl = lane & 15, gbase = lane & 48;
  const long u = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (u >= p.U) return;   // whole groups leave together
  const float* f = p.feats + (size_t)u * p.ldf;
  const float head = sl < 8 ? f[sl] : 0.f;
  float h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = __shfl(head, gbase + e, 64);
  if (sl < 3) p.means[u * 3 + sl] = p.pts[u * 3 + sl];
  const float pd = 1.f / (1.f + expf(-h[0]));
  const float ex = p.op_exp;
  if (sl == 0) p.opac[u] = 0.5f * (1.f - (ex == 1.f ? (1.f - pd) : powf(1.f - pd, ex)) + (ex == 1.f ? pd : powf(pd, 1.f / ex)));
  float s[3], q[4];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float x = h[1 + e];
    const float sp = x > 20.f ? x : log1pf(expf(x));  // F.softplus (beta=1, threshold=20)
    s[e] = fminf(0.001f * sp, 0.3f);
  }
  float n2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { q[e] = h[4 + e]; n2 += q[e] * q[e]; }
  const float inv = 1.f / (sqrtf(n2) + 1e-8f);
#pragma unroll
  for (int e = 0; e < 4; ++e) q[e] *= inv;
  const float i = q[0], j = q[1], k = q[2], r = q[3];
  const float two_s = 2.0f / (i * i + j * j + k * k + r * r);
  const float R[9] = {1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                      two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                      two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)};
  // cov = R diag(s) diag(s)^T R^T; lane sl < 9 stores element sl
  float mine = 0.f, sc = 0.f, rt = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 3; ++m) acc += (R[a * 3 + m] * s[m]) * s[m] * R[b * 3 + m];
      if (sl == a * 3 + b) mine = acc;
    }
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (sl == e) sc = s[e];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (sl == e) rt = q[e];
  if (sl < 9) p.cov[u * 9 + sl] = mine;
  if (sl < 3) p.scales[u * 3 + sl] = sc;
  if (sl < 4) p.rot[u * 4 + sl] = rt;
  const int nsh = 3 * p.dsh;
  for (int e = sl; e < nsh; e += 16) p.sh[(size_t)u * nsh + e] = f[8 + e] * p.shmask[e % p.dsh];
}

// ------------------------------------------------------------------------------------------------
// fp32 "skinny" linear for the camera head (13-21 rows): y[M][N] = epi(x[M][K] . W[N][K]^T).  One wave per output
// column; the weight row is streamed once (the op is weight-bandwidth bound), x comes from L2.
struct LinP {
  const float* x; const float* w; const float* b; float* y; const float* res; const float* gamma;
  int M, N, K, ldx, ldy, ldr, act;
};
// One wave per NPW consecutive outputs: every x fragment is used against NPW weight rows (with one output per wave the 13 activation
// loads per 16-byte weight load, not the weight stream, set the pace: 0.7 TB/s).  The activations are staged per workgroup in LDS and read
// from there by its four waves (round 4; before, every wave re-read all of x through L1 / L2: 326 MB of L2 traffic beside the 50 MB qkv
// weight).  Late round 4: K is walked in chunks of 256 (one 16-byte weight load per lane and row); the chunk's activations arrive by 16-byte
// LDS-DMA in a double buffer of only 2 x M KB - the first form's 52 KB chunk limited a CU to two or three workgroups, and what this kernel
// needs is loads in flight - and the weights of chunk c + 1 are requested before the FMAs of chunk c.  The host picks NPW so that the
// workgroups fill whole rounds of 256 CUs (N = 6144: 3 -> 512 workgroups instead of 384; N = 2048: 2 -> 256 instead of 128).  The sum of
// an output runs over k in index order within a lane (explicit FMAs: the same bits from every instantiation), then across the lanes.
template <int MAXM, int NPW>
__global__ __launch_bounds__(256) void linear_f32_kernel(const LinP p) {
  constexpr int KC = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x [M][KC] floats
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = (blockIdx.x * 4 + wave) * NPW;   // (a wave past N still takes part in the staging and the barriers)
  float acc[NPW][MAXM];
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int m = 0; m < MAXM; ++m) acc[j][m] = 0.f;
  const float* wr[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) wr[j] = p.w + (size_t)min(n0 + j, p.N - 1) * p.K + lane * 4;   // (rows past N are computed and dropped)
  const int nchunks = (p.K + KC - 1) / KC;
  auto stage = [&](int c) {   // one 1-KiB piece per row; lanes past the end of K re-read the row's last float4 (their slots are never read)
    const int k0 = c * KC, kc = min(KC, p.K - k0);     // K % 4 == 0
    float* sb = (float*)smem + (size_t)(c & 1) * p.M * KC;
    for (int m = wave; m < p.M; m += 4) glds16(p.x + (size_t)m * p.ldx + k0 + min(lane * 4, kc - 4), sb + m * KC);
  };
  auto loadw = [&](int c, f32x4 (&wv)[NPW]) {
    if (c * KC + lane * 4 < p.K) {
#pragma unroll
      for (int j = 0; j < NPW; ++j) wv[j] = *(const f32x4*)(wr[j] + c * KC);
    }
  };
  f32x4 wv[NPW], wn[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) wv[j] = wn[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  stage(0);
  loadw(0, wv);
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // chunk c has landed; everyone is done with the other buffer
    if (c + 1 < nchunks) {
      stage(c + 1);
      loadw(c + 1, wn);
    }
    if (c * KC + lane * 4 < p.K) {
      const float* sx = (const float*)smem + (size_t)(c & 1) * p.M * KC + lane * 4;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < p.M) {
          const f32x4 xv = *(const f32x4*)(sx + m * KC);
#pragma unroll
          for (int j = 0; j < NPW; ++j)
            acc[j][m] = __builtin_fmaf(wv[j][3], xv[3], __builtin_fmaf(wv[j][2], xv[2], __builtin_fmaf(wv[j][1], xv[1], __builtin_fmaf(wv[j][0], xv[0], acc[j][m]))));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j) wv[j] = wn[j];
  }
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int n = n0 + j;
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const float t = wave_sum(acc[j][m]);
      if (lane == m) mine = t;
    }
    if (n < p.N && lane < p.M) {
      float v = mine + (p.b ? p.b[n] : 0.f);
      if (p.act == V3A_ACT_GELU_ERF) v = gelu_erf(v);
      else if (p.act == V3A_ACT_SILU) v = silu(v);
      else if (p.act == V3A_ACT_RELU) v = fmaxf(v, 0.f);
      if (p.gamma) v *= p.gamma[n];
      if (p.res) v += p.res[(size_t)lane * p.ldr + n];
      p.y[(size_t)lane * p.ldy + n] = v;
    }
  }
}

template <int MAXM, int NPW>
int launch_linear_f32(const LinP& p, void* stream) {
  hipLaunchKernelGGL((linear_f32_kernel<MAXM, NPW>), dim3((unsigned)((p.N + 4 * NPW - 1) / (4 * NPW))), dim3(256), (size_t)2 * p.M * 256 * 4, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

// fp32 attention over a handful of tokens (camera-head trunk: S views x heads, vggt/layers/attention.py:49-80 without
// qk-norm / rope).  qkv [S][3*C]; one workgroup per head.
struct SmAtP { const float* qkv; float* o; int S, H, hd; float scale; };
__global__ __launch_bounds__(64) void attention_small_f32_kernel(const SmAtP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = (float*)smem;
  float* sk = sq + p.S * p.hd;
  float* sv = sk + p.S * p.hd;
  float* sp = sv + p.S * p.hd;  // [S]
  const int h = blockIdx.x, lane = threadIdx.x, C = p.H * p.hd;
  for (int i = lane; i < p.S * p.hd; i += 64) {
    const int s = i / p.hd, d = i - s * p.hd;
    const float* r = p.qkv + (size_t)s * 3 * C + h * p.hd + d;
    sq[i] = r[0]; sk[i] = r[C]; sv[i] = r[2 * C];
  }
  __syncthreads();
  for (int i = 0; i < p.S; ++i) {
    float sc = -1e30f;
    if (lane < p.S) {
      float a = 0.f;
      for (int d = 0; d < p.hd; ++d) a += sq[i * p.hd + d] * sk[lane * p.hd + d];
      sc = a * p.scale;
    }
    const float mx = wave_max(sc);
    const float e = lane < p.S ? expf(sc - mx) : 0.f;
    const float den = wave_sum(e);
    if (lane < p.S) sp[lane] = e / den;
    __syncthreads();
    for (int d = lane; d < p.hd; d += 64) {
      float a = 0.f;
      for (int j = 0; j < p.S; ++j) a += sp[j] * sv[j * p.hd + d];
      p.o[(size_t)i * C + h * p.hd + d] = a;
    }
    __syncthreads();
  }
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH)

extern "C" int v3a_qknorm_rope2d(void* qk, int M, int ld, int C, const float* qw, const float* qb, const float* kw,
                                 const float* kb, const float* cos_sin, int rows_per_frame, int n_special, int n_valid,
                                 int wp, float eps, void* stream) {
  if (!qk || !qw || !qb || !kw || !kb) return V3A_ERR_ARG;
  if (M <= 0 || C <= 0 || C % 64 || ld % 8 || ld < 2 * C) return V3A_ERR_SHAPE;
  if (cos_sin && (rows_per_frame <= 0 || wp <= 0)) return V3A_ERR_ARG;
  QkP p{(char*)qk, M, ld, C, qw, qb, kw, kb, cos_sin, rows_per_frame > 0 ? rows_per_frame : M, n_special, n_valid, wp > 0 ? wp : 1, eps};
  hipLaunchKernelGGL(qknorm_rope2d_kernel, dim3(nblk((long)M * (C >> 2))), dim3(256), 0, (hipStream_t)stream, p);
  return LAUNCH_OK();
}

extern "C" int v3a_latent_upsample_t_cl(const float* z, void* y, int C, int Tl, int H, int W, void* stream) {
  if (!z || !y) return V3A_ERR_ARG;
  if (C <= 0 || C % 8 || Tl <= 0 || H <= 0 || W <= 0) return V3A_ERR_SHAPE;
  UpTP p{z, (char*)y, C, Tl, H * W};
  const long n = (long)((Tl - 1) * 4 + 1) * H * W * (C >> 3);
  hipLaunchKernelGGL(latent_upT_cl_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, p);
  return LAUNCH_OK();
}

extern "C" int v3a_bilinear_cl(const void* x, void* y, const void* add, const float* table, int T, int h, int w, int H,
                               int W, int C, int align_corners, int relu, int out_f32, void* stream) {
  if (!x || !y) return V3A_ERR_ARG;
  if (T <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return V3A_ERR_SHAPE;
  BilP p{(const char*)x, (char*)y, (const char*)add, table, T, h, w, H, W, C, align_corners, relu, out_f32};
  hipLaunchKernelGGL(bilinear_cl_kernel, dim3(nblk((long)T * H * W * (C >> 3))), dim3(256), 0, (hipStream_t)stream, p);
  return LAUNCH_OK();
}

extern "C" int v3a_depth_unproject(const float* raw, int ld, const float* cam, float* depth, float* conf, float* pts,
                                   int S, int H, int W, void* stream) {
  if (!raw || !cam || !depth || !conf || !pts) return V3A_ERR_ARG;
  if (S <= 0 || H <= 0 || W <= 0 || ld < 2) return V3A_ERR_SHAPE;
  UnpP p{raw, ld, cam, depth, conf, pts, S, H, W};
  hipLaunchKernelGGL(depth_unproject_kernel, dim3(nblk((long)S * H * W)), dim3(256), 0, (hipStream_t)stream, p);
  return LAUNCH_OK();
}

extern "C" int v3a_gaussian_adapter(const float* pts, const float* feats, int ldf, long U, int sh_degree, float opacity_exponent,
                                    const float* sh_mask, float* means, float* cov, float* sh, float* opac, float* scales,
                                    float* rot, void* stream) {
  if (!pts || !feats || !sh_mask || !means || !cov || !sh || !opac || !scales || !rot) return V3A_ERR_ARG;
  const int dsh = (sh_degree + 1) * (sh_degree + 1);
  if (U <= 0 || ldf < 8 + 3 * dsh) return V3A_ERR_SHAPE;
  AdP p{pts, feats, ldf, U, dsh, opacity_exponent, sh_mask, means, cov, sh, opac, scales, rot};
  hipLaunchKernelGGL(gaussian_adapter_kernel, dim3((unsigned)((U * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return LAUNCH_OK();
}

extern "C" int v3a_linear_f32(const float* x, const float* w, const float* bias, float* y, const float* residual,
                              const float* gamma, int M, int N, int K, int ldx, int ldy, int ldr, int act, void* stream) {
  if (!x || !w || !y) return V3A_ERR_ARG;
  if (M <= 0 || M > 32 || N <= 0 || K <= 0 || K % 4 || ldx % 4) return V3A_ERR_SHAPE;
  LinP p{x, w, bias, y, residual, gamma, M, N, K, ldx, ldy, ldr, act};
  // outputs per wave: the count whose workgroups (4 waves) come closest to whole rounds of 256 CUs; ties go to the larger count (more
  // reuse of every staged activation)
  const int maxnpw = M <= 16 ? 4 : 2;
  int best = 1;
  double beff = -1.0;
  for (int npw = 1; npw <= maxnpw; ++npw) {
    const long wgs = (N + 4 * npw - 1) / (4 * npw);
    const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    if (eff >= beff - 1e-9) { beff = eff; best = npw; }
  }
  if (M <= 16) {
    switch (best) {
      case 4: return launch_linear_f32<16, 4>(p, stream);
      case 3: return launch_linear_f32<16, 3>(p, stream);
      case 2: return launch_linear_f32<16, 2>(p, stream);
      default: return launch_linear_f32<16, 1>(p, stream);
    }
  }
  return best == 2 ? launch_linear_f32<32, 2>(p, stream) : launch_linear_f32<32, 1>(p, stream);
}

extern "C" int v3a_attention_small_f32(const float* qkv, float* out, int S, int H, int hd, float scale, void* stream) {
  if (!qkv || !out) return V3A_ERR_ARG;
  if (S <= 0 || S > 64 || H <= 0 || hd <= 0) return V3A_ERR_SHAPE;
  const size_t lds = ((size_t)3 * S * hd + 64) * 4;
  if (lds > 64 * 1024) return V3A_ERR_SHAPE;
  SmAtP p{qkv, out, S, H, hd, scale};
  hipLaunchKernelGGL(attention_small_f32_kernel, dim3(H), dim3(64), lds, (hipStream_t)stream, p);
  return LAUNCH_OK();
}
