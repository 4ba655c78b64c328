// Skinny bf16 GEMM for gfx950: one operand has <= 128 rows (a prompt's tokens through the UMT5 text encoder, time / camera
// embeddings), the other is a large weight matrix that is read exactly once — a pure weight-streaming problem.
//   C[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ residual[m][n]),  M <= 128          (or the transposed store C[n][m])
// replaces nn.Linear on <= 64 rows: transformers UMT5 q/k/v/o/wi_0/wi_1/wo (modeling_umt5.py) and the same small-M calls
// elsewhere.  The tile GEMM (gemm_bf16.hip) would put such a problem on N/128 <= 80 workgroups; here the weight rows are cut
// into 64-row strips x KSPLIT K-ranges so that >= 256 workgroups stream disjoint weight bytes:
//   workgroup = 4 waves, one 64-row weight strip, one K-range; wave w takes a quarter of the K-range and accumulates
//   D[n][m] (64 x 64, four 32x32x16 MFMA accumulators) from fragments loaded straight from global memory, 8 k-steps (128
//   contiguous bytes per weight row) in flight per batch; the four waves add up through LDS; K-range partials go to a fp32
//   scratch [KSPLIT][M][N] and a second tiny kernel sums them in fixed order (deterministic) and applies the epilogue.
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

struct SkP {
  const char* X; const char* W;   // bf16: X [M][ldx] (small), W [N][ldw] (streamed)
  float* part;                     // [ksplit][M][N] fp32
  int M, N, K, ldx, ldw;
  int kper;                        // K elements per workgroup (multiple of 256)
};

constexpr int SK_NB = 2;   // 64 weight rows per workgroup

// SK_MB 32-row blocks of X (64 or 128 rows, padded), SK_U k-steps per load batch (register budget: 2 x U x (NB+MB) x 4)
template <int SK_MB, int SK_U>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkP p) {
  constexpr int MP = 32 * SK_MB;
  extern __shared__ float red[];  // 4 waves x D[m][n]: 64 KB (MP = 64) or 128 KB (MP = 128)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int nstrips = (p.N + 63) / 64;
  const int strip = blockIdx.x % nstrips, ksp = blockIdx.x / nstrips;
  const int n0 = strip * 64;
  const int k0 = ksp * p.kper + wave * (p.kper / 4);
  const int kq = p.kper / 4;  // this wave's K extent (multiple of 64)

  const char* wp[SK_NB];
  const char* xp[SK_MB];
#pragma unroll
  for (int j = 0; j < SK_NB; ++j) {
    int n = n0 + 32 * j + l31;
    n = n < p.N ? n : p.N - 1;
    wp[j] = p.W + ((size_t)n * p.ldw + k0 + 8 * hi) * 2;
  }
#pragma unroll
  for (int i = 0; i < SK_MB; ++i) {
    int m = 32 * i + l31;
    m = m < p.M ? m : p.M - 1;
    xp[i] = p.X + ((size_t)m * p.ldx + k0 + 8 * hi) * 2;
  }
  f32x16 acc[SK_NB][SK_MB];
#pragma unroll
  for (int j = 0; j < SK_NB; ++j)
#pragma unroll
    for (int i = 0; i < SK_MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int nb = kq / (16 * SK_U);  // batches (kq is a multiple of 128 by construction)
  bf16x8 wf[2][SK_U][SK_NB], xf[2][SK_U][SK_MB];
  auto load = [&](int buf, int b) {
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
#pragma unroll
      for (int j = 0; j < SK_NB; ++j) wf[buf][u][j] = *(const bf16x8*)(wp[j] + (size_t)(b * SK_U + u) * 32);
#pragma unroll
      for (int i = 0; i < SK_MB; ++i) xf[buf][u][i] = *(const bf16x8*)(xp[i] + (size_t)(b * SK_U + u) * 32);
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int u = 0; u < SK_U; ++u)
#pragma unroll
      for (int j = 0; j < SK_NB; ++j)
#pragma unroll
        for (int i = 0; i < SK_MB; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[buf][u][j], xf[buf][u][i], acc[j][i], 0, 0, 0);
  };
  load(0, 0);
  for (int b = 0; b < nb; b += 2) {
    if (b + 1 < nb) load(1, b + 1);
    mma(0);
    if (b + 2 < nb) load(0, b + 2);
    if (b + 1 < nb) mma(1);
  }
  // D layout: lane (col m = l31, hi) register r <-> row n = (r & 3) + 8 (r >> 2) + 4 hi
  float* mine = red + wave * (MP * 64);
#pragma unroll
  for (int j = 0; j < SK_NB; ++j)
#pragma unroll
    for (int i = 0; i < SK_MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi, m = 32 * i + l31;
        mine[m * 64 + n] = acc[j][i][r];
      }
  __syncthreads();
  // MP*64 sums of four; partial layout [ksp][m][N]
  float* out = p.part + (size_t)ksp * p.M * p.N;
#pragma unroll
  for (int e = 0; e < MP * 64 / 256; ++e) {
    const int idx = e * 256 + tid;
    const int m = idx >> 6, n = idx & 63;
    const float v = (red[idx] + red[MP * 64 + idx]) + (red[2 * MP * 64 + idx] + red[3 * MP * 64 + idx]);
    if (m < p.M && n0 + n < p.N) out[(size_t)m * p.N + n0 + n] = v;
  }
}

struct SkF {
  const float* part; char* C; const float* bias; const char* res;
  int M, N, ksplit, ldc, ldr, act, flags, transposed;
};

__global__ __launch_bounds__(256) void gemm_skinny_finish_kernel(const SkF p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * p.N) return;
  const int m = (int)(idx / p.N), n = (int)(idx % p.N);
  float v = 0.f;
  for (int s = 0; s < p.ksplit; ++s) v += p.part[((size_t)s * p.M + m) * p.N + n];
  if (p.bias) v += p.bias[n];
  // rounding points of v3a_gemm_bf16_nt (bf16 autocast): the linear output is a bf16 value, the activation reads it and
  // produces a bf16 value, then the residual add
  v = round_bf16(v);
  if (p.act == V3A_ACT_GELU_TANH) v = round_bf16(gelu_tanh(v));
  else if (p.act == V3A_ACT_GELU_ERF) v = round_bf16(gelu_erf(v));
  else if (p.act == V3A_ACT_SILU) v = round_bf16(silu(v));
  else if (p.act == V3A_ACT_RELU) v = fmaxf(v, 0.f);
  const size_t orow = p.transposed ? (size_t)n : (size_t)m, ocol = p.transposed ? (size_t)m : (size_t)n;
  if (p.res) {
    const size_t ro = orow * p.ldr + ocol;
    v += (p.flags & V3A_GEMM_RES_F32) ? ((const float*)p.res)[ro] : bf16_to_f32(((const unsigned short*)p.res)[ro]);
  }
  const size_t co = orow * p.ldc + ocol;
  if (p.flags & V3A_GEMM_OUT_F32) ((float*)p.C)[co] = v;
  else ((unsigned short*)p.C)[co] = f32_to_bf16(v);
}

int pick_ksplit(int N, int K) {
  const int strips = (N + 63) / 64;
  int ks = 1;
  while (strips * ks < 256 && ks < 8 && K % (ks * 2 * 512) == 0) ks *= 2;  // each wave keeps >= 128 k per workgroup quarter
  return ks;
}

}  // namespace

extern "C" long v3a_gemm_skinny_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || M > 128 || N <= 0 || K <= 0 || K % 512) return V3A_ERR_SHAPE;
  return (long)pick_ksplit(N, K) * M * N * 4;
}

extern "C" int v3a_gemm_skinny_bf16(const v3a_gemm_skinny_args* a, void* stream) {
  if (!a || !a->X || !a->W || !a->C || !a->workspace) return V3A_ERR_ARG;
  if (a->M <= 0 || a->M > 128 || a->N <= 0 || a->K <= 0) return V3A_ERR_SHAPE;
  if (a->K % 512 || a->ldx % 8 || a->ldw % 8) return V3A_ERR_SHAPE;
  if (a->flags & ~(V3A_GEMM_RES_F32 | V3A_GEMM_OUT_F32)) return V3A_ERR_ARG;
  const int ks = pick_ksplit(a->N, a->K);
  if (a->workspace_bytes < (long)ks * a->M * a->N * 4) return V3A_ERR_WORKSPACE;
  SkP p = {(const char*)a->X, (const char*)a->W, (float*)a->workspace, a->M, a->N, a->K, a->ldx, a->ldw, a->K / ks};
  const int strips = (a->N + 63) / 64;
  if (a->M <= 64) {
    hipLaunchKernelGGL((gemm_skinny_kernel<2, 8>), dim3((unsigned)(strips * ks)), dim3(256), 4 * 64 * 64 * 4, (hipStream_t)stream, p);
  } else {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)gemm_skinny_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 64 * 4) != hipSuccess)
        return V3A_ERR_LAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<4, 4>), dim3((unsigned)(strips * ks)), dim3(256), 4 * 128 * 64 * 4, (hipStream_t)stream, p);
  }
  SkF f = {(const float*)a->workspace, (char*)a->C, a->bias, (const char*)a->residual, a->M, a->N, ks, a->ldc, a->ldr, a->act, a->flags,
           a->transposed_out};
  const long tot = (long)a->M * a->N;
  hipLaunchKernelGGL(gemm_skinny_finish_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
