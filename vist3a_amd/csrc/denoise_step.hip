// The per-step glue of the denoise loop as ONE launch: what the reference does with ~25 tiny
// PyTorch kernels between two DiT forwards (/root/reference/inference_t23d.py:94-103 -> diffusers 0.33.1 WanPipeline.__call__ and
// UniPCMultistepScheduler.step, restated in vist3a_amd/wan/pipeline.py / scheduler.py):
//     noise   = unpatchify(DiT output tokens)                                         (a permute + copy)
//     noise   = n_u + g (n_c - n_u)                                                   classifier-free guidance, bf16 arithmetic
//     m       = sample - sigma * noise                                                flow prediction -> x0
//     sample' = corrector(last_sample, m_prev, [m_prev2], m)                          UniC, from the second step on
//     prev    = predictor(sample', m, [m_prev])                                       UniP (order 1 or 2)
//     tokens  = patchify(bf16(prev))   for both CFG batch items                       the next forward's input
// Every intermediate is rounded exactly where the PyTorch ops round it (bf16 for the guidance arithmetic and sigma * noise, fp32 for
// every scheduler product / difference - separate multiplies and subtractions, NO fused multiply-add, division by the UniPC ratio as
// multiplication by its fp32 reciprocal the way ATen's CUDA div-by-scalar does), so the latents are bit-identical to the tensor-op
// loop: tests/test_boundary_gpu.py::test_fused_denoise_loop_is_bit_identical_to_tensor_ops / test_fused_step_kernel_matches_tensor_ops_for_every_order.
// Layouts: latents / history [16][T][H][W] fp32; DiT output tokens [B N][(ph pw) c] bf16 (proj_out order), DiT input tokens
// [B N][c (ph pw)] bf16 (patch_embedding order), N = T (H/2) (W/2) tokens in (t, h, w) order, patch (1, 2, 2).
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

// Separately rounded fp32 operations.  HIP's __fmul_rn / __fsub_rn are plain `a * b` / `a - b` in the headers and hipcc contracts
// them into FMAs (-ffp-contract=fast; `#pragma clang fp contract(off)` did not stop the backend either): every result passes through
// an empty asm, which the optimiser cannot fuse across.
__device__ __forceinline__ float opaque(float r) { asm("" : "+v"(r)); return r; }
__device__ __forceinline__ float mul(float a, float b) { return opaque(a * b); }
__device__ __forceinline__ float sub(float a, float b) { return opaque(a - b); }
__device__ __forceinline__ float add(float a, float b) { return opaque(a + b); }

__global__ __launch_bounds__(256) void unipc_cfg_step_kernel(const v3a_unipc_step_args a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const int HW = a.H * a.W;
  const long total = (long)a.C * a.T * HW;
  if (e >= total) return;
  const int w = (int)(e % a.W), h = (int)(e / a.W % a.H), t = (int)(e / HW % a.T), c = (int)(e / ((long)HW * a.T));
  const int N = a.T * (a.H / 2) * (a.W / 2);
  const long n = ((long)t * (a.H / 2) + h / 2) * (a.W / 2) + w / 2;
  const int sp = (h & 1) * 2 + (w & 1);            // position inside the 2 x 2 patch
  const int ld_out = 4 * a.C, ld_tok = 4 * a.C;
  const unsigned short* out = (const unsigned short*)a.dit_out;
  // ---- unpatchify + classifier-free guidance (bf16 tensor arithmetic: every op rounds to bf16) ----
  float noise = bf16_to_f32(out[n * ld_out + sp * a.C + c]);
  if (a.guided) {
    const float nu = bf16_to_f32(out[(N + n) * ld_out + sp * a.C + c]);
    const float d = round_bf16(sub(noise, nu));
    const float g = round_bf16(mul(d, a.guidance));
    noise = round_bf16(add(nu, g));
  }
  // ---- x0 prediction: sample - bf16(sigma * noise) ----
  const float sample = a.sample[e];
  const float m = sub(sample, round_bf16(mul(noise, a.sigma)));
  const float m1 = a.m_prev1 ? a.m_prev1[e] : 0.f;     // model_outputs[-1] before this step
  const float m2 = a.m_prev2 ? a.m_prev2[e] : 0.f;     // model_outputs[-2] before this step
  // ---- UniC corrector on the previous sample ----
  float x = sample;
  if (a.corr_order > 0) {
    float xt = sub(mul(a.last_sample[e], a.cc1), mul(m1, a.cc2));
    float corr = 0.f;
    if (a.corr_order == 2) corr = add(0.f, mul(mul(sub(m2, m1), a.c_inv_rk), a.c_rho0));
    xt = sub(xt, mul(add(corr, mul(sub(m, m1), a.c_rho_last)), a.cc3));
    x = xt;
  }
  // ---- UniP predictor from the corrected sample ----
  float p = sub(mul(x, a.pc1), mul(m, a.pc2));
  if (a.pred_order == 2) {
    const float pred = add(0.f, mul(mul(sub(m1, m), a.p_inv_rk), a.p_rho0));
    p = sub(p, mul(pred, a.pc3));
  }
  a.m_out[e] = m;
  a.sample_corrected[e] = x;
  a.prev[e] = p;
  // ---- next forward's input tokens: bf16(prev), patchified, one copy per CFG batch item ----
  if (a.tok) {
    const unsigned short pb = f32_to_bf16(p);
    unsigned short* tok = (unsigned short*)a.tok;
    for (int b = 0; b < a.batch; ++b) tok[((long)b * N + n) * ld_tok + c * 4 + sp] = pb;
  }
}

}  // namespace

extern "C" int v3a_unipc_cfg_step(const v3a_unipc_step_args* a, void* stream) {
  if (!a || !a->dit_out || !a->sample || !a->m_out || !a->sample_corrected || !a->prev) return V3A_ERR_ARG;
  if (a->C <= 0 || a->T <= 0 || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1)) return V3A_ERR_SHAPE;
  if (a->corr_order < 0 || a->corr_order > 2 || a->pred_order < 1 || a->pred_order > 2) return V3A_ERR_ARG;
  if (a->corr_order > 0 && (!a->last_sample || !a->m_prev1)) return V3A_ERR_ARG;
  if ((a->corr_order == 2 && !a->m_prev2) || (a->pred_order == 2 && !a->m_prev1)) return V3A_ERR_ARG;
  // dit_out holds `batch` row blocks of N tokens: the guided form reads rows [0, 2N) whether or not tok is written
  if (a->batch < 1 || a->batch > 2 || (a->guided && a->batch != 2)) return V3A_ERR_ARG;
  const long total = (long)a->C * a->T * a->H * a->W;
  hipLaunchKernelGGL(unipc_cfg_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
