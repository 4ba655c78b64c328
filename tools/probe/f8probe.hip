// Probe of the gfx950 block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, fp8 e4m3 operands, unit scales): which (row, k)
// each byte of a lane's 8 operand VGPRs feeds.  Host packs A/B under a candidate layout; see tools/probe/f8probe.py.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(16))) float v16f;
__global__ void k_mfma(const int* a, const int* b, float* c, int scale) {
  v8i A, B;
  for (int i = 0; i < 8; ++i) { A[i] = a[threadIdx.x * 8 + i]; B[i] = b[threadIdx.x * 8 + i]; }
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, scale, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = acc[i];
}
__global__ void k_cvt(const float* x, int* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], 0, false);
    y[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], lo, true);
  }
}
extern "C" void f8_mfma(const int* a, const int* b, float* c, int scale, void* s) { hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, (hipStream_t)s, a, b, c, scale); }
extern "C" void f8_cvt(const float* x, int* y, int n, void* s) { hipLaunchKernelGGL(k_cvt, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, x, y, n); }
