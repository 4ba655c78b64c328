// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD, by operand register file and dependency pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_forms.hip -o /tmp/mfma_forms && /tmp/mfma_forms
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define M_VVV(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define M_VVA(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
#define M_AVV(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))
#define M_AVA(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "a"(b))
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {2, 3, 4, 5, 6, 7, 8, (short)threadIdx.x};
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) { M_VVV(c0, a, b); M_VVV(c1, a, b); M_VVV(c2, a, b); M_VVV(c3, a, b); M_VVV(c0, a, b); M_VVV(c1, a, b); M_VVV(c2, a, b); M_VVV(c3, a, b); }
    if constexpr (MODE == 1) { M_VVA(c0, a, b); M_VVA(c1, a, b); M_VVA(c2, a, b); M_VVA(c3, a, b); M_VVA(c0, a, b); M_VVA(c1, a, b); M_VVA(c2, a, b); M_VVA(c3, a, b); }
    if constexpr (MODE == 2) { M_AVV(c0, a, b); M_AVV(c1, a, b); M_AVV(c2, a, b); M_AVV(c3, a, b); M_AVV(c0, a, b); M_AVV(c1, a, b); M_AVV(c2, a, b); M_AVV(c3, a, b); }
    if constexpr (MODE == 3) { M_VVV(c0, a, b); M_VVV(c1, a, b); M_VVV(c0, a, b); M_VVV(c1, a, b); M_VVV(c0, a, b); M_VVV(c1, a, b); M_VVV(c0, a, b); M_VVV(c1, a, b); }   // two accumulators
    if constexpr (MODE == 4) { M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); M_VVV(c0, a, b); }   // one chain
    if constexpr (MODE == 5) { M_AVV(c0, a, b); M_AVV(c1, a, b); M_AVV(c2, a, b); M_AVV(c3, a, b); M_AVV(c4, a, b); M_AVV(c5, a, b); M_AVV(c6, a, b); M_AVV(c7, a, b); }   // eight accumulators (PV)
    if constexpr (MODE == 6) { M_VVA(c0, a, b); M_VVA(c1, a, b); M_VVA(c2, a, b); M_VVA(c3, a, b); M_AVV(c4, a, b); M_AVV(c5, a, b); M_AVV(c6, a, b); M_AVV(c7, a, b); }   // mixed files
    if constexpr (MODE == 7) {   // builtin (compiler-chosen registers), four accumulators
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int waves) {
  float* out; long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, waves * 64>>>(out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, waves * 64>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s waves/CU %d: %6.1f clock64 ticks per MFMA, %6.2f ns per MFMA per SIMD (wall)\n", name, waves, (double)c / (iters * 8.0),
         ms * 1e6 / (iters * 8.0) / (waves > 4 ? waves / 4.0 : 1.0));
}
int main() {
  run<0>("dst V, A V, B V, 4 accumulators", 4);
  run<1>("dst V, A V, B AGPR, 4 accumulators", 4);
  run<2>("dst AGPR, A V, B V, 4 accumulators", 4);
  run<3>("dst V, 2 accumulators in rotation", 4);
  run<4>("dst V, one dependent chain", 4);
  run<5>("dst AGPR, 8 accumulators", 4);
  run<6>("4 x (dst V, B AGPR) + 4 x (dst AGPR)", 4);
  run<7>("builtin, 4 accumulators", 4);
  run<0>("dst V, A V, B V, 4 accumulators", 8);
  return 0;
}
