// Probe: what v_permlane32_swap returns (gfx950).  out[0..63] = r[0], out[64..127] = r[1] for a = lane, b = 1000 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]); printf("\nr1:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]); printf("\n");
  return 0;
}
