// v_permlane32_swap on gfx950: which halves end up where (used by the pair-format stores of vq_nearest.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
}
int main() {
  unsigned *d, h[128];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("a = lane, b = lane + 100\nr.x: lane0 %u lane31 %u lane32 %u lane63 %u\nr.y: lane0 %u lane31 %u lane32 %u lane63 %u\n",
         h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127]);
  return 0;
}
