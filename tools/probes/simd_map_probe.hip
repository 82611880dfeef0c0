// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  extern __shared__ char smem[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 0) smem[0] = 1;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 8 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipLaunchKernelGGL(probe, dim3(1024), dim3(512), 150000, 0, d);
  hipDeviceSynchronize();
  static unsigned h[4096 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int same4 = 0, same1 = 0;
  for (int b = 0; b < 1024; ++b) {
    if (b < 6) { printf("block %d simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u se %u\n", (h[b*8] >> 8) & 15, (h[b*8] >> 13) & 7); }
    for (int w = 0; w < 4; ++w) same4 += ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3);
    for (int w = 0; w < 8; w += 2) same1 += ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 1] >> 4) & 3);
  }
  printf("pairs (w, w+4) on one SIMD: %d / 4096; pairs (w, w+1): %d / 4096\n", same4, same1);
  return 0;
}
