// ds_read_b64_tr_b16 semantics probe (gfx950).  LDS holds lds[i] = i (16-bit); every lane of a wave supplies its own
// 8-byte aligned address; the probe prints, per lane, the four 16-bit values it receives.  Two address patterns:
//   A: lane t of a 16-lane group g points at row (4 g + t / 4), columns 4 (t % 4) of a [rows][64] image
//   B: lane t points at row t of a [rows][64] image, column block 4 g   (16 rows x 4 columns per group)
// hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o tools/probes/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 *lds_v4;

__global__ void probe(short *out, int pattern) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, g = l >> 4;
  int e;
  if (pattern == 0) e = (4 * g + t / 4) * 64 + 4 * (t % 4);
  else e = t * 64 + 4 * g;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + e));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
  short *d, h[256];
  hipMalloc(&d, sizeof h);
  for (int pattern = 0; pattern < 2; ++pattern) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("pattern %c (value = row * 64 + column)\n", 'A' + pattern);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
      printf("\n");
    }
  }
  return 0;
}
