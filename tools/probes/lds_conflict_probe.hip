// LDS bank-conflict probe for the access patterns of the attention kernels (gfx950): every pattern is its own kernel
// (8 waves, the instruction repeated 256 times), so that rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE gives
// the conflict share per pattern:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_conflict_probe.hip -o tools/probes/lds_conflict_probe
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- tools/probes/lds_conflict_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int REP = 256;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

#define PROBE_READ(NAME, TYPE, ADDR)                                                                   \
  __global__ __launch_bounds__(512) void NAME(float *out) {                                            \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                     \
    const int ql = lane & 31, half = lane >> 5; (void)ql; (void)half; (void)wave;                      \
    for (int i = tid; i < 160 * 1024 / 4; i += 512) reinterpret_cast<float *>(smem)[i] = 1.f;         \
    __syncthreads();                                                                                   \
    TYPE acc = {};                                                                                     \
    for (int it = 0; it < REP; ++it) {                                                                 \
      const int t = it & 3; (void)t;                                                                   \
      acc += *reinterpret_cast<const TYPE *>(smem + (ADDR));                                           \
      asm volatile("" : "+v"(acc));                                                                    \
    }                                                                                                  \
    out[tid] = acc[0];                                                                                 \
  }
#define PROBE_WRITE(NAME, TYPE, ADDR)                                                                  \
  __global__ __launch_bounds__(512) void NAME(float *out) {                                            \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                     \
    const int ql = lane & 31, half = lane >> 5; (void)ql; (void)half; (void)wave;                      \
    TYPE v = {};                                                                                       \
    for (int it = 0; it < REP; ++it) {                                                                 \
      const int t = it & 3; (void)t;                                                                   \
      asm volatile("" : "+v"(v));                                                                      \
      *reinterpret_cast<TYPE *>(smem + (ADDR)) = v;                                                    \
    }                                                                                                  \
    __syncthreads();                                                                                   \
    out[tid] = reinterpret_cast<float *>(smem)[tid];                                                   \
  }

// fragment reads: 16 bytes at (row = lane & 31, k-block t, half)
PROBE_READ(frag_b128_row144, f4, ql * 144 + half * 16 + t * 32)                       // padded rows (HD 64 + 8 ushorts)
PROBE_READ(frag_b128_row128_xor, f4, ql * 128 + (((2 * t + half) ^ ((ql >> 1) & 7)) * 16))   // 128-byte rows, XOR swizzle
PROBE_READ(frag_b128_row128_plain, f4, ql * 128 + half * 16 + t * 32)                 // 128-byte rows, nothing
PROBE_READ(frag_b128_row80, f4, ql * 80 + half * 16 + (t & 1) * 32)                   // V^T rows at KT = 32 (+ 8 ushorts)
// skew buffer [8 waves][32 queries][68 floats]
PROBE_WRITE(skew_w_b128_ld68, f4, wave * 8704 + ql * 272 + (32 * (t & 1) + 8 * (t >> 1) + 4 * half) * 4)
PROBE_READ(skew_r_b32_ld68, f2, wave * 8704 + ((69 * ql + 31 - 4 * half - (t + 8 * t)) * 4 & ~7))   // (b64-aligned stand-in)
__global__ __launch_bounds__(512) void skew_r_b32_true(float *out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, half = lane >> 5;
  for (int i = tid; i < 160 * 1024 / 4; i += 512) reinterpret_cast<float *>(smem)[i] = 1.f;
  __syncthreads();
  float acc = 0.f;
  const float *rd = reinterpret_cast<const float *>(smem + wave * 8704) + 69 * ql + 31 - 4 * half;
  for (int it = 0; it < REP / 16; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += rd[-((r & 3) + 8 * (r >> 2))];
    asm volatile("" : "+v"(acc));
  }
  out[tid] = acc;
}
// staging writes: 8 bytes at (row = idx / 16, quad = idx % 16), idx = tid: K / E rows
PROBE_WRITE(putrow_b64_row144, f2, (tid >> 4) * 144 + (tid & 15) * 8)
PROBE_WRITE(putrow_b64_row128_xor, f2, (tid >> 4) * 128 + ((((tid & 15) >> 1) ^ (((tid >> 4) >> 1) & 7)) * 16) + (tid & 1) * 8)
// V^T staging: thread = (key group kg = tid / 16, dim quad qd = tid % 16), element e: row d = 4 qd + e (e = t), 8 bytes
PROBE_WRITE(vt_w_b64_row144, f2, (4 * (tid & 15) + t) * 144 + (2 * (((tid >> 4) & 15) >> 2) + ((tid >> 4) & 1)) * 16 + ((((tid >> 4) & 15) >> 1) & 1) * 8)

// V^T with the 16-byte slot XOR-ed by (row >> 2) & 7 on top of the padded rows: what the staging writes want
PROBE_WRITE(vt_w_b64_row144_xorq, f2, (4 * (tid & 15) + t) * 144 + (((2 * (((tid >> 4) & 15) >> 2) + ((tid >> 4) & 1)) ^ ((tid & 15) & 7)) * 16) + ((((tid >> 4) & 15) >> 1) & 1) * 8)
PROBE_READ(frag_b128_row144_xorq, f4, ql * 144 + (((2 * t + half) ^ ((ql >> 2) & 7)) * 16))
// the same for KT = 32 rows (80 bytes: 4 slots, XOR by (row >> 2) & 3)
PROBE_WRITE(vt_w_b64_row80_xorq, f2, (4 * (tid & 15) + t) * 80 + (((2 * (((tid >> 4) & 7) >> 2) + ((tid >> 4) & 1)) ^ ((tid & 15) & 3)) * 16) + ((((tid >> 4) & 7) >> 1) & 1) * 8)
PROBE_READ(frag_b128_row80_xorq, f4, ql * 80 + (((2 * (t & 1) + half) ^ ((ql >> 2) & 3)) * 16))

// V^T, 128-byte rows (64 keys), slot c of row d at c ^ f(d), f(d) = ((d >> 2) & 7) ^ ((d & 2) << 1)
#define VT_F(d_) ((((d_) >> 2) & 7) ^ (((d_) & 2) << 1))
PROBE_WRITE(vt_w_b64_row128_f, f2, (4 * (tid & 15) + t) * 128 + (((2 * (((tid >> 4) & 15) >> 2) + ((tid >> 4) & 1)) ^ VT_F(4 * (tid & 15) + t)) * 16) + ((((tid >> 4) & 15) >> 1) & 1) * 8)
PROBE_READ(frag_b128_row128_f, f4, ql * 128 + (((2 * t + half) ^ VT_F(ql)) * 16))

int main() {
  float *d;
  hipMalloc(&d, 512 * sizeof(float));
#define RUN(K)                                                                                                 \
  hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  hipLaunchKernelGGL(K, dim3(256), dim3(512), 160 * 1024, 0, d);
  RUN(frag_b128_row144) RUN(frag_b128_row128_xor) RUN(frag_b128_row128_plain) RUN(frag_b128_row80)
  RUN(skew_w_b128_ld68) RUN(skew_r_b32_ld68) RUN(skew_r_b32_true) RUN(putrow_b64_row144) RUN(putrow_b64_row128_xor) RUN(vt_w_b64_row144)
  RUN(vt_w_b64_row144_xorq) RUN(frag_b128_row144_xorq) RUN(vt_w_b64_row80_xorq) RUN(frag_b128_row80_xorq)
  RUN(vt_w_b64_row128_f) RUN(frag_b128_row128_f)
  hipDeviceSynchronize();
  printf("done: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
