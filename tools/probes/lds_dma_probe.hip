// Probe of the LDS-DMA building block used by csrc/conv_pair_f16.hip (run on the GPU box):
//   * `buffer_load_dwordx4 ... offen lds` issued from inline asm (the compiler neither counts nor waits for it)
//   * M0 = wave-uniform LDS byte address; lane l lands at M0 + 16 l
//   * an out-of-range voffset writes ZEROS to the lane's LDS slot (free zero padding)
//   * several DMAs in flight, retired in order by counted s_waitcnt vmcnt(N)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef int i32x4 __attribute__((ext_vector_type(4)));

// M0 is NOT saved: nothing else in these kernels uses it (checked in the .s)
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 rsrc, unsigned soff = 0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

extern "C" __global__ __launch_bounds__(256) void probe(const float* in, float* out, unsigned nbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // raw buffer descriptor: base address (48 bits), stride 0, num_records = bytes, flags 0x00020000 (as make_buffer_rsrc)
  const unsigned long long base = (unsigned long long)in;
  const i32x4 rsv = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)nbytes, 0x00020000};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // poison LDS
  for (int i = tid; i < 16384 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = -7.f;
  __syncthreads();
  // 4 DMAs per wave (4 KiB per wave = 16 KiB per WG), lane order permuted by ^3, lane 5 of DMA 1 out of range
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned voff = (unsigned)(blockIdx.x * 16384 + wave * 4096 + j * 1024 + (lane ^ 3) * 16);
    if (j == 1 && lane == 5) voff = 0xFFFFFFF0u;
    if (j == 2 && lane == 9) voff = nbytes;      // first byte past the buffer
    unsigned soff = 0;
    if (j == 3) {                                 // uniform part of the address in the SGPR offset
      soff = (unsigned)(blockIdx.x * 16384 + 4096);
      voff -= soff;                               // >= 0 for wave >= 1; wave 0: wraps "negative" (lanes of wave 0 test that)
      if (lane == 7) voff = 0x7FFFFFF0u;          // out-of-range marker that cannot wrap when soffset is added
      if (lane == 11) voff = 0xFFFFFFF0u;         // the marker that WOULD wrap if the range check included soffset
    }
    dma16(lds0 + wave * 4096 + j * 1024, voff, rsv, soff);
  }
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // DMAs 0 and 1 landed
  float4 a = *reinterpret_cast<float4*>(smem + wave * 4096 + 0 * 1024 + lane * 16);
  float4 b = *reinterpret_cast<float4*>(smem + wave * 4096 + 1 * 1024 + lane * 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float4 c = *reinterpret_cast<float4*>(smem + wave * 4096 + 2 * 1024 + lane * 16);
  float4 d = *reinterpret_cast<float4*>(smem + wave * 4096 + 3 * 1024 + lane * 16);
  float4* o = reinterpret_cast<float4*>(out) + (size_t)blockIdx.x * 1024 + wave * 256;
  o[0 * 64 + lane] = a; o[1 * 64 + lane] = b; o[2 * 64 + lane] = c; o[3 * 64 + lane] = d;
}

int main() {
  const int nblk = 512;
  const size_t n = (size_t)nblk * 4096;   // floats
  std::vector<float> h(n), r(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 100003) + 1.f;
  float *din, *dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0xff, n * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 16384, 0, din, dout, (unsigned)(n * 4));
    hipDeviceSynchronize();
  }
  hipMemcpy(r.data(), dout, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, zeros_ok = 0, zeros_bad = 0, wrap_zero = 0, wrap_data = 0, neg_ok = 0, neg_zero = 0;
  for (int b = 0; b < nblk; ++b) for (int w = 0; w < 4; ++w) for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const size_t oi = ((size_t)b * 1024 + w * 256 + j * 64 + l) * 4 + e;
    const bool oob = (j == 1 && l == 5) || (j == 2 && l == 9) || (j == 3 && l == 7);
    if (j == 3 && (l == 11 || w == 0)) { if (l == 11) { if (r[oi] == 0.f) ++wrap_zero; else ++wrap_data; } else if (r[oi] == h[((size_t)b * 4096 + w * 1024 + j * 256 + (l ^ 3) * 4 + e)]) ++neg_ok; else ++neg_zero; continue; }
    const size_t src = (size_t)b * 4096 + w * 1024 + j * 256 + (l ^ 3) * 4 + e;
    if (oob) { if (r[oi] == 0.f) ++zeros_ok; else { ++zeros_bad; if (zeros_bad < 5) printf("oob lane got %f\n", r[oi]); } }
    else if (r[oi] != h[src]) { if (++bad < 10) printf("mismatch b%d w%d j%d l%d e%d: %f vs %f\n", b, w, j, l, e, r[oi], h[src]); }
  }
  printf("LDS-DMA probe: %zu mismatches, out-of-range lanes zero-filled %zu / %zu\n", bad, zeros_ok, zeros_ok + zeros_bad);
  printf("voffset 0xFFFFFFF0 + soffset: zero %zu, data %zu (range check %s soffset)\n", wrap_zero, wrap_data, wrap_data ? "INCLUDES" : "excludes");
  printf("'negative' voffset + soffset (valid sum): data %zu, zero %zu\n", neg_ok, neg_zero);
  return (bad || zeros_bad) ? 1 : 0;
}
