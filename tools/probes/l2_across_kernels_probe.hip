// Does data a kernel brought into the (per-XCD) L2 survive into the next dependent kernel on MI355X?
// Pairs of launches over cold 1-MiB chunks of a 1-GiB buffer: touch(chunk) ; consume(chunk)  against
// touch(other chunk) ; consume(chunk).  Workgroup j of both kernels reads the same bytes (same XCD: j % 8).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void touch(const float4 *p, float *sink) {   // 256 workgroups x 256 threads x 16 B = 1 MiB
  const float4 v = p[blockIdx.x * 256 + threadIdx.x];
  if (v.x == 123.456f) sink[0] = v.y;                   // never true: keeps the load
}
__global__ void consume(const float4 *p, float *out) {
  const float4 v = p[blockIdx.x * 256 + threadIdx.x];
  float s = (v.x + v.y) + (v.z + v.w);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

int main() {
  const size_t total = (size_t)1 << 30, chunk = (size_t)1 << 20, n = total / chunk;
  char *buf; float *out, *sink;
  hipMalloc(&buf, total); hipMemset(buf, 0, total);
  hipMalloc(&out, 4096 * 4); hipMalloc(&sink, 64);
  hipStream_t st; hipStreamCreate(&st);
  // replayed as a graph of 128 iterations (direct launches are bound by the host: ~2.8 us each)
  auto run = [&](const char *name, int mode) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 128; ++i) {
      const size_t c = ((size_t)i * 37) % n, other = (c + n / 2) % n;
      const float4 *pc = reinterpret_cast<const float4 *>(buf + c * chunk);
      const float4 *po = reinterpret_cast<const float4 *>(buf + other * chunk);
      if (mode == 0) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, pc, sink);
      if (mode == 1) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, po, sink);
      hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, st, pc, out);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    const int reps = 20;
    for (int rep = 0; rep < 2; ++rep) {
      hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 1) printf("%-44s %.2f us per iteration\n", name, us / (reps * 128));
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  };
  run("touch(same chunk) ; consume(chunk)", 0);
  run("touch(other chunk) ; consume(chunk)", 1);
  run("consume(chunk) alone", 2);
  return 0;
}
