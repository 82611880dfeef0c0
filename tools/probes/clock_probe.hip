// What shader clock does a latency-bound chain of small launches run at?  The decode loop keeps 16-64 of the 256 CUs busy with
// ~5 us kernels: this probe replays such a chain (a graph of N tiny dependent kernels on G workgroups) and lets every kernel
// record clock64() (shader cycles) and wall_clock64() (constant 100 MHz) at its start and end; cycles per wall tick = sclk.
// Then the same chain with a "heater" -- a long-running kernel on a second stream that keeps H other workgroups busy with
// FMAs -- to see whether the power management raises the clock when more of the chip is active.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o tools/probes/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void link(long long *rec, int i, float *sink, const float *src, int work) {
  const long long c0 = clock64(), w0 = wall_clock64();
  float a = src[threadIdx.x & 63] + (float)i;
  for (int k = 0; k < work; ++k) a = __builtin_fmaf(a, 1.0000001f, 0.5f);     // a dependent chain: `work` x ~4-8 cycles
  if (a == 12345.f) sink[0] = a;
  const long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { rec[4 * i] = c0; rec[4 * i + 1] = w0; rec[4 * i + 2] = c1; rec[4 * i + 3] = w1; }
}
__global__ void heater(volatile int *stop, float *sink, int fma) {
  float a = (float)threadIdx.x, b = 1.f;
  while (!*stop) {
    for (int k = 0; k < 4096; ++k) { a = __builtin_fmaf(a, 1.0000001f, 0.5f); if (fma) b = __builtin_fmaf(b, 0.9999999f, a); }
  }
  if (a + b == 12345.f) sink[1] = a;
}

int main() {
  const int N = 2000;
  long long *rec; float *sink, *src; int *stop;
  hipMalloc(&rec, 4 * N * sizeof(long long)); hipMalloc(&sink, 64); hipMalloc(&src, 256);
  hipHostMalloc(&stop, sizeof(int), hipHostMallocMapped); *stop = 0;
  hipMemset(src, 0, 256);
  hipStream_t st, hs; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&hs, hipStreamNonBlocking);
  for (int G : {16, 64}) {
    for (int H : {0, 64, 192}) {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(link, dim3(G), dim3(256), 0, st, rec, i, sink, src, 400);
      hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      *stop = 0;
      if (H) hipLaunchKernelGGL(heater, dim3(H), dim3(256), 0, hs, stop, sink, 1);
      for (int rep = 0; rep < 3; ++rep) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
      *stop = 1; hipStreamSynchronize(hs);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(4 * N); hipMemcpy(h.data(), rec, 4 * N * sizeof(long long), hipMemcpyDeviceToHost);
      std::vector<double> mhz, dur, gap;
      for (int i = N / 2; i < N; ++i) {
        const double cyc = (double)(h[4 * i + 2] - h[4 * i]), ticks = (double)(h[4 * i + 3] - h[4 * i + 1]);
        if (ticks > 0) { mhz.push_back(cyc / ticks * 100.0); dur.push_back(ticks * 10.0); }
        if (i + 1 < N) gap.push_back((double)(h[4 * (i + 1) + 1] - h[4 * i + 3]) * 10.0);
      }
      std::sort(mhz.begin(), mhz.end()); std::sort(dur.begin(), dur.end()); std::sort(gap.begin(), gap.end());
      printf("chain of %d kernels on %3d workgroups, heater on %3d: %.2f us per link | in-kernel %.0f ns at %.0f MHz (median), gap between links %.0f ns\n",
             N, G, H, ms * 1e3 / N, dur[dur.size() / 2], mhz[mhz.size() / 2], gap[gap.size() / 2]);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
