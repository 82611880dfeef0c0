// Cost of one dependent kernel launch in a stream on MI355X: N launches of a one-thread kernel (each reads what
// the previous wrote), direct and as a replayed hipGraph.  Build: hipcc -O2 --offload-arch=gfx950 launch_chain_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void step(int *p) { *p = *p + 1; }
__global__ void step_wide(int *p, float *buf) {   // 256 workgroups, every one touches memory
  buf[blockIdx.x * 256 + threadIdx.x] += 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *p = *p + 1;
}

__global__ void spin(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
}
struct Big { float pad[48]; int *p; };
__global__ void step_args(Big b) { *b.p = *b.p + 1; }

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2000;
  int *p; float *buf;
  hipMalloc(&p, 4); hipMemset(p, 0, 4);
  hipMalloc(&buf, 256 * 256 * 4); hipMemset(buf, 0, 256 * 256 * 4);
  hipStream_t st; hipStreamCreate(&st);
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 100; ++i) launch();
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) launch();
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %.2f us per launch\n", name, us / N);
  };
  run("one thread, direct", [&] { hipLaunchKernelGGL(step, dim3(1), dim3(1), 0, st, p); });
  run("256 workgroups, direct", [&] { hipLaunchKernelGGL(step_wide, dim3(256), dim3(256), 0, st, p, buf); });
  {  // host cost of a launch alone: the GPU is kept busy by a 20 ms kernel, the launches only queue up behind it
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, st, 2000000LL);   // 100 MHz wall clock
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(step, dim3(1), dim3(1), 0, st, p);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %.2f us per launch (host only)\n", "enqueue behind a busy GPU", us / 1000);
    Big b{}; b.p = p;
    hipStreamSynchronize(st);
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, st, 2000000LL);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(step_args, dim3(64), dim3(256), 0, st, b);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %.2f us per launch (host only)\n", "same, 200-byte arguments", us / 1000);
    hipStreamSynchronize(st);
  }
  // graph of 64 dependent launches
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(step_wide, dim3(256), dim3(256), 0, st, p, buf);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 4; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  const int R = N / 64;
  for (int i = 0; i < R; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("%-28s %.2f us per launch\n", "256 workgroups, graph of 64", us / (R * 64));
  int h; hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost);
  printf("counter %d\n", h);
  return 0;
}
