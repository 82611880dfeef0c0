// Optional per-launch timing (HIP events on the launch stream) used by bench.py
// to price each kernel against its roofline.  Off by default; when off the cost
// is one thread-local load per launch.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace isi {
namespace prof {

enum KernelId {
  K_CONV_128x128 = 0,
  K_CONV_128x64,
  K_CONV_128x32,
  K_CONV_GATHER,   // element-wise gather A loader (NCHW / odd channel counts)
  K_VQ_NEAREST,
  K_RESBLOCK,
  K_CONVT_SMALL,
  K_REL_ATTENTION,
  K_CONV_BF16X3,
  K_REL_ATTENTION_BWD,
  K_CONV_BF16X6,
  K_CONV_F16X3,
  K_CONV_PAIR_128_PAIROUT,   // conv_pair_kernel<128, true, 0> alone (also part of the K_CONV_F16X3 family in bench.py)
  K_COUNT
};

const char *kernel_name(int id);
bool enabled();
// One record (kernel id, algorithmic work, a start / stop event pair) per launch when profiling is on.  The events
// are handed to hipExtLaunchKernelGGL (ISI_PROF_LAUNCH): the dispatch packet's own start / end timestamps are
// used, no extra event packets enter the stream (hipEventRecord pairs cost ~5 % of a forward's stream time).
struct Scope {
  Scope(int kernel_id, double flops, double bytes, hipStream_t stream);
  int slot;
  hipStream_t stream;
  hipEvent_t start() const;
  hipEvent_t stop() const;
};

#define ISI_PROF_LAUNCH(scope, kern, grid, block, smem, stream, ...)                                              \
  do {                                                                                                            \
    if ((scope).slot >= 0)                                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, smem, stream, (scope).start(), (scope).stop(), 0, __VA_ARGS__);    \
    else                                                                                                          \
      hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);                                           \
  } while (0)

int enable(int on);
int read(int kernel_id, long long *launches, double *ms, double *flops, double *bytes);

}  // namespace prof
}  // namespace isi
