// Optional per-launch timing (HIP events on the launch stream) used by bench.py
// to price each kernel against its roofline.  Off by default; when off the cost
// is one thread-local load per launch.
#pragma once
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
  K_COUNT
};

const char *kernel_name(int id);
bool enabled();
// Records start/stop events around a launch when profiling is on.
struct Scope {
  Scope(int kernel_id, double flops, double bytes, hipStream_t stream);
  ~Scope();
  int slot;
  hipStream_t stream;
};

int enable(int on);
int read(int kernel_id, long long *launches, double *ms, double *flops, double *bytes);

}  // namespace prof
}  // namespace isi
