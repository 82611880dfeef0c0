// Internal helpers shared by the HIP translation units of libisi_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "isi_hip.h"

namespace isi {

void set_last_error(const char *msg);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    set_last_error(buf);
    return ISI_E_LAUNCH;
  }
  return ISI_OK;
}

inline int invalid(const char *msg) {
  set_last_error(msg);
  return ISI_E_INVALID;
}

inline int unsupported(const char *msg) {
  set_last_error(msg);
  return ISI_E_UNSUPPORTED;
}

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

constexpr int kBK = 32;  // K-chunk of the implicit GEMM; packed weights are padded to it

}  // namespace isi
