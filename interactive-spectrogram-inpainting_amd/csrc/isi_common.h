// Internal helpers shared by the HIP translation units of libisi_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "isi_hip.h"
#include "knobs.h"

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

// hipFuncSetAttribute and the CU count are PER DEVICE: a process that drives a second GPU (tests on cuda:1, one
// process over several devices) must not reuse the first device's cached answer.
inline int current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0;
}
struct DeviceOnce {               // "has this been done on the current device?" (racing threads repeat an idempotent call)
  unsigned long long bits[4] = {0, 0, 0, 0};
  bool done() const { const int d = current_device() & 255; return (bits[d >> 6] >> (d & 63)) & 1ull; }
  void mark() { const int d = current_device() & 255; bits[d >> 6] |= 1ull << (d & 63); }
};
inline int current_device_cu_count() {
  if (const int forced = knobs().cu_count; forced > 0) return forced;   // (a stream with a CU mask: tools/concurrent_halves.py)
  static int cached[256];         // 0 = not asked yet
  const int d = current_device() & 255;
  if (!cached[d]) {
    int n = 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cached[d] = n;
  }
  return cached[d];
}

constexpr int kBK = 32;  // K-chunk of the implicit GEMM; packed weights are padded to it

#if defined(__HIPCC__)
// ---- cross-lane reductions on the VALU's DPP path (gfx9 controls).  `__shfl_xor` compiles to ds_bpermute_b32 -- a trip
// through the LDS crossbar, ~100 cycles, six of them chained per 64-lane reduction -- which is what the latency-bound
// one-row kernels of the decoding loop spent a third of their time in; a DPP step costs a VALU instruction.
// dpp(v, ctrl): every lane reads v of the lane the control selects (rows of 16 lanes).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// XCD-aware decode of a (tile, head, batch) launch.  Workgroup `id` of a launch runs on XCD id % 8 (observed placement,
// a speed matter only) and every XCD has its own 4 MB L2: all tiles of one (batch, head) pair -- they stream the same
// keys / values -- are therefore given ids of ONE residue class.  The launch is 1-D with xcd_grid(tiles, pairs)
// workgroups; a pair index beyond the last pair (pairs not a multiple of 8) returns false and the workgroup leaves.
__host__ __device__ __forceinline__ unsigned xcd_grid(int tiles, int pairs) { return 8u * (unsigned)((pairs + 7) / 8) * (unsigned)tiles; }
// `heavy_first`: tile-major order inside an XCD (every pair's tile 0, then every pair's tile 1, ...): with tiles of
// unequal cost (causal masks, heaviest = tile 0) the long ones all start first; otherwise pair-major (a pair's tiles are
// neighbours in time: ~4 instead of 8 pairs' operands live in the L2 at once).
__device__ __forceinline__ bool xcd_tile(int tiles, int pairs, bool heavy_first, int &tile, int &pair) {
  const int id = (int)blockIdx.x, slot = id >> 3;
  const int ppx = (pairs + 7) / 8;          // pairs per XCD
  int pl;
  if (heavy_first) { tile = slot / ppx; pl = slot % ppx; }
  else { tile = slot % tiles; pl = slot / tiles; }
  pair = pl * 8 + (id & 7);
  return pair < pairs;
}
// sum over aligned groups of 4 / 8 / 16 lanes, result in every lane of the group
__device__ __forceinline__ float group4_sum(float v) {
  v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float group8_sum(float v) { v = group4_sum(v); return v + dpp_f32<0x141>(v); }   // row_half_mirror
__device__ __forceinline__ float row16_sum(float v) { v = group8_sum(v); return v + dpp_f32<0x140>(v); }    // row_mirror
// the value of lane l ^ 32 (v_permlane32_swap: one VALU instruction instead of a ds_bpermute round trip)
__device__ __forceinline__ float xor32_f32(const float v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const u32x2_ r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // .x: upper lanes get the lower's, .y: lower get the upper's
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r.x : r.y);
}
// sum / maximum over the 64 lanes, result in every lane (broadcast through an SGPR)
__device__ __forceinline__ float wave64_sum(float v) {
  v = row16_sum(v);
  v += dpp_f32<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp_f32<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave64_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));  // every lane: its row's maximum
  // rows 1 and 3 take in their left neighbour's, then rows 2 and 3 row 1's: masked-off rows read 0, so combine by hand
  const float r15 = dpp_f32<0x142, 0xA>(v);
  v = ((threadIdx.x >> 4) & 1) ? fmaxf(v, r15) : v;
  const float r31 = dpp_f32<0x143, 0xC>(v);
  v = ((threadIdx.x >> 5) & 1) ? fmaxf(v, r31) : v;
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif

}  // namespace isi
