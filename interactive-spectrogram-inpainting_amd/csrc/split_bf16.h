// Split-bf16 helpers shared by the attention kernels: an fp32 value is carried as hi + lo in bf16
// (hi = rne(x), lo = rne(x - hi)); a product a*b is evaluated as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// the bf16 matrix pipe with fp32 accumulation (the dropped lo*lo term is ~2^-18 of the product).
#pragma once
#include <hip/hip_runtime.h>

namespace isi {
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

// (a, b) -> packed bf16 pairs: hi and lo parts (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split2(const float a, const float b, unsigned &hi, unsigned &lo) {
  const f32x2_t v = {a, b};
  const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
  const bf16x2_t l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2_t), bf16x2_t);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_f4(const float4 v, uint2 &hi, uint2 &lo) {
  split2(v.x, v.y, hi.x, lo.x);
  split2(v.z, v.w, hi.y, lo.y);
}
// 16 accumulator registers of a 32x32 tile -> the two 16-deep k-blocks of an MFMA B operand (hi and lo):
// k-slot (half h, e) of block t is accumulator row 16 t + 8 (e >> 2) + 4 h + (e & 3) = register 8 t + e
__device__ __forceinline__ void split_acc16(const float *v, s16x8_t *hi, s16x8_t *lo) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    unsigned hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(v[8 * t + 2 * e], v[8 * t + 2 * e + 1], hh[e], ll[e]);
    hi[t] = __builtin_bit_cast(s16x8_t, make_uint4(hh[0], hh[1], hh[2], hh[3]));
    lo[t] = __builtin_bit_cast(s16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));
  }
}
}  // namespace isi

#define ISI_MFB(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0)
