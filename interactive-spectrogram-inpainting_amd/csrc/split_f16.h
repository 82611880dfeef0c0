// Split-f16 arithmetic shared by the convolution, residual-block, first-layer, quantiser and packing kernels
// (ISI_CONV_F16X3, ISI_CONV_W16, ISI_CONV_*_PAIR in include/isi_hip.h).  ONE definition on purpose: producers and
// consumers must compute the same pieces bit for bit.
//
//   x s = hi + lo,  hi = f16(x s),  lo = f16(x s - hi),  s a power of two
//
// Two 11-bit pieces hold 22 significand bits and the rounding of lo leaves |x - (hi + lo) / s| <= 2^-24 |x|; a product
// hi.hi + hi.lo + lo.hi drops lo.lo <= 2^-24 of it: fp32-grade products from THREE f16 MFMAs (the six-term bf16 split
// needs six).  The price is f16's range: |s x| must stay below 65504 (an overflow turns into Inf / NaN in the output,
// never into a silently wrong value) and lo keeps all its bits only while |s x| >= 2^-3 (below, the absolute error
// floor is 2^-25 / s; the matrix pipe does not flush f16 subnormals).  Activations are scaled by 2^2 (|x| < 16384,
// floor 7e-9), weights and code vectors by 2^10 (|w| < 64, floor 3e-11); accumulators are rescaled by 2^-12 in the
// epilogue (all exact).
//
// Activation PAIR format: an element's 4 bytes = hi (low half) | lo << 16 of 4 x.  Weight pair format: a quad of
// floats becomes {hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3} of 1024 w in the same 16 bytes.
#pragma once
#include <hip/hip_runtime.h>

namespace isi {
namespace f16s {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kScaleA = 4.f, kScaleB = 1024.f, kUnscale = 1.f / (4.f * 1024.f);

// four floats -> packed hi / lo pieces (v_pk_mul_f32, v_cvt_pk_f16_f32: ~14 VALU instructions)
__device__ __forceinline__ void split4(const float4 v, const float s, uint2 &hi, uint2 &lo) {
  const f32x2 a = f32x2{v.x, v.y} * s, b = f32x2{v.z, v.w} * s;
  const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
  const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), f16x2);
  const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), f16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}
// weight pair quad {hi0..3 | lo0..3} -> pieces (no arithmetic)
__device__ __forceinline__ void weight_quad(const float4 v, uint2 &hi, uint2 &lo) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  hi = make_uint2(u.x, u.y);
  lo = make_uint2(u.z, u.w);
}
__device__ __forceinline__ uint4 weight_encode(const float4 v) {
  uint2 hi, lo;
  split4(v, kScaleB, hi, lo);
  return make_uint4(hi.x, hi.y, lo.x, lo.y);
}
// activation pairs
__device__ __forceinline__ unsigned pair_encode(const float v) {
  const float t = v * kScaleA;
  const _Float16 h = (_Float16)t;
  const _Float16 l = (_Float16)(t - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ float pair_decode(const unsigned u) {
  return ((float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)) +
          (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16))) * (1.f / kScaleA);
}
// four pair elements -> packed hi / lo pieces: four v_perm_b32
__device__ __forceinline__ void pair_quad(const float4 v, uint2 &hi, uint2 &lo) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  hi = make_uint2(__builtin_amdgcn_perm(u.y, u.x, 0x05040100u), __builtin_amdgcn_perm(u.w, u.z, 0x05040100u));
  lo = make_uint2(__builtin_amdgcn_perm(u.y, u.x, 0x07060302u), __builtin_amdgcn_perm(u.w, u.z, 0x07060302u));
}

}  // namespace f16s
}  // namespace isi
