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
// Activation PAIR format ("pair8"): every aligned group of 8 consecutive channels (32 bytes, the footprint of 8 floats)
// holds {hi0 .. hi7 | lo0 .. lo7}: sixteen bytes of f16 hi pieces followed by sixteen bytes of f16 lo pieces of 4 x.
// A 16-byte piece IS an MFMA operand fragment (8 consecutive k of one plane), so a consumer stages a tile with plain
// 16-byte copies -- `buffer_load ... lds` straight into LDS in conv_pair_f16.hip -- and no conversion or permute.
// Weight pair format: the same blocking over k ({hi0 .. hi7 | lo0 .. lo7} of 1024 w per group of 8 floats).
// Tensors in these formats have channel / k counts that are multiples of 8.
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
// ---- pair8 groups: 8 floats <-> {hi[8] | lo[8]}
__device__ __forceinline__ void group8_encode(const float4 a, const float4 b, const float s, uint4 &hi, uint4 &lo) {
  uint2 h0, l0, h1, l1;
  split4(a, s, h0, l0);
  split4(b, s, h1, l1);
  hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
  lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
__device__ __forceinline__ void pair8_encode(const float4 a, const float4 b, uint4 &hi, uint4 &lo) {
  group8_encode(a, b, kScaleA, hi, lo);
}
__device__ __forceinline__ void weight8_encode(const float4 a, const float4 b, uint4 &hi, uint4 &lo) {
  group8_encode(a, b, kScaleB, hi, lo);
}
__device__ __forceinline__ float pair_value(const unsigned short h, const unsigned short l) {
  return ((float)__builtin_bit_cast(_Float16, h) + (float)__builtin_bit_cast(_Float16, l)) * (1.f / kScaleA);
}
// (hi + lo) / 4 of a group: exact sums (two 11-bit pieces fit an fp32 significand), exact scaling
__device__ __forceinline__ void pair8_decode(const uint4 hi, const uint4 lo, float4 &a, float4 &b) {
  a = make_float4(pair_value(hi.x & 0xffffu, lo.x & 0xffffu), pair_value(hi.x >> 16, lo.x >> 16),
                  pair_value(hi.y & 0xffffu, lo.y & 0xffffu), pair_value(hi.y >> 16, lo.y >> 16));
  b = make_float4(pair_value(hi.z & 0xffffu, lo.z & 0xffffu), pair_value(hi.z >> 16, lo.z >> 16),
                  pair_value(hi.w & 0xffffu, lo.w & 0xffffu), pair_value(hi.w >> 16, lo.w >> 16));
}
// ---- mixed-precision fmas on packed f16 pieces (v_fma_mix_f32; hipcc has no builtin and does not form it from
// (float)h + (float)l): HALF selects the low / high f16 of the dwords.  Register-only VALU statements: no memory
// counters, no MFMA operands involved (do not feed them an MFMA result directly: the compiler pads no hazard for asm).
//   mix_sum<HALF>(hw, lw)   = f32(hw.half) + f32(lw.half)          (exact for the two pieces of a pair: 4 x)
//   mix_diff<HALF>(hw, x)   = x - f32(hw.half)                     (exact: the residual the lo piece encodes)
template <int HALF>
__device__ __forceinline__ float mix_sum(const unsigned hw, const unsigned lw) {
  float d;
  if constexpr (HALF) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hw), "v"(lw));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hw), "v"(lw));
  return d;
}
template <int HALF>
__device__ __forceinline__ float mix_diff(const unsigned hw, const float x) {
  float d;
  if constexpr (HALF) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hw), "v"(x));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hw), "v"(x));
  return d;
}
// two values already in units of the pair scale -> their packed hi and lo pieces (hi = f16(x), lo = f16(x - hi))
__device__ __forceinline__ void split2_scaled(const float a, const float b, unsigned &hi, unsigned &lo) {
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const f16x2 l = __builtin_convertvector(f32x2{mix_diff<0>(hi, a), mix_diff<1>(hi, b)}, f16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

// one element of a pair8 tensor (flat element index i): for the rare scalar accessor
__device__ __forceinline__ float pair8_load(const unsigned short *base, const int64_t i) {
  const int64_t g = i >> 3;
  const int e = (int)(i & 7);
  return pair_value(base[g * 16 + e], base[g * 16 + 8 + e]);
}

}  // namespace f16s
}  // namespace isi
