// First encoder layer for gfx950: Conv2d(2 -> 32 / 64, kernel 4, stride 2, padding 1) + ReLU on the NCHW input
// (reference vqvae/encoder_decoder.py:95-99 with in_channel = 2, the mel-IF spectrogram).
//
// An HBM-oriented layer -- 34 MB in, 268 MB out at B = 64, 4.3 GFLOP (31 us of the exact-fp32 matrix pipe) -- that
// the generic implicit-GEMM kernel ran through its element-wise gather loader in 107 us (here: 87 us) (integer divisions per gathered element, LDS staging of a
// K = 32 "GEMM", 4-byte stores).  Here:
//   * a wave owns 32 consecutive output pixels; the 16 input values a lane needs per pixel (its channel pair is
//     selected by the half-wave) are fetched straight from global memory with out-of-range buffer offsets as zero
//     padding -- every input element is used by four taps, the texture cache absorbs the re-reads; no LDS, no
//     divisions per element;
//   * the weights ([Cout][32], 8 KB) live in registers for the lifetime of the wave (8 tiles);
//   * 16 exact-fp32 MFMA steps per 32-channel tile in the SAME k pairing and order as the generic kernel, so the
//     results are bit-identical to it (ISI_CONV_F16X3: 6 split-f16 MFMAs instead, 87 -> 81 us);
//   * tiles are handed out grid-stride, so that the resident workgroups write one contiguous region at any time
//     (8 consecutive tiles per workgroup put the concurrent writes 256 KB apart: 81 -> 75 us);
//   * the 32 x Cout tile goes through a per-wave LDS transpose so that every lane stores 16 contiguous bytes and a
//     wave writes whole 256-byte pixel rows (fp32 or the split-f16 pair format, ISI_CONV_OUT_PAIR).
#include <cstdlib>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_f16.h"

namespace isi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int TILES_PER_WAVE = 8;   // 2: 92 us, 8: 87 us at B = 64
constexpr int LDT = 68;   // LDS row of the transpose buffer (floats): 16-byte aligned rows, shifted banks

struct FirstArgs {
  const float *in, *w, *bias;
  float *out;
  unsigned in_bytes;
  int s0n, s0c, s0h, s0w;   // input element strides
  int H, W, OH, OW, M;      // M = B * OH * OW output pixels
  int relu, out_pair;
  float *out2;              // with out_pair: dense fp32 twin of the output (the training tape), or null
  const float *gate;        // fp32 output only: dense tensor laid out like it; out = gate > 0 ? value : 0 (the input gradient
                            // of the decoder's 2-channel last layer IS this convolution: its ReLU mask rides here)
};


// transposed tile (32 pixels x COUT fp32 in LDS) -> whole pixel rows: 16-byte stores of fp32 quads or, for a
// pair-format output (split_f16.h: {hi[8] | lo[8]} per group of 8 channels), of the two pieces of a group
template <int COUT>
__device__ __forceinline__ void store_tile(const FirstArgs &p, const float *tb, const int m0, const int lane) {
  if (p.out_pair) {
    constexpr int GP = COUT / 8;   // 32-byte groups per pixel
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int it = 0; it < 32 * GP / 64; ++it) {
      const int idx = it * 64 + lane;
      const int px = idx / GP, g = idx - px * GP;
      uint4 hi, lo;
      f16s::pair8_encode(*reinterpret_cast<const float4 *>(tb + px * LDT + g * 8),
                         *reinterpret_cast<const float4 *>(tb + px * LDT + g * 8 + 4), hi, lo);
      if (!p.out2 && 32 * GP % 64 == 0) {
        // (round 5) a lane owns one 32-byte group {hi | lo}: writing hi then lo made every store instruction touch HALF of
        // each 64-byte run (16 bytes on, 16 off).  v_permlane32_swap hands the lower half-wave both halves' hi pieces and the
        // upper one the lo pieces (as in conv_pair_f16.hip's epilogue): instruction A then writes the groups of lanes 0-31
        // whole -- hi from the lower lane, lo from lane + 32 --, instruction B those of lanes 32-63: 1 KiB contiguous each.
        const u32x2v sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
        const u32x2v sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
        const u32x2v sz = __builtin_amdgcn_permlane32_swap(hi.z, lo.z, false, false);
        const u32x2v sw = __builtin_amdgcn_permlane32_swap(hi.w, lo.w, false, false);
        const uint4 first = make_uint4(sx.x, sy.x, sz.x, sw.x), second = make_uint4(sx.y, sy.y, sz.y, sw.y);
        const int up = lane >> 5;                                    // 0: this lane writes hi pieces, 1: lo pieces
        const int idx_a = it * 64 + (lane & 31), idx_b = idx_a + 32; // the (pixel, group) pairs of the two instructions
        const int pa = idx_a / GP, ga = idx_a - pa * GP, pb = idx_b / GP, gb = idx_b - pb * GP;
        if (m0 + pa < p.M) reinterpret_cast<uint4 *>(p.out + (size_t)(m0 + pa) * COUT + ga * 8)[up] = first;
        if (m0 + pb < p.M) reinterpret_cast<uint4 *>(p.out + (size_t)(m0 + pb) * COUT + gb * 8)[up] = second;
        continue;
      }
      if (m0 + px < p.M) {
        uint4 *o = reinterpret_cast<uint4 *>(p.out + (size_t)(m0 + px) * COUT + g * 8);
        o[0] = hi;
        o[1] = lo;
        if (p.out2) {   // (uniform) the same 8 channels as fp32: the tape of a training step
          float4 *o2 = reinterpret_cast<float4 *>(p.out2 + (size_t)(m0 + px) * COUT + g * 8);
          o2[0] = *reinterpret_cast<const float4 *>(tb + px * LDT + g * 8);
          o2[1] = *reinterpret_cast<const float4 *>(tb + px * LDT + g * 8 + 4);
        }
      }
    }
    return;
  }
  constexpr int QP = COUT / 4;   // 16-byte quads per pixel
#pragma unroll
  for (int it = 0; it < 32 * QP / 64; ++it) {
    const int idx = it * 64 + lane;
    const int px = idx / QP, q = idx - px * QP;
    if (m0 + px < p.M) {
      float4 v = *reinterpret_cast<const float4 *>(tb + px * LDT + q * 4);
      if (p.gate) {   // (uniform)
        const float4 g = *reinterpret_cast<const float4 *>(p.gate + (size_t)(m0 + px) * COUT + q * 4);
        v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f; v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<float4 *>(p.out + (size_t)(m0 + px) * COUT + q * 4) = v;
    }
  }
}

template <int NT>   // NT = Cout / 32
__global__ __launch_bounds__(256) void conv_first_kernel(const FirstArgs p) {
  constexpr int COUT = NT * 32;
  __shared__ __attribute__((aligned(16))) float tbuf[4][32 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  float *tb = tbuf[wave];

  // weights of this lane's output channels: k = 8 s + 4 half + e  (packed [Cout][32], k = tap * 2 + c)
  float4 bw[4][NT];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bw[s][j] = *reinterpret_cast<const float4 *>(p.w + (size_t)(32 * j + col) * 32 + 8 * s + 4 * half);
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias[j] = p.bias ? p.bias[32 * j + col] : 0.f;

  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  const int ohw = p.OH * p.OW;

  for (int t = 0; t < TILES_PER_WAVE; ++t) {
    // grid-stride tile order: at any time the resident workgroups write ONE contiguous region (a workgroup that
    // owned 8 consecutive tiles put the concurrent writes 256 KB apart -- the same memory channels)
    const int tile = (t * (int)gridDim.x + (int)blockIdx.x) * 4 + wave;
    const int m0 = tile * 32;
    const int m = m0 + col;
    // ---- this lane's 16 input values: channel e & 1, row 2 oy - 1 + s, column 2 ox - 1 + (e >> 1) + 2 half
    // (issuing the next tile's loads under this tile's MFMAs was measured: 87 -> 92 us, the loads are not what a
    // wave waits for)
    float a[4][4];
    {
      const bool ok = m < p.M;
      const int b = ok ? m / ohw : 0;
      const int rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1 + 2 * half;
      const int base = b * p.s0n;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int iy = iy0 + s;
        const bool yok = ok && (unsigned)iy < (unsigned)p.H;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ix = ix0 + (e >> 1);
          const bool in_range = yok && (unsigned)ix < (unsigned)p.W;
          const unsigned off = in_range ? (unsigned)(base + (e & 1) * p.s0c + iy * p.s0h + ix * p.s0w) * 4u : OOB;
          a[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, off, 0, 0));
        }
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = e == 0 ? bw[s][j].x : e == 1 ? bw[s][j].y : e == 2 ? bw[s][j].z : bw[s][j].w;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][e], bv, acc[j], 0, 0, 0);
        }
    // ---- bias, ReLU, (pair encoding) -> LDS transpose -> 16-byte stores of whole pixel rows
    __syncthreads();   // the previous tile's reads of tb are done (uniform trip count; wave-local ordering
                       // without s_barrier was measured: no difference)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[j][r] + bias[j];
        if (p.relu) v = fmaxf(v, 0.f) + (v - v);   // NaN-propagating rectifier (see conv_igemm_f32.hip)
        tb[row * LDT + 32 * j + col] = v;
      }
    __syncthreads();
    store_tile<COUT>(p, tb, m0, lane);
  }
}

// ---- split-f16 variant (ISI_CONV_F16X3): the 32 MFMAs of 16 passes per tile (31 us of the exact-fp32 pipe per
// launch at B = 64) become 12 MFMAs of 8 passes on the f16 pipe, products as in split_f16.h.  A lane's k-block of an
// MFMA step is 8 consecutive k = (4 taps of one kernel row) x (2 channels): step st, half-wave h -> kernel row
// 2 st + h, the four columns, both channels.
typedef short s16x8f __attribute__((ext_vector_type(8)));
template <int NT>
__global__ __launch_bounds__(256) void conv_first_f16x3_kernel(const FirstArgs p) {
  constexpr int COUT = NT * 32;
  __shared__ __attribute__((aligned(16))) float tbuf[4][32 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  float *tb = tbuf[wave];

  s16x8f bh[2][NT], bl[2][NT];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float *wr = p.w + (size_t)(32 * j + col) * 32 + 16 * st + 8 * half;
      uint2 h0, l0, h1, l1;
      f16s::split4(*reinterpret_cast<const float4 *>(wr), f16s::kScaleB, h0, l0);
      f16s::split4(*reinterpret_cast<const float4 *>(wr + 4), f16s::kScaleB, h1, l1);
      bh[st][j] = __builtin_bit_cast(s16x8f, make_uint4(h0.x, h0.y, h1.x, h1.y));
      bl[st][j] = __builtin_bit_cast(s16x8f, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias[j] = p.bias ? p.bias[32 * j + col] : 0.f;

  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  const int ohw = p.OH * p.OW;

  for (int t = 0; t < TILES_PER_WAVE; ++t) {
    // grid-stride tile order: at any time the resident workgroups write ONE contiguous region (a workgroup that
    // owned 8 consecutive tiles put the concurrent writes 256 KB apart -- the same memory channels)
    const int tile = (t * (int)gridDim.x + (int)blockIdx.x) * 4 + wave;
    const int m0 = tile * 32;
    const int m = m0 + col;
    float a[2][8];   // [step][(kw, c)]
    {
      const bool ok = m < p.M;
      const int b = ok ? m / ohw : 0;
      const int rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int iy0 = 2 * oy - 1 + half, ix0 = 2 * ox - 1;
      const int base = b * p.s0n;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int iy = iy0 + 2 * st;
        const bool yok = ok && (unsigned)iy < (unsigned)p.H;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ix = ix0 + (i >> 1);
          const bool in_range = yok && (unsigned)ix < (unsigned)p.W;
          const unsigned off = in_range ? (unsigned)(base + (i & 1) * p.s0c + iy * p.s0h + ix * p.s0w) * 4u : OOB;
          a[st][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, off, 0, 0));
        }
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      uint2 h0, l0, h1, l1;
      f16s::split4(make_float4(a[st][0], a[st][1], a[st][2], a[st][3]), f16s::kScaleA, h0, l0);
      f16s::split4(make_float4(a[st][4], a[st][5], a[st][6], a[st][7]), f16s::kScaleA, h1, l1);
      const s16x8f ah = __builtin_bit_cast(s16x8f, make_uint4(h0.x, h0.y, h1.x, h1.y));
      const s16x8f al = __builtin_bit_cast(s16x8f, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16s::f16x8, al), __builtin_bit_cast(f16s::f16x8, bh[st][j]), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16s::f16x8, ah), __builtin_bit_cast(f16s::f16x8, bl[st][j]), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16s::f16x8, ah), __builtin_bit_cast(f16s::f16x8, bh[st][j]), acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[j][r] * f16s::kUnscale + bias[j];
        if (p.relu) v = fmaxf(v, 0.f) + (v - v);
        tb[row * LDT + 32 * j + col] = v;
      }
    __syncthreads();
    store_tile<COUT>(p, tb, m0, lane);
  }
}

template <int NT, bool F16>
int launch_first(const FirstArgs &a, hipStream_t stream) {
  const int tiles = (a.M + 31) / 32;
  const int per_wg = 4 * TILES_PER_WAVE;
  dim3 grid((tiles + per_wg - 1) / per_wg);
  const double flops = 2.0 * a.M * (NT * 32) * 32;
  const double bytes = 4.0 * ((double)a.M / (a.OH * a.OW) * 2.0 * a.H * a.W + (double)a.M * NT * 32 + NT * 32 * 32);
  prof::Scope scope(prof::K_CONV_GATHER, flops, bytes, stream);
  if (F16) {
    ISI_PROF_LAUNCH(scope, conv_first_f16x3_kernel<NT>, grid, dim3(256), 0, stream, a);
  } else {
    ISI_PROF_LAUNCH(scope, conv_first_kernel<NT>, grid, dim3(256), 0, stream, a);
  }
  return check_launch("conv_first_f32");
}

}  // namespace

// Shapes this kernel takes over from the generic convolution (conv2d_batched_f32 asks before it plans its own launch).
bool conv_first_applicable(const isi_src *s0, const isi_src *s1, const isi_src *res, const isi_dst *dst, int Cout,
                           int KH, int KW, int stride, int pad, int OH, int OW, int nz) {
  const bool off = knobs().no_conv_first != 0;   // measurements / tests: the generic gather kernel instead
  if (off || nz != 1 || (s1 && s1->ptr) || (res && res->ptr)) return false;
  if (s0->C != 2 || KH != 4 || KW != 4 || stride != 2 || pad != 1) return false;
  if (Cout != 32 && Cout != 64) return false;
  if (dst->sc != 1 || dst->sw != Cout || dst->sh != (int64_t)OW * Cout || dst->sn != (int64_t)OH * OW * Cout) return false;
  return (reinterpret_cast<uintptr_t>(dst->ptr) & 15) == 0;
}

int conv_first_f32(const isi_src *s0, const float *packed_w, const float *bias, const isi_dst *dst, int B, int H,
                   int W, int Cout, int OH, int OW, int64_t in_extent, int flags, hipStream_t stream, float *twin,
                   const float *gate) {
  FirstArgs a;
  if (twin && !(flags & ISI_CONV_OUT_PAIR)) return unsupported("conv_first: an fp32 twin accompanies a pair-format output");
  if (gate && (flags & (ISI_CONV_OUT_PAIR | ISI_CONV_GATE_PAIR))) return unsupported("conv_first: the gated epilogue writes and reads fp32");
  a.out2 = twin; a.gate = gate;
  a.in = s0->ptr; a.w = packed_w; a.bias = bias; a.out = dst->ptr;
  a.in_bytes = (unsigned)(in_extent * 4);
  a.s0n = (int)s0->sn; a.s0c = (int)s0->sc; a.s0h = (int)s0->sh; a.s0w = (int)s0->sw;
  a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.M = B * OH * OW;
  a.relu = flags & ISI_CONV_RELU; a.out_pair = (flags & ISI_CONV_OUT_PAIR) ? 1 : 0;
  // split-f16 products only where the caller asked for them and not for the six-term mode (most precise wins)
  if ((flags & ISI_CONV_F16X3) && !(flags & ISI_CONV_BF16X6))
    return Cout == 64 ? launch_first<2, true>(a, stream) : launch_first<1, true>(a, stream);
  return Cout == 64 ? launch_first<2, false>(a, stream) : launch_first<1, false>(a, stream);
}

}  // namespace isi
