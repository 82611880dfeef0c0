// Fused RosinalityResBlock for gfx950 (exact-fp32 matrix pipe).
//
//   out = [relu]( r + W2 * relu(W1 (*) r + b1) + b2 )        r = rectified input
//
// i.e. reference vqvae/encoder_decoder.py:22-35 with its in-place first ReLU
// already applied by the producer of r.  One launch replaces the 3x3 conv
// (C -> R), the ReLU, the 1x1 conv (R -> C) and the residual add; the R-channel
// intermediate never leaves the CU:
//
//   GEMM1  [128 px x 9C] x [9C x 32]   a workgroup owns 2 x 64 output pixels; per
//          32-channel slice the 4 x 66 input halo and the W1 slice are staged once
//          in LDS and all nine taps read shifted windows of it; each of the 4 waves
//          owns 32 pixels x 32 hidden channels
//   h      = relu(acc1 + b1) written to the wave's own LDS rows
//   GEMM2  [32 px x 32] x [32 x C] per wave, W2 resident in LDS, then
//          + b2 + r (centre pixel, re-read from L2), ReLU, store.
//
// Requirements (else the caller uses two isi_conv2d_f32 launches): channels-last
// dense input/output, C % 32 == 0, C <= 128, R <= 32.
#include <cstdlib>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_f16.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ResKArgs {
  const float *in, *w1, *b1, *w2, *b2;
  float *out;
  unsigned in_bytes, w1_bytes, w2_bytes;
  int C, R, H, W, relu;
  int in_pair, out_pair;   // ISI_CONV_IN0_PAIR / ISI_CONV_OUT_PAIR (split-f16 variant with pack-time weight pieces only)
};

namespace {
constexpr int LDK = 36;             // padded LDS row (floats)
constexpr int TW = 64, HWD = TW + 2;   // output tile: TH rows x 64 pixels (TH = 2: 4 waves, TH = 4: 8 waves); halo row 66
constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ float elem(const float4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}
}  // namespace

// Workgroup = 2 x 64 output pixels.  Per 32-channel slice of the input the
// (2+2) x (64+2) halo and the [9 taps][32][32] slice of W1 are staged ONCE in
// LDS (global -> registers -> LDS, the next slice's loads in flight under the
// MFMAs); the nine taps then read shifted windows of the same halo: 144 MFMAs
// per wave between barriers and 4.4x less global->LDS traffic than im2col.
// ---- opt-in split-bf16 products (ISI_CONV_BF16X3, see conv_igemm_f32.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = 32;  // bf16 plane row (elements): 64 B unpadded; 16-B slots XOR-swizzled with (row >> 2) & 3
                         // (conflict-free ds_read_b128 lane groups and ds_write_b64 groups, see conv_igemm_f32.hip)
__device__ __forceinline__ int bf_slot(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 8; }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// hi = bf16(x) (round to nearest even: v_cvt_pk_bf16_f32), lo = bf16(x - hi)
__device__ __forceinline__ void split_bf16x4(const float4 v, uint2 &hi, uint2 &lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const bf16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}
// x = hi + mid + lo exactly ("bf16x6", ISI_CONV_BF16X6)
__device__ __forceinline__ void split3_bf16x4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
  const bf16x2 la = __builtin_convertvector(ra - __builtin_convertvector(ma, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(rb - __builtin_convertvector(mb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  mid = make_uint2(__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb));
  lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}
// split-f16 pieces, pack-time weight pieces and activation pairs: split_f16.h
using f16s::f16x8;
constexpr float kF16ScaleA = f16s::kScaleA, kF16ScaleB = f16s::kScaleB, kF16Unscale = f16s::kUnscale;
__device__ __forceinline__ void split_f16x4(const float4 v, const float s, uint2 &hi, uint2 &lo) { f16s::split4(v, s, hi, lo); }
template <int PREC>
__device__ __forceinline__ void split_x4(const float4 v, const float s16, uint2 &hi, uint2 &mid, uint2 &lo) {
  if constexpr (PREC == 2) split3_bf16x4(v, hi, mid, lo);
  else if constexpr (PREC >= 3) split_f16x4(v, s16, hi, lo);
  else split_bf16x4(v, hi, lo);
}
// weights (PREC < 4: split while staged; PREC 4 = blocked pair pieces in memory, ISI_CONV_W16: stored as they are)
template <int PREC>
__device__ __forceinline__ void split_w4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
  split_x4<PREC>(v, kF16ScaleB, hi, mid, lo);
}
#define ISI_MH(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0)
#define ISI_MF(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0)
// a = {hi, lo, mid}, b = {hi, lo, mid}; smallest terms first
template <int PREC>
__device__ __forceinline__ f32x16 mfma_split(const s16x8 *a, const s16x8 *b, f32x16 acc) {
  if constexpr (PREC == 2) {
    ISI_MF(a[1], b[0]); ISI_MF(a[0], b[1]); ISI_MF(a[2], b[2]); ISI_MF(a[2], b[0]); ISI_MF(a[0], b[2]); ISI_MF(a[0], b[0]);
  } else if constexpr (PREC >= 3) {
    ISI_MH(a[1], b[0]); ISI_MH(a[0], b[1]); ISI_MH(a[0], b[0]);
  } else {
    ISI_MF(a[1], b[0]); ISI_MF(a[0], b[1]); ISI_MF(a[0], b[0]);
  }
  return acc;
}
#undef ISI_MF
#undef ISI_MH

// TH = 2: 4 waves, 2 x 64 pixels; TH = 4: 8 waves, 4 x 64 pixels -- one W1 slice and a (TH + 2)-row halo serve
// twice the pixels (1.55 instead of 2.06 staged halo rows per output row: a quarter less staging and split work),
// and the three-plane six-term variant then runs two waves per SIMD inside ONE workgroup per CU.
template <int TC, int PREC = 0, int TH = 2>  // TC = C / 32; PREC 0 exact fp32, 1 bf16x3, 2 bf16x6, 3 f16x3
__global__ __launch_bounds__(TH * 128) void resblock_f32_kernel(const ResKArgs p) {
  constexpr int NTH = TH * 128;                               // threads
  constexpr int HH = TH + 2, HPIX = HH * HWD;                 // halo (TH + 2) x 66 pixels
  constexpr int NA = (HPIX * 8 + NTH - 1) / NTH;              // halo quads per thread
  constexpr int NWT = TH == 2 ? 9 : 5;                        // W1 taps staged per thread
  constexpr bool BF = PREC >= 1;
  constexpr int NP = PREC == 2 ? 3 : 2;   // bf16 pieces per value: hi, lo(, mid)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ah = smem;                     // [HPIX][LDK]      halo slice   (later: h, [128][LDK])
  float *W1s = Ah + HPIX * LDK;         // [9][32][LDK]     W1 slice     (later: W2, [C][LDK])
  // split-bf16: bf16 planes instead -- halo [NP][HPIX][LDB], W1 slice [NP][9*32][LDB]  (piece 0 hi, 1 lo, 2 mid)
  unsigned short *Apl = reinterpret_cast<unsigned short *>(smem);
  unsigned short *Wpl = Apl + NP * HPIX * LDB;
  constexpr int APS = HPIX * LDB, WPS = 9 * 32 * LDB;   // plane strides

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int C = TC * 32;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;

  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w2), 0, p.w2_bytes, 0x00020000);

  const int lq = tid & 7, ln = tid >> 3;
  const int lr = ln & 31, lt = ln >> 5;   // W1 staging: hidden channel lr, taps lt, lt + NTH / 256, ...
  // byte offsets (channel slice 0) of this thread's halo quads / W1 quads
  unsigned a_off[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int i = tid + NTH * j;
    const int pix = i >> 3;
    const int hy = pix / HWD, hx = pix - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = pix < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    a_off[j] = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * C + lq * 4) * 4u : OOB;
  }
  const unsigned w_off = lr < p.R ? (unsigned)(lr * 9 * C + lq * 4) * 4u : OOB;

  float4 ra[NA], rw[NWT];
  auto load_slice = [&](int c) {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = buf_load4(rsi, a_off[j] == OOB ? OOB : a_off[j] + (unsigned)c * 128u);
#pragma unroll
    for (int k = 0; k < NWT; ++k) {
      const int t = TH == 2 ? k : 2 * k + lt;
      rw[k] = buf_load4(rs1, (w_off == OOB || t >= 9) ? OOB : w_off + (unsigned)(t * C + c * 32) * 4u);
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int i = tid + NTH * j;
      if (j < NA - 1 || i < HPIX * 8) {
        if constexpr (BF) {
          uint2 hi, mid, lo;
          if (PREC == 4 && p.in_pair) {
            // pair8 source: piece lq of the pixel's 128-byte slice = plane (lq & 1) of channel group (lq >> 1)
            *reinterpret_cast<float4 *>(Apl + (lq & 1) * APS + (i >> 3) * LDB + bf_slot(i >> 3, lq >> 1)) = ra[j];
            continue;
          }
          split_x4<PREC>(ra[j], kF16ScaleA, hi, mid, lo);
          const int wo = (i >> 3) * LDB + bf_slot(i >> 3, lq >> 1) + (lq & 1) * 4;
          *reinterpret_cast<uint2 *>(Apl + wo) = hi;
          *reinterpret_cast<uint2 *>(Apl + APS + wo) = lo;
          if constexpr (PREC == 2) *reinterpret_cast<uint2 *>(Apl + 2 * APS + wo) = mid;
        } else {
          *reinterpret_cast<float4 *>(Ah + (i >> 3) * LDK + lq * 4) = ra[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NWT; ++k) {
      const int t = TH == 2 ? k : 2 * k + lt;
      if (t >= 9) continue;
      if constexpr (PREC == 4) {
        *reinterpret_cast<float4 *>(Wpl + (lq & 1) * WPS + (t * 32 + lr) * LDB + bf_slot(t * 32 + lr, lq >> 1)) = rw[k];
      } else if constexpr (BF) {
        uint2 hi, mid, lo;
        split_w4<PREC>(rw[k], hi, mid, lo);
        const int wo = (t * 32 + lr) * LDB + bf_slot(t * 32 + lr, lq >> 1) + (lq & 1) * 4;
        *reinterpret_cast<uint2 *>(Wpl + wo) = hi;
        *reinterpret_cast<uint2 *>(Wpl + WPS + wo) = lo;
        if constexpr (PREC == 2) *reinterpret_cast<uint2 *>(Wpl + 2 * WPS + wo) = mid;
      } else {
        *reinterpret_cast<float4 *>(W1s + (t * 32 + lr) * LDK + lq * 4) = rw[k];
      }
    }
  };

  // Two accumulators (even / odd K steps): back-to-back MFMAs on ONE accumulator
  // stall whenever anything else issues between them, alternating chains do not.
  f32x16 acc1, acc1b;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc1b[r] = 0.f; }

  const int frow = lane & 31, fq = lane >> 5;
  const int ry = wave >> 1, rx = (wave & 1) * 32 + frow;  // this lane's pixel inside the tile
  const float *a_base = Ah + (ry * HWD + rx) * LDK + fq * 4;
  const float *b_base = W1s + frow * LDK + fq * 4;

  load_slice(0);
  for (int c = 0; c < TC; ++c) {
    store_slice();
    __syncthreads();
    if (c + 1 < TC) load_slice(c + 1);  // in flight under the 144 MFMAs below
    if constexpr (BF) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int arow = (ry + t / 3) * HWD + rx + (t % 3), brow = t * 32 + frow;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ao = arow * LDB + bf_slot(arow, s * 2 + fq), bo = brow * LDB + bf_slot(brow, s * 2 + fq);
          s16x8 av[NP], bv[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            av[q] = *reinterpret_cast<const s16x8 *>(Apl + q * APS + ao);
            bv[q] = *reinterpret_cast<const s16x8 *>(Wpl + q * WPS + bo);
          }
          if (s == 0) acc1 = mfma_split<PREC>(av, bv, acc1);
          else acc1b = mfma_split<PREC>(av, bv, acc1b);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < (BF ? 0 : 9); ++t) {
      const float *a = a_base + ((t / 3) * HWD + (t % 3)) * LDK;
      const float *bb = b_base + t * 32 * LDK;
#pragma unroll
      for (int s = 0; s < 4; s += 2) {
        const float4 af0 = *reinterpret_cast<const float4 *>(a + s * 8);
        const float4 bf0 = *reinterpret_cast<const float4 *>(bb + s * 8);
        const float4 af1 = *reinterpret_cast<const float4 *>(a + s * 8 + 8);
        const float4 bf1 = *reinterpret_cast<const float4 *>(bb + s * 8 + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(af0, e), elem(bf0, e), acc1, 0, 0, 0);
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(af1, e), elem(bf1, e), acc1b, 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] += acc1b[r];

  // ---- W2 -> LDS (over the dead W1 slice), h = relu(acc1 + b1) -> LDS (over the dead halo)
  f32x16 acc2[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  if constexpr (BF) {
#pragma unroll
    for (int j = 0; j < (TC * 32 + NTH / 8 - 1) / (NTH / 8); ++j) {
      const int n = ln + (NTH / 8) * j;
      if (n >= C) continue;
      uint2 hi, mid, lo;
      const float4 wv = buf_load4(rs2, (unsigned)(n * 32 + lq * 4) * 4u);
      if constexpr (PREC == 4) {
        *reinterpret_cast<float4 *>(Wpl + (lq & 1) * WPS + n * LDB + bf_slot(n, lq >> 1)) = wv;
        continue;
      }
      split_w4<PREC>(wv, hi, mid, lo);
      const int wo = n * LDB + bf_slot(n, lq >> 1) + (lq & 1) * 4;
      *reinterpret_cast<uint2 *>(Wpl + wo) = hi;
      *reinterpret_cast<uint2 *>(Wpl + WPS + wo) = lo;
      if constexpr (PREC == 2) *reinterpret_cast<uint2 *>(Wpl + 2 * WPS + wo) = mid;
    }
    const float b1 = frow < p.R ? p.b1[frow] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
      const float hpre = (PREC >= 3 ? acc1[r] * kF16Unscale : acc1[r]) + b1;
      const float hv = frow < p.R ? (hpre < 0.f ? 0.f : hpre) : 0.f;   // NaN-propagating rectifier (torch.relu)
      const int wo = row * LDB + bf_slot(row, frow >> 3) + (frow & 7);
      if constexpr (PREC >= 3) {
        const float hs_ = hv * kF16ScaleA;
        const _Float16 h0 = (_Float16)hs_;
        Apl[wo] = __builtin_bit_cast(unsigned short, h0);
        Apl[APS + wo] = __builtin_bit_cast(unsigned short, (_Float16)(hs_ - (float)h0));
        continue;
      }
      const __bf16 hh = (__bf16)hv;
      const float r1 = hv - (float)hh;
      if constexpr (PREC == 2) {
        const __bf16 hm = (__bf16)r1;
        const __bf16 hl = (__bf16)(r1 - (float)hm);
        Apl[2 * APS + wo] = __builtin_bit_cast(unsigned short, hm);
        Apl[APS + wo] = __builtin_bit_cast(unsigned short, hl);
      } else {
        Apl[APS + wo] = __builtin_bit_cast(unsigned short, (__bf16)r1);
      }
      Apl[wo] = __builtin_bit_cast(unsigned short, hh);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int arow = wave * 32 + frow;
      const int ao = arow * LDB + bf_slot(arow, s * 2 + fq);
      s16x8 av[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) av[q] = *reinterpret_cast<const s16x8 *>(Apl + q * APS + ao);
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int brow = j * 32 + frow;
        const int bo = brow * LDB + bf_slot(brow, s * 2 + fq);
        s16x8 bv[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) bv[q] = *reinterpret_cast<const s16x8 *>(Wpl + q * WPS + bo);
        acc2[j] = mfma_split<PREC>(av, bv, acc2[j]);
      }
    }
  } else {
  float *W2s = W1s;
#pragma unroll
  for (int j = 0; j < (TC * 32 + NTH / 8 - 1) / (NTH / 8); ++j) {
    const int n = ln + (NTH / 8) * j;
    if (n < C) *reinterpret_cast<float4 *>(W2s + n * LDK + lq * 4) = buf_load4(rs2, (unsigned)(n * 32 + lq * 4) * 4u);
  }
  float *hs = Ah + wave * 32 * LDK;
  {
    const float b1 = frow < p.R ? p.b1[frow] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fq;
      const float hpre = acc1[r] + b1;
      hs[row * LDK + frow] = frow < p.R ? (hpre < 0.f ? 0.f : hpre) : 0.f;
    }
  }
  __syncthreads();

  // ---- GEMM2: [32 px x 32] x [32 x C]
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 af = *reinterpret_cast<const float4 *>(hs + frow * LDK + fq * 4 + s * 8);
    float4 bf[TC];
#pragma unroll
    for (int j = 0; j < TC; ++j)
      bf[j] = *reinterpret_cast<const float4 *>(W2s + (j * 32 + frow) * LDK + fq * 4 + s * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < TC; ++j)  // rotate over the accumulators
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(af, e), elem(bf[j], e), acc2[j], 0, 0, 0);
  }
  }

  // ---- epilogue: + b2 + r, ReLU, store.  Row r of this wave's tile = pixel
  // (y0 + ry, x0 + (wave&1)*32 + row); lanes = 32 consecutive channels.  All
  // residual loads of a channel tile are issued before the first use and
  // out-of-image pixels are dropped by out-of-range buffer offsets: no branches,
  // no load -> wait -> store serialisation.
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.in_bytes, 0x00020000);
  const int gy = y0 + ry;
  const int rowbase = (b * p.H + gy) * p.W;
  unsigned eoff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gx = x0 + (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
    eoff[r] = (gy < p.H && gx < p.W) ? (unsigned)((rowbase + gx) * C + frow) * 4u : OOB;
  }
  // pair8 tensors: channel n of a pixel has its hi piece at byte (n / 8) 32 + (n % 8) 2 and its lo piece 16 bytes on
  const unsigned pair_ch = (unsigned)((frow >> 3) * 32 + (frow & 7) * 2);
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const float b2 = p.b2[j * 32 + frow];
    float res[16];
    if (PREC == 4 && p.in_pair) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned o = eoff[r] == OOB ? OOB : eoff[r] - (unsigned)frow * 4u + (unsigned)j * 128u + pair_ch;
        const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rsi, o, 0, 0);
        const unsigned short l = __builtin_amdgcn_raw_buffer_load_b16(rsi, o == OOB ? OOB : o + 16u, 0, 0);
        res[r] = f16s::pair_value(h, l);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               rsi, eoff[r] == OOB ? OOB : eoff[r] + (unsigned)j * 128u, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = (PREC >= 3 ? acc2[j][r] * kF16Unscale : acc2[j][r]) + b2 + res[r];
      if (p.relu) v = v < 0.f ? 0.f : v;
      res[r] = v;
    }
    if (PREC == 4 && p.out_pair) {
      // a lane of this layout owns one channel of 16 pixels: the pair8 pieces go out as 2-byte stores
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = res[r] * kF16ScaleA;
        const _Float16 h = (_Float16)t;
        const _Float16 l = (_Float16)(t - (float)h);
        const unsigned o = eoff[r] == OOB ? OOB : eoff[r] - (unsigned)frow * 4u + (unsigned)j * 128u + pair_ch;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, h), rso, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, l), rso, o == OOB ? OOB : o + 16u, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, res[r]), rso,
                                              eoff[r] == OOB ? OOB : eoff[r] + (unsigned)j * 128u, 0, 0);
    }
  }
}

template <int TC, int PREC, int TH>
static int launch_res_t(const ResKArgs &a, int B, hipStream_t stream) {
  auto kern = resblock_f32_kernel<TC, PREC, TH>;
  constexpr int HPIX = (TH + 2) * HWD;
  constexpr size_t smem = PREC ? (size_t)(HPIX + 9 * 32) * LDB * (PREC == 2 ? 3 : 2) * sizeof(unsigned short)
                             : (size_t)(HPIX * LDK + 9 * 32 * LDK) * sizeof(float);
  static_assert(9 * 32 >= TC * 32 && HPIX >= TH * 64, "aliased regions must fit");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(resblock)");
    attr_set.mark();
  }
  const double C = a.C, R = a.R, M = (double)B * a.H * a.W;
  prof::Scope scope(prof::K_RESBLOCK, 2.0 * M * R * 9 * C + 2.0 * M * C * R,
                    4.0 * (2.0 * M * C + 10.0 * C * R), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, B), dim3(TH * 128), smem, stream, a);
  return check_launch("resblock_f32");
}

// tile height: 8-wave 4 x 64 tiles for the six-term variant (ISI_RES_TH = 2 / 4 overrides, for measurements)
template <int TC, int PREC>
static int launch_res(const ResKArgs &a, int B, hipStream_t stream) {
  const int forced = knobs().res_th;
  const int th = forced == 2 || forced == 4 ? forced : (PREC == 2 ? 4 : 2);
  return th == 4 ? launch_res_t<TC, PREC, 4>(a, B, stream) : launch_res_t<TC, PREC, 2>(a, B, stream);
}

bool resblock_fusable(int C, int R) { return C % 32 == 0 && C >= 32 && C <= 128 && R >= 1 && R <= 32; }

// in/out: dense channels-last [B,H,W,C].  w1: packed 3x3 weight [R][9C]; w2: packed 1x1 weight [C][32].
int resblock_f32(const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                 float *out, int B, int H, int W, int C, int R, int relu, hipStream_t stream, float *twin, float *hidden) {
  if (!in || !w1 || !b1 || !w2 || !b2 || !out) return invalid("resblock: null pointer");
  if ((twin || hidden) && !((relu & ISI_CONV_IN0_PAIR) && (relu & ISI_CONV_F16X3) && (relu & ISI_CONV_W16) &&
                            !(relu & ISI_CONV_BF16X6) && resblock_pair_preferred(B, H, W, C, R)))
    return unsupported("resblock: the training side outputs (fp32 twin, hidden activation) come from the pair kernel only");
  if (B <= 0 || H <= 0 || W <= 0) return invalid("resblock: bad shape");
  if (!resblock_fusable(C, R)) return unsupported("resblock: need C % 32 == 0, C <= 128, R <= 32");
  const int64_t elems = (int64_t)B * H * W * C;
  if (elems > ((int64_t)1 << 30)) return unsupported("resblock: tensor spans 4 GiB or more");
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15)
    return invalid("resblock: pointers must be 16-byte aligned");
  if (B > 65535 || (H + 1) / 2 > 65535) return unsupported("resblock: grid too large");
  ResKArgs a;
  a.in = in; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.out = out;
  a.in_bytes = (unsigned)(elems * 4);
  a.w1_bytes = (unsigned)((size_t)R * 9 * C * 4);  // packed [R][9C], 9C % 32 == 0
  a.w2_bytes = (unsigned)((size_t)C * 32 * 4);
  a.C = C; a.R = R; a.H = H; a.W = W; a.relu = relu & 1;
  a.in_pair = (relu & ISI_CONV_IN0_PAIR) ? 1 : 0;
  a.out_pair = (relu & ISI_CONV_OUT_PAIR) ? 1 : 0;
  if ((a.in_pair || a.out_pair) && (!((relu & ISI_CONV_F16X3) && (relu & ISI_CONV_W16)) || (relu & ISI_CONV_BF16X6)))
    return unsupported("resblock: pair formats need ISI_CONV_F16X3 | ISI_CONV_W16");
#define ISI_RES(PREC)                                          \
  switch (C / 32) {                                            \
    case 1: return launch_res<1, PREC>(a, B, stream);          \
    case 2: return launch_res<2, PREC>(a, B, stream);          \
    case 3: return launch_res<3, PREC>(a, B, stream);          \
    default: return launch_res<4, PREC>(a, B, stream);         \
  }
  if (relu & ISI_CONV_BF16X6) { ISI_RES(2) }
  if ((relu & ISI_CONV_F16X3) && (relu & ISI_CONV_W16)) {   // split-f16 pair copies behind the fp32 weights
    a.w1 = w1 + (size_t)R * 9 * C;
    a.w2 = w2 + (size_t)C * 32;
    // pair-format input: the LDS-DMA kernel (resblock_pair_f16.hip)
    if (a.in_pair && resblock_pair_preferred(B, H, W, C, R) && !(relu & ISI_CONV_BF16X6))
      return resblock_pair_f16(in, a.w1, b1, a.w2, b2, out, B, H, W, C, relu & 1, a.out_pair, stream, twin, hidden);
    ISI_RES(4)
  }
  if (relu & ISI_CONV_F16X3) { ISI_RES(3) }
  if (relu & ISI_CONV_BF16X3) { ISI_RES(1) }
  ISI_RES(0)
#undef ISI_RES
}

}  // namespace isi
