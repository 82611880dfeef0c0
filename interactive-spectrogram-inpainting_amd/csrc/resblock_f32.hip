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
#include "isi_common.h"
#include "prof.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ResKArgs {
  const float *in, *w1, *b1, *w2, *b2;
  float *out;
  unsigned in_bytes, w1_bytes, w2_bytes;
  int C, R, H, W, relu;
};

namespace {
constexpr int LDK = 36;             // padded LDS row (floats)
constexpr int TH = 2, TW = 64;      // output tile: 2 rows x 64 pixels = 128 GEMM rows
constexpr int HH = TH + 2, HWD = TW + 2, HPIX = HH * HWD;  // halo 4 x 66 pixels
constexpr int NA = (HPIX * 8 + 255) / 256;                 // halo quads per thread (9)
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
constexpr int LDB = 40;  // bf16 plane row (elements): 80 B
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
__device__ __forceinline__ f32x16 mfma3(const s16x8 ah, const s16x8 al, const s16x8 bh, const s16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
  return acc;
}

template <int TC, bool BF = false>  // TC = C / 32
__global__ __launch_bounds__(256) void resblock_f32_kernel(const ResKArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ah = smem;                     // [HPIX][LDK]      halo slice   (later: h, [128][LDK])
  float *W1s = Ah + HPIX * LDK;         // [9][32][LDK]     W1 slice     (later: W2, [C][LDK])
  // bf16x3: bf16 planes instead -- halo hi/lo [HPIX][LDB], W1 slice hi/lo [9*32][LDB]
  unsigned short *Ahi = reinterpret_cast<unsigned short *>(smem);
  unsigned short *Alo = Ahi + HPIX * LDB;
  unsigned short *Whi = Alo + HPIX * LDB;
  unsigned short *Wlo = Whi + 9 * 32 * LDB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int C = TC * 32;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;

  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w2), 0, p.w2_bytes, 0x00020000);

  const int lq = tid & 7, ln = tid >> 3;
  // byte offsets (channel slice 0) of this thread's halo quads / W1 quads
  unsigned a_off[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int i = tid + 256 * j;
    const int pix = i >> 3;
    const int hy = pix / HWD, hx = pix - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = pix < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    a_off[j] = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * C + lq * 4) * 4u : OOB;
  }
  const unsigned w_off = ln < p.R ? (unsigned)(ln * 9 * C + lq * 4) * 4u : OOB;

  float4 ra[NA], rw[9];
  auto load_slice = [&](int c) {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = buf_load4(rsi, a_off[j] == OOB ? OOB : a_off[j] + (unsigned)c * 128u);
#pragma unroll
    for (int t = 0; t < 9; ++t)
      rw[t] = buf_load4(rs1, w_off == OOB ? OOB : w_off + (unsigned)(t * C + c * 32) * 4u);
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int i = tid + 256 * j;
      if (j < NA - 1 || i < HPIX * 8) {
        if constexpr (BF) {
          uint2 hi, lo;
          split_bf16x4(ra[j], hi, lo);
          *reinterpret_cast<uint2 *>(Ahi + (i >> 3) * LDB + lq * 4) = hi;
          *reinterpret_cast<uint2 *>(Alo + (i >> 3) * LDB + lq * 4) = lo;
        } else {
          *reinterpret_cast<float4 *>(Ah + (i >> 3) * LDK + lq * 4) = ra[j];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if constexpr (BF) {
        uint2 hi, lo;
        split_bf16x4(rw[t], hi, lo);
        *reinterpret_cast<uint2 *>(Whi + (t * 32 + ln) * LDB + lq * 4) = hi;
        *reinterpret_cast<uint2 *>(Wlo + (t * 32 + ln) * LDB + lq * 4) = lo;
      } else {
        *reinterpret_cast<float4 *>(W1s + (t * 32 + ln) * LDK + lq * 4) = rw[t];
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
        const int ao = ((ry + t / 3) * HWD + rx + (t % 3)) * LDB + fq * 8;
        const int bo = (t * 32 + frow) * LDB + fq * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const s16x8 ah = *reinterpret_cast<const s16x8 *>(Ahi + ao + s * 16);
          const s16x8 al = *reinterpret_cast<const s16x8 *>(Alo + ao + s * 16);
          const s16x8 bh = *reinterpret_cast<const s16x8 *>(Whi + bo + s * 16);
          const s16x8 bl = *reinterpret_cast<const s16x8 *>(Wlo + bo + s * 16);
          if (s == 0) acc1 = mfma3(ah, al, bh, bl, acc1);
          else acc1b = mfma3(ah, al, bh, bl, acc1b);
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
    for (int j = 0; j < TC; ++j) {
      const int n = ln + 32 * j;
      uint2 hi, lo;
      split_bf16x4(buf_load4(rs2, (unsigned)(n * 32 + lq * 4) * 4u), hi, lo);
      *reinterpret_cast<uint2 *>(Whi + n * LDB + lq * 4) = hi;
      *reinterpret_cast<uint2 *>(Wlo + n * LDB + lq * 4) = lo;
    }
    const float b1 = frow < p.R ? p.b1[frow] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
      const float hv = frow < p.R ? fmaxf(acc1[r] + b1, 0.f) : 0.f;
      const __bf16 hh = (__bf16)hv;
      const __bf16 hl = (__bf16)(hv - (float)hh);
      Ahi[row * LDB + frow] = __builtin_bit_cast(unsigned short, hh);
      Alo[row * LDB + frow] = __builtin_bit_cast(unsigned short, hl);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ao = (wave * 32 + frow) * LDB + s * 16 + fq * 8;
      const s16x8 ah = *reinterpret_cast<const s16x8 *>(Ahi + ao);
      const s16x8 al = *reinterpret_cast<const s16x8 *>(Alo + ao);
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int bo = (j * 32 + frow) * LDB + s * 16 + fq * 8;
        acc2[j] = mfma3(ah, al, *reinterpret_cast<const s16x8 *>(Whi + bo), *reinterpret_cast<const s16x8 *>(Wlo + bo),
                        acc2[j]);
      }
    }
  } else {
  float *W2s = W1s;
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int n = ln + 32 * j;
    *reinterpret_cast<float4 *>(W2s + n * LDK + lq * 4) = buf_load4(rs2, (unsigned)(n * 32 + lq * 4) * 4u);
  }
  float *hs = Ah + wave * 32 * LDK;
  {
    const float b1 = frow < p.R ? p.b1[frow] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fq;
      hs[row * LDK + frow] = frow < p.R ? fmaxf(acc1[r] + b1, 0.f) : 0.f;
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
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const float b2 = p.b2[j * 32 + frow];
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             rsi, eoff[r] == OOB ? OOB : eoff[r] + (unsigned)j * 128u, 0, 0));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc2[j][r] + b2 + res[r];
      if (p.relu) v = fmaxf(v, 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rso,
                                            eoff[r] == OOB ? OOB : eoff[r] + (unsigned)j * 128u, 0, 0);
    }
  }
}

template <int TC, bool BF>
static int launch_res(const ResKArgs &a, int B, hipStream_t stream) {
  auto kern = resblock_f32_kernel<TC, BF>;
  constexpr size_t smem = BF ? (size_t)(HPIX + 9 * 32) * LDB * 2 * sizeof(unsigned short)
                             : (size_t)(HPIX * LDK + 9 * 32 * LDK) * sizeof(float);
  static_assert(9 * 32 >= TC * 32 && HPIX >= 128, "aliased regions must fit");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(resblock)");
    attr_set = true;
  }
  const double C = a.C, R = a.R, M = (double)B * a.H * a.W;
  prof::Scope scope(prof::K_RESBLOCK, 2.0 * M * R * 9 * C + 2.0 * M * C * R,
                    4.0 * (2.0 * M * C + 10.0 * C * R), stream);
  hipLaunchKernelGGL(kern, dim3((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, B), dim3(256), smem, stream, a);
  return check_launch("resblock_f32");
}

bool resblock_fusable(int C, int R) { return C % 32 == 0 && C >= 32 && C <= 128 && R >= 1 && R <= 32; }

// in/out: dense channels-last [B,H,W,C].  w1: packed 3x3 weight [R][9C]; w2: packed 1x1 weight [C][32].
int resblock_f32(const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                 float *out, int B, int H, int W, int C, int R, int relu, hipStream_t stream) {
  if (!in || !w1 || !b1 || !w2 || !b2 || !out) return invalid("resblock: null pointer");
  if (B <= 0 || H <= 0 || W <= 0) return invalid("resblock: bad shape");
  if (!resblock_fusable(C, R)) return unsupported("resblock: need C % 32 == 0, C <= 128, R <= 32");
  const int64_t elems = (int64_t)B * H * W * C;
  if (elems > ((int64_t)1 << 30)) return unsupported("resblock: tensor spans 4 GiB or more");
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15)
    return invalid("resblock: pointers must be 16-byte aligned");
  if (B > 65535 || (H + TH - 1) / TH > 65535) return unsupported("resblock: grid too large");
  ResKArgs a;
  a.in = in; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.out = out;
  a.in_bytes = (unsigned)(elems * 4);
  a.w1_bytes = (unsigned)((size_t)R * 9 * C * 4);  // packed [R][9C], 9C % 32 == 0
  a.w2_bytes = (unsigned)((size_t)C * 32 * 4);
  a.C = C; a.R = R; a.H = H; a.W = W; a.relu = relu & 1;
  if (relu & ISI_CONV_BF16X3) {
    switch (C / 32) {
      case 1: return launch_res<1, true>(a, B, stream);
      case 2: return launch_res<2, true>(a, B, stream);
      case 3: return launch_res<3, true>(a, B, stream);
      default: return launch_res<4, true>(a, B, stream);
    }
  }
  switch (C / 32) {
    case 1: return launch_res<1, false>(a, B, stream);
    case 2: return launch_res<2, false>(a, B, stream);
    case 3: return launch_res<3, false>(a, B, stream);
    default: return launch_res<4, false>(a, B, stream);
  }
}

}  // namespace isi
