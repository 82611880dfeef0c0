// ConvTranspose2d(kernel 4, stride 2, padding 1) with very few output channels
// (the decoder's last layer, C/2 -> in_channel; reference
// vqvae/encoder_decoder.py:204-207), exact fp32 on the matrix pipe.
//
// With N = Cout <= 4 the 4-phase implicit GEMM wastes > 90 % of every MFMA.  A
// transposed convolution is the adjoint of a convolution, so instead
//
//   Y'[m][(ky,kx,co)] = sum_ci x[m][ci] * W[ci][co][ky][kx]        (a GEMM: pixels x 16*Cout, K = Cin,
//                                                                   no wasted columns for Cout = 2)
//   out[2m-1+ky, 2n-1+kx, co] += Y'[(m,n)][(ky,kx,co)]              (col2im)
//
// and the scatter-add becomes a gather because a workgroup computes Y' for its
// TH x 32 patch of input pixels (TH = 8, or 4 for short maps) PLUS a one-pixel halo
// (10 x 34 = 340 rows, 11 MFMA row tiles), parks Y' in LDS and then every output pixel
// of its 2TH x 64 output patch sums its four contributions.  Out-of-image halo pixels are zero rows (zero
// padding).  HBM: the input (the largest activation of the network) is read once
// plus halo overlap served by L2; the output is written once, coalesced along W.
//
// Weight layout (isi_pack_convT_k4s2_weight_f32 when this kernel applies):
//   wn[(ky*4+kx)*Cout + co][ci].
#include <algorithm>
#include <cstdlib>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvTSmallArgs {
  const float *in, *wn, *bias;
  float *out;
  unsigned in_bytes;
  int H, W, Cin, Cout, relu;
  int sn, sc, sh, sw, vec;  // source element strides; vec: channel-contiguous, 16-B aligned quads
  int on, oc, oh, ow;       // output element strides
};

namespace {
constexpr int TW = 32, HW_ = TW + 2;   // input-pixel tile width, halo row
constexpr int LDX = 36;                // staged input row: one 32-channel slice + 4 (conflict-free b128)
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ float elem4(const float4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}
}  // namespace

// TH: input rows per workgroup; NT: 32-wide column tiles of Y' (16 * Cout <= 32 * NT); CIN in {32, 64}.
// The input is staged one 32-channel slice at a time (LDS ~41 KB for TH = 4, NT = 1: three
// workgroups per CU) and the next slice is prefetched into registers under the MFMAs.
template <int TH, int NT, int CIN>
__global__ __launch_bounds__(256) void convT_k4s2_small_kernel(const ConvTSmallArgs p) {
  constexpr int HPIX = (TH + 2) * HW_;          // halo pixels
  constexpr int RT = (HPIX + 31) / 32;          // MFMA row tiles
  constexpr int ROWS = RT * 32;
  constexpr int TPW = (RT + 3) / 4;             // row tiles per wave
  constexpr int NS = CIN / 32;                  // channel slices
  constexpr int NLD = ROWS * 8 / 256;           // float4 staged per thread and slice
  constexpr int LDW = CIN + 4;
  constexpr int LDY = NT * 32 + 1;
  constexpr int XFLOATS = ROWS * LDX > ROWS * LDY ? ROWS * LDX : ROWS * LDY;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *xs = sm;                             // [ROWS][LDX]   (later: Y' [ROWS][LDY])
  float *ws = sm + XFLOATS;                   // [NT*32][LDW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
  const __amdgpu_buffer_rsrc_t rsi =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);

  // ---- this thread's staging slots: halo pixel (i >> 3), quad (i & 7) of the slice
  unsigned off[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = tid + 256 * it;
    const int pix = i >> 3, q = i & 7;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = pix < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    off[it] = ok ? (unsigned)(b * p.sn + gy * p.sh + gx * p.sw + q * 4 * p.sc) : OOB;  // elements
  }
  i32x4 stg[NLD];
  auto load_slice = [&](int s) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      if (p.vec) {
        stg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsi, off[it] == OOB ? OOB : (off[it] + s * 32) * 4u, 0, 0);
      } else {  // NCHW / strided source: element loads
#pragma unroll
        for (int e = 0; e < 4; ++e)
          stg[it][e] = __builtin_amdgcn_raw_buffer_load_b32(
              rsi, off[it] == OOB ? OOB : (off[it] + (s * 32 + e) * p.sc) * 4u, 0, 0);
      }
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = tid + 256 * it;
      *reinterpret_cast<i32x4 *>(xs + (i >> 3) * LDX + (i & 7) * 4) = stg[it];
    }
  };
  load_slice(0);
  const int N = 16 * p.Cout;
  for (int i = tid; i < NT * 32 * (CIN / 4); i += 256) {
    const int n = i / (CIN / 4), q = i - n * (CIN / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = *reinterpret_cast<const float4 *>(p.wn + (size_t)n * CIN + q * 4);
    *reinterpret_cast<float4 *>(ws + n * LDW + q * 4) = v;
  }
  store_slice();
  __syncthreads();

  // ---- Y' = X W^T : wave w owns row tiles w, w + 4, ...
  const int frow = lane & 31, fq = lane >> 5;
  f32x16 acc[TPW][NT];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) load_slice(s + 1);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      float4 bf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bf[j] = *reinterpret_cast<const float4 *>(ws + (j * 32 + frow) * LDW + s * 32 + kq * 8 + fq * 4);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int rt = wave + 4 * i;
        if (rt < RT) {  // wave-uniform
          const float4 af = *reinterpret_cast<const float4 *>(xs + (rt * 32 + frow) * LDX + kq * 8 + fq * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(elem4(af, e), elem4(bf[j], e), acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // every wave is done reading this slice
    if (s + 1 < NS) {
      store_slice();
      __syncthreads();
    }
  }
  float *ys = xs;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int rt = wave + 4 * i;
    if (rt < RT) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ys[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq) * LDY + j * 32 + frow] = acc[i][j][r];
    }
  }
  __syncthreads();

  // ---- col2im as a gather: thread -> output column ox_l, rows oyb + 4 rr
  const int ox_l = tid & 63, oyb = tid >> 6;
  const int ox = 2 * x0 + ox_l;
  // column taps: kx = (ox_l + 1) % 2 + 2 jx ; halo column hc = (ox_l + 1 - kx) / 2 + 1
  int kxs[2], hcs[2];
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    kxs[jx] = ((ox_l + 1) & 1) + 2 * jx;
    hcs[jx] = (ox_l + 1 - kxs[jx]) / 2 + 1;
  }
  if (ox >= 2 * p.W) return;
#pragma unroll
  for (int rr = 0; rr < TH / 2; ++rr) {
    const int oy_l = oyb + 4 * rr;
    const int oy = 2 * y0 + oy_l;
    if (oy >= 2 * p.H) continue;
    for (int co = 0; co < p.Cout; ++co) {
      float v = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int jy = 0; jy < 2; ++jy) {
        const int ky = ((oy_l + 1) & 1) + 2 * jy;
        const int hr = (oy_l + 1 - ky) / 2 + 1;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx)
          v += ys[(hr * HW_ + hcs[jx]) * LDY + (ky * 4 + kxs[jx]) * p.Cout + co];
      }
      if (p.relu) v = v < 0.f ? 0.f : v;   // like torch.relu, NaN stays NaN (fmaxf would drop it)
      p.out[b * p.on + co * p.oc + oy * p.oh + ox * p.ow] = v;
    }
  }
}

// ---- the same layer inside the split-f16 PAIR pipeline (round 3).  The exact-fp32 kernel above spends 45 us of its
// 106 us (B = 64, 64 -> 2 at 64 x 256) on the fp32 matrix pipe (32 MFMAs of 16 passes per 32-pixel row tile) and
// stages through registers; its input is the largest activation of the network.  Here the input arrives in the pair
// format (the transposed convolution in front of it writes it), so
//   * a tile's 6 x 34 halo pixels x 64 channels (256 bytes per pixel: 16 pieces) go global -> LDS by LDS-DMA in ONE
//     stage -- no K loop, no staging registers, no conversion;  piece c of halo row r sits at position c ^ (r & 15)
//     (swizzle on the DMA's source side: conflict-free ds_read_b128);
//   * Y' = X W^T runs as three-term split-f16 products (12 MFMAs of 8 passes per row tile: 1/10 of the matrix time),
//     the weight fragments straight from the blocked pair copy of the packed weight (ISI_CONV_W16);
//   * Y' overlays the staged tile in LDS and the col2im gather + NCHW store are those of the kernel above.
// 52 KB of LDS: three workgroups per CU hide the one memory round trip a workgroup makes.  The kernel is bound by
// reading its input once (268 MB at B = 64).
namespace {
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dma16_small(const unsigned lds_addr, const unsigned voff, const i32x4 rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
}  // namespace

template <int TH>
__global__ __launch_bounds__(256) void convT_k4s2_small_pair_kernel(const ConvTSmallArgs p) {
  constexpr int CIN = 64, ROWB = CIN * 4;       // bytes per pixel: 8 groups x {hi[8] | lo[8]}
  constexpr int HPIX = (TH + 2) * HW_;          // halo pixels
  constexpr int RT = (HPIX + 31) / 32;          // MFMA row tiles
  constexpr int ROWS = RT * 32;
  constexpr int TPW = (RT + 3) / 4;             // row tiles per wave
  static_assert(HPIX % 4 == 0, "whole DMAs");
  constexpr int NDMA = HPIX / 4;                // 1-KiB DMAs (4 rows each): only the halo pixels are staged (52 KB at
                                                // TH = 4: three workgroups per CU); the last row tile's surplus rows
                                                // re-read the last pixel and produce Y' rows nobody gathers
  constexpr int PER = (NDMA + 3) / 4;           // per wave
  constexpr int LDY = 33;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  char *xs = reinterpret_cast<char *>(sm);      // [ROWS][256 B]   (later: Y' [ROWS][LDY] floats)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: consecutive workgroup ids run on different XCDs (id % 8), each with its own L2; an XCD gets a
  // contiguous range of tiles so that the halo pixels shared by neighbouring tiles are re-read from ITS L2, not from HBM
  int x0, y0, b;
  {
    const int ntx = gridDim.x, nty = gridDim.y, nt = ntx * nty * (int)gridDim.z;
    const int id = (int)blockIdx.x + ntx * ((int)blockIdx.y + nty * (int)blockIdx.z);
    const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    x0 = (t % ntx) * TW;
    y0 = ((t / ntx) % nty) * TH;
    b = t / (ntx * nty);
  }
  const unsigned long long ib = (unsigned long long)p.in;
  const i32x4 rsi = i32x4{(int)(unsigned)ib, (int)((unsigned)(ib >> 32) & 0xffffu), (int)p.in_bytes, 0x00020000};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)xs;

  // ---- the whole tile in one stage: DMA d of this wave covers halo rows 4 d .. 4 d + 3
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int d = wave + 4 * q;                 // uniform
    if (d < NDMA) {
      const int row = 4 * d + (lane >> 4);
      const int hy = row / HW_, hx = row - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = row < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const unsigned piece = (unsigned)(((lane & 15) ^ (row & 15)) * 16);
      const unsigned vo = ok ? (unsigned)(b * p.sn + gy * p.sh + gx * p.sw) * 4u + piece : 0x7FFFFFF0u;
      dma16_small(lds0 + (unsigned)(d * 1024), vo, rsi);
    }
  }
  // weight fragments (B operand: column n = (ky, kx, co) = frow, k-block kb of k-step s): blocked pair copy, 64 x 4 B
  // per row: group 2 s + kb at + 32 (2 s + kb), hi piece first
  const int frow = lane & 31, kb = lane >> 5;
  const int N = 16 * p.Cout;
  s16x8 wh[4], wl[4];
  {
    const uint4 *w16 = reinterpret_cast<const uint4 *>(p.wn + (size_t)N * CIN);      // pair copy behind the packed weight
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
      if (frow < N) { h = w16[frow * 16 + 2 * (2 * s_ + kb)]; l = w16[frow * 16 + 2 * (2 * s_ + kb) + 1]; }
      wh[s_] = __builtin_bit_cast(s16x8, h);
      wl[s_] = __builtin_bit_cast(s16x8, l);
    }
  }
  // (requested now: at the end it would be a dependent L2 round trip in front of the stores)
  float bias_v[2];
#pragma unroll
  for (int co = 0; co < 2; ++co) bias_v[co] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- Y' = X W^T : wave w owns row tiles w, w + 4, ...; lo terms first, hi.hi last
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int rt = wave + 4 * i;
    if (rt < RT) {  // wave-uniform
      const int row = min(rt * 32 + frow, HPIX - 1);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int pc = 2 * (2 * s_ + kb);
        const s16x8 ah = *reinterpret_cast<const s16x8 *>(xs + row * ROWB + ((pc ^ (row & 15)) << 4));
        const s16x8 al = *reinterpret_cast<const s16x8 *>(xs + row * ROWB + (((pc + 1) ^ (row & 15)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_, al), __builtin_bit_cast(f16x8_, wh[s_]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_, ah), __builtin_bit_cast(f16x8_, wl[s_]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_, ah), __builtin_bit_cast(f16x8_, wh[s_]), acc[i], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every wave is done reading the tile: Y' overlays it
  float *ys = sm;
  constexpr float kUn = 1.f / (4.f * 1024.f);   // the pair scales (split_f16.h), undone exactly
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int rt = wave + 4 * i;
    if (rt < RT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ys[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb) * LDY + frow] = acc[i][r] * kUn;
    }
  }
  __syncthreads();

  // ---- col2im as a gather (as in the fp32 kernel): thread -> output column ox_l, rows oyb + 4 rr
  const int ox_l = tid & 63, oyb = tid >> 6;
  const int ox = 2 * x0 + ox_l;
  int kxs[2], hcs[2];
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    kxs[jx] = ((ox_l + 1) & 1) + 2 * jx;
    hcs[jx] = (ox_l + 1 - kxs[jx]) / 2 + 1;
  }
  if (ox >= 2 * p.W) return;
#pragma unroll
  for (int rr = 0; rr < TH / 2; ++rr) {
    const int oy_l = oyb + 4 * rr;
    const int oy = 2 * y0 + oy_l;
    if (oy >= 2 * p.H) continue;
#pragma unroll
    for (int co = 0; co < 2; ++co) {
      if (co >= p.Cout) continue;
      float v = bias_v[co];
#pragma unroll
      for (int jy = 0; jy < 2; ++jy) {
        const int ky = ((oy_l + 1) & 1) + 2 * jy;
        const int hr = (oy_l + 1 - ky) / 2 + 1;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx)
          v += ys[(hr * HW_ + hcs[jx]) * LDY + (ky * 4 + kxs[jx]) * p.Cout + co];
      }
      if (p.relu) v = v < 0.f ? 0.f : v;   // like torch.relu, NaN stays NaN
      p.out[b * p.on + co * p.oc + oy * p.oh + ox * p.ow] = v;
    }
  }
}

// ---- the col2im half alone (round 3, "decoder tail"): when the transposed convolution in FRONT of the few-channel
// layer has already projected its pixels onto this layer's taps (convT_pair_f16.hip, YP mode), the input here is
// Y' [B][H][W][32] fp32 -- 128 bytes per pixel instead of the 256-byte activation -- and what is left is
//   out[b][co][2 y - 1 + ky][2 x - 1 + kx] = bias[co] + sum of the four Y'[b][y][x][(ky, kx, co)] that land there.
// A tile's (TH + 2) x 34 pixels come in by LDS-DMA (16-byte piece c of row r at position c ^ (r & 7)) and every output
// pixel gathers its four contributions in the fixed order of the kernels above.  Purely memory-bound.
template <int TH>
__global__ __launch_bounds__(256) void convT_gather_kernel(const ConvTSmallArgs p) {
  constexpr int HPIX = (TH + 2) * HW_;          // halo pixels
  static_assert(HPIX % 8 == 0, "whole DMAs");
  constexpr int NDMA = HPIX / 8;                // 1-KiB DMAs (8 rows of 128 B)
  constexpr int PER = (NDMA + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int x0, y0, b;
  {
    const int ntx = gridDim.x, nty = gridDim.y, nt = ntx * nty * (int)gridDim.z;
    const int id = (int)blockIdx.x + ntx * ((int)blockIdx.y + nty * (int)blockIdx.z);
    const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    x0 = (t % ntx) * TW;
    y0 = ((t / ntx) % nty) * TH;
    b = t / (ntx * nty);
  }
  const unsigned long long ib = (unsigned long long)p.in;
  const i32x4 rsi = i32x4{(int)(unsigned)ib, (int)((unsigned)(ib >> 32) & 0xffffu), (int)p.in_bytes, 0x00020000};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int d = wave + 4 * q;                 // uniform
    if (d < NDMA) {
      const int row = 8 * d + (lane >> 3);
      const int hy = row / HW_, hx = row - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const unsigned piece = (unsigned)(((lane & 7) ^ (row & 7)) * 16);
      const unsigned vo = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * 32) * 4u + piece : 0x7FFFFFF0u;
      dma16_small(lds0 + (unsigned)(d * 1024), vo, rsi);
    }
  }
  float bias_v[2];
#pragma unroll
  for (int co = 0; co < 2; ++co) bias_v[co] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float *ys = sm;
  auto yv = [&](const int row, const int c) { return ys[row * 32 + ((((c >> 2) ^ (row & 7)) << 2) | (c & 3))]; };

  const int ox_l = tid & 63, oyb = tid >> 6;
  const int ox = 2 * x0 + ox_l;
  int kxs[2], hcs[2];
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    kxs[jx] = ((ox_l + 1) & 1) + 2 * jx;
    hcs[jx] = (ox_l + 1 - kxs[jx]) / 2 + 1;
  }
  if (ox >= 2 * p.W) return;
#pragma unroll
  for (int rr = 0; rr < TH / 2; ++rr) {
    const int oy_l = oyb + 4 * rr;
    const int oy = 2 * y0 + oy_l;
    if (oy >= 2 * p.H) continue;
#pragma unroll
    for (int co = 0; co < 2; ++co) {
      if (co >= p.Cout) continue;
      float v = bias_v[co];
#pragma unroll
      for (int jy = 0; jy < 2; ++jy) {
        const int ky = ((oy_l + 1) & 1) + 2 * jy;
        const int hr = (oy_l + 1 - ky) / 2 + 1;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) v += yv(hr * HW_ + hcs[jx], (ky * 4 + kxs[jx]) * p.Cout + co);
      }
      if (p.relu) v = v < 0.f ? 0.f : v;   // like torch.relu, NaN stays NaN
      p.out[b * p.on + co * p.oc + oy * p.oh + ox * p.ow] = v;
    }
  }
}

__global__ void pack_convT_small_kernel(const float *__restrict__ w, float *__restrict__ out, int Cin,
                                        int Cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * Cin * Cout) return;
  const int ci = i % Cin, n = i / Cin;  // n = (ky*4 + kx)*Cout + co
  const int co = n % Cout, t = n / Cout;
  const int ky = t >> 2, kx = t & 3;
  out[i] = w[((ci * Cout + co) * 4 + ky) * 4 + kx];
}

bool convT_small_applicable(int Cin, int Cout) {
  return Cout >= 1 && Cout <= 4 && (Cin == 32 || Cin == 64);
}
// the pair-pipeline form: 64 pair-format input channels, 16 Cout <= 32 columns of Y'
bool convT_small_pair_ok(int Cin, int Cout) { return Cin == 64 && Cout >= 1 && Cout <= 2; }

int pack_convT_small_weight_f32(const float *w, float *packed, int Cin, int Cout, hipStream_t stream) {
  const int total = 16 * Cin * Cout;
  hipLaunchKernelGGL(pack_convT_small_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, packed,
                     Cin, Cout);
  return check_launch("pack_convT_small_weight_f32");
}

template <int TH, int NT, int CIN>
static int launch_small(const ConvTSmallArgs &a, int B, hipStream_t stream) {
  auto kern = convT_k4s2_small_kernel<TH, NT, CIN>;
  constexpr int ROWS = (((TH + 2) * HW_ + 31) / 32) * 32;
  // staged slice or the Y' overlay ([ROWS][NT*32+1]), whichever is larger, + the weights
  const size_t smem = ((size_t)ROWS * std::max(LDX, NT * 32 + 1) + (size_t)NT * 32 * (CIN + 4)) * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem) != hipSuccess)
    return check_launch("hipFuncSetAttribute(convT_small)");
  if ((a.H + TH - 1) / TH > 65535) return unsupported("convT_small: grid too large");
  const double M = (double)B * a.H * a.W;
  prof::Scope scope(prof::K_CONVT_SMALL, 2.0 * M * 16 * a.Cin * a.Cout,
                    4.0 * (M * a.Cin + 4.0 * M * a.Cout + 16.0 * a.Cin * a.Cout), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, B), dim3(256), smem, stream, a);
  return check_launch("convT_k4s2_small_f32");
}

template <int TH>
static int dispatch_small(const ConvTSmallArgs &a, int B, hipStream_t stream) {
  const bool one = 16 * a.Cout <= 32;
  if (a.Cin == 32) return one ? launch_small<TH, 1, 32>(a, B, stream) : launch_small<TH, 2, 32>(a, B, stream);
  return one ? launch_small<TH, 1, 64>(a, B, stream) : launch_small<TH, 2, 64>(a, B, stream);
}

template <int TH>
static int launch_small_pair(const ConvTSmallArgs &a, int B, hipStream_t stream) {
  auto kern = convT_k4s2_small_pair_kernel<TH>;
  constexpr int HPIX = (TH + 2) * HW_, ROWS = ((HPIX + 31) / 32) * 32;
  constexpr size_t smem = (size_t)HPIX * 256;   // the staged tile; Y' [ROWS][33] floats overlays it
  static_assert((size_t)ROWS * 33 * 4 <= smem, "Y' overlay");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(convT_small_pair)");
    attr_set.mark();
  }
  if ((a.H + TH - 1) / TH > 65535) return unsupported("convT_small: grid too large");
  const double M = (double)B * a.H * a.W;
  prof::Scope scope(prof::K_CONVT_SMALL, 2.0 * M * 16 * a.Cin * a.Cout,
                    4.0 * (M * a.Cin + 4.0 * M * a.Cout + 16.0 * a.Cin * a.Cout), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, B), dim3(256), smem, stream, a);
  return check_launch("convT_k4s2_small_pair");
}

// pair-format input (dense channels-last [B,H,W,64]), fp32 output with arbitrary strides; wn = packed weight followed by
// its blocked pair copy (pack_convT_k4s2_weight + ISI_CONV_W16)
int convT_k4s2_small_pair_f16(const float *in, const float *wn, const float *bias, float *out, int B, int H, int W,
                              int Cin, int Cout, int on, int oc, int oh, int ow, int relu, hipStream_t stream) {
  if (!convT_small_pair_ok(Cin, Cout)) return unsupported("convT_small_pair: need Cin == 64 and Cout <= 2");
  if (B > 65535) return unsupported("convT_small: grid too large");
  const int64_t in_elems = (int64_t)B * H * W * Cin;
  if (in_elems * 4 >= 0x70000000ll) return unsupported("convT_small_pair: tensor spans 1.75 GiB or more");
  ConvTSmallArgs a;
  a.in = in; a.wn = wn; a.bias = bias; a.out = out;
  a.in_bytes = (unsigned)(in_elems * 4);
  a.sn = H * W * Cin; a.sc = 1; a.sh = W * Cin; a.sw = Cin; a.vec = 1;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
  a.on = on; a.oc = oc; a.oh = oh; a.ow = ow;
  return launch_small_pair<4>(a, B, stream);
}

// yprime: dense [B][H][W][32] fp32 (convT_pair_f16's tail form); out: [B][Cout][2H][2W] with arbitrary strides
int convT_gather_f32(const float *yprime, const float *bias, float *out, int B, int H, int W, int Cout, int on, int oc,
                     int oh, int ow, int relu, hipStream_t stream) {
  if (Cout < 1 || Cout > 2) return unsupported("convT_gather: Cout <= 2");
  if (B > 65535) return unsupported("convT_gather: grid too large");
  const int64_t in_elems = (int64_t)B * H * W * 32;
  if (in_elems * 4 >= 0x70000000ll) return unsupported("convT_gather: tensor spans 1.75 GiB or more");
  ConvTSmallArgs a;
  a.in = yprime; a.wn = nullptr; a.bias = bias; a.out = out;
  a.in_bytes = (unsigned)(in_elems * 4);
  a.sn = H * W * 32; a.sc = 1; a.sh = W * 32; a.sw = 32; a.vec = 1;
  a.H = H; a.W = W; a.Cin = 32; a.Cout = Cout; a.relu = relu;
  a.on = on; a.oc = oc; a.oh = oh; a.ow = ow;
  constexpr int TH = 6;                          // (TH + 2) x 34 pixels x 128 B = 34 KB: four workgroups per CU
  auto kern = convT_gather_kernel<TH>;
  constexpr size_t smem = (size_t)(TH + 2) * HW_ * 128;
  if ((H + TH - 1) / TH > 65535) return unsupported("convT_gather: grid too large");
  const double M = (double)B * H * W;
  prof::Scope scope(prof::K_CONVT_SMALL, 2.0 * M * 16 * Cout, 4.0 * (M * 32 + 4.0 * M * Cout), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3((W + TW - 1) / TW, (H + TH - 1) / TH, B), dim3(256), smem, stream, a);
  return check_launch("convT_gather_f32");
}

// src / dst strides arbitrary (32-bit range checked by the caller); in_elems = extent of the source.
int convT_k4s2_small_f32(const float *in, const float *wn, const float *bias, float *out, int B, int H,
                         int W, int Cin, int Cout, int64_t in_elems, int sn, int sc, int sh, int sw, int on,
                         int oc, int oh, int ow, int relu, hipStream_t stream) {
  if (!convT_small_applicable(Cin, Cout))
    return unsupported("convT_small: need Cout <= 4 and Cin in {32, 64}");
  if (B > 65535) return unsupported("convT_small: grid too large");
  ConvTSmallArgs a;
  a.in = in; a.wn = wn; a.bias = bias; a.out = out;
  a.in_bytes = (unsigned)(in_elems * 4);
  a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw;
  a.vec = sc == 1 && sn % 4 == 0 && sh % 4 == 0 && sw % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
  a.on = on; a.oc = oc; a.oh = oh; a.ow = ow;
  // 8-row tiles recompute less halo (340 rows of Y' per 256 pixels vs 204 per 128); short maps keep 4
  const int th_env = knobs().convt_th;
  const int th = th_env ? th_env : (H >= 8 ? 8 : 4);
  return th == 8 ? dispatch_small<8>(a, B, stream) : dispatch_small<4>(a, B, stream);
}

}  // namespace isi
