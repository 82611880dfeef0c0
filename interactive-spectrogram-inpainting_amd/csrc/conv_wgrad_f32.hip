// Weight gradient of the (transposed) convolutions, gfx950 exact-fp32 matrix pipe.
//
//   dW[co][k] = sum over output pixels m of  dY[m][co] * A[m][k]
//   A = the im2col view of the layer input the forward implicit GEMM used
//       (k = (kh*KW + kw)*Cin + ci, zero padded; conv_igemm_f32.hip)
//
// i.e. the GEMM  dY^T (Cout x M)  x  A (M x K), reduced over pixels.  A workgroup
// owns a 128(co) x 128(k) tile of dW and a contiguous range of 32-pixel chunks
// (blockIdx.z = phase * nsplit + split); both operands are staged pixel-major in
// LDS ([32 pixels][128 + 4]) and read with conflict-free ds_read_b32 as MFMA
// fragments.  Splits write partial tiles; reduce_partials_kernel sums them in a
// fixed order (deterministic, no float atomics).
//
// Replaces autograd's conv weight-gradient kernels behind `loss.backward()`
// (reference train_vqvae.py:181) for nn.Conv2d / nn.ConvTranspose2d of
// vqvae/encoder_decoder.py:95-112,138,199-215 and vqvae/vqvae.py:149-150,175-201.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "split_f16.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct WgradKArgs {
  const float *x0, *x1, *dy;
  const int64_t *onehot_idx;   // when set, A[m][k] = (onehot_idx[m] == k): the one-hot matrix of bottleneck.py:75
  float *partial;
  float *db_partial;           // optional [splits][Cout] column sums of dY (bias gradient)
  unsigned x0_bytes, x1_bytes, dy_bytes;
  int C0, Cin, vec, dvec;      // vec: sources are channels-last with 16-B aligned quads; dvec: Cout % 4 == 0
  int s0n, s0c, s0h, s0w, s1n, s1h, s1w;
  int dn, dh, dw;              // dY element strides in GEMM-grid pixels (channel stride 1)
  int H, W, OH, OW, Cout, K, Kpad, KW, stride, pad, M;
  int convT, dst_sh, dst_sw;   // transposed conv: phase offset into dY
  int nsplit, chunks_per_split;
  int nz, zs_x0, zs_dy;        // nz > 1: blockIdx.z also enumerates nz independent (x, dY) pairs (element strides)
  // WgradBand (conv_wgrad_split_kernel only): output channel c of dY is non-zero only on the pixels of units
  // [lo_slope c + lo_base, hi_slope c + hi_base], a unit = win_rpu consecutive pixels; win_rpu = 0: everywhere
  int win_rpu, wlo_slope, wlo_base, whi_slope, whi_base;
  // conv_wgrad_halo_kernel only: bit 0 / 1 = source 0 / 1 holds split-f16 PAIRS (the training forward's pair tensors,
  // ISI_CONV_IN0_PAIR / IN1_PAIR): the staging decodes (hi + lo) / 4 -- exactly the value the forward's products saw
  int x_pair;
};

namespace {
constexpr int LDT = 132;  // LDS row (floats): 128 + 4
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned uniform_off = 0) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, uniform_off, 0);
  return *reinterpret_cast<float4 *>(&v);
}
}  // namespace

__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const WgradKArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ds = smem;                  // [2][32][LDT]  dY chunk  (pixel-major, co inner)
  float *Xs = smem + 2 * 32 * LDT;   // [2][32][LDT]  im2col chunk (pixel-major, k inner)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int co0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int per_z = (p.convT ? 4 : 1) * p.nsplit;
  const int zb = (int)blockIdx.z / per_z, zrem = (int)blockIdx.z - zb * per_z;
  const int phase = zrem / p.nsplit, split = zrem - phase * p.nsplit;
  const int py = p.convT ? phase >> 1 : 0, px = p.convT ? phase & 1 : 0;
  const int pad_y = p.convT ? 1 - py : p.pad, pad_x = p.convT ? 1 - px : p.pad;
  const int dy_off = p.convT ? py * p.dst_sh + px * p.dst_sw : 0;
  const float *x0p = p.x0 + (size_t)zb * p.zs_x0, *x1p = p.x1 + (size_t)zb * p.zs_x0, *dyp = p.dy + (size_t)zb * p.zs_dy;

  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x0p), 0, p.x0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x1p), 0, p.x1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dyp), 0, p.dy_bytes, 0x00020000);

  // staging role: pixel row tid/32 + 8j (j < 4), quad tid%32 of the 128-wide tile
  const int srow = tid >> 5, sq = tid & 31;
  // this thread's K position is fixed for the whole kernel
  const int kk = k0 + sq * 4;
  const bool kvalid = kk < p.K;
  int tap = 0, c = 0, kh = 0, kw = 0;
  if (kvalid) { tap = kk / p.Cin; c = kk - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW; }
  const bool second = c >= p.C0;
  const int cc = second ? c - p.C0 : c;
  const int co = co0 + sq * 4;
  const bool covalid = co < p.Cout;  // Cout % 4 == 0 is required by the launcher when vec

  const int chunk_begin = split * p.chunks_per_split;
  const int nchunks_total = (p.M + 31) / 32;
  const int chunk_end = min(nchunks_total, chunk_begin + p.chunks_per_split);

  float4 rdq[4], rxq[4];
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = ch * 32 + srow + 8 * j;
      unsigned doff = OOB, xoff = OOB;
      float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M) {
        const int b = m / (p.OH * p.OW);
        const int rem = m - b * (p.OH * p.OW);
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        if (covalid) doff = (unsigned)(dy_off + b * p.dn + oy * p.dh + ox * p.dw + co) * 4u;
        if (p.onehot_idx) {
          const int id = (int)p.onehot_idx[m];
          xs = make_float4(id == kk ? 1.f : 0.f, id == kk + 1 ? 1.f : 0.f, id == kk + 2 ? 1.f : 0.f,
                           id == kk + 3 ? 1.f : 0.f);
        } else if (p.vec) {
          const int iy = oy * p.stride - pad_y + kh, ix = ox * p.stride - pad_x + kw;
          if (kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            xoff = second ? (unsigned)(b * p.s1n + iy * p.s1h + ix * p.s1w + cc) * 4u
                          : (unsigned)(b * p.s0n + iy * p.s0h + ix * p.s0w + cc) * 4u;
        } else {
          // element-wise gather (NCHW input / Cin % 4 != 0): a quad may straddle taps
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k1 = kk + e;
            float val = 0.f;
            if (k1 < p.K) {
              const int tp = k1 / p.Cin, c1 = k1 - tp * p.Cin;
              const int kh1 = tp / p.KW, kw1 = tp - kh1 * p.KW;
              const int iy1 = oy * p.stride - pad_y + kh1, ix1 = ox * p.stride - pad_x + kw1;
              if ((unsigned)iy1 < (unsigned)p.H && (unsigned)ix1 < (unsigned)p.W)
                val = c1 < p.C0 ? x0p[b * p.s0n + c1 * p.s0c + iy1 * p.s0h + ix1 * p.s0w]
                                : x1p[b * p.s1n + (c1 - p.C0) + iy1 * p.s1h + ix1 * p.s1w];
            }
            t[e] = val;
          }
          xs = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
      if (p.dvec) {
        rdq[j] = buf_load4(rd, doff);
      } else {
        // Cout % 4 != 0 (e.g. the 2-channel spectrogram gradient): element-wise
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (doff != OOB) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) t[e] = dyp[(doff >> 2) + e];
        }
        rdq[j] = make_float4(t[0], t[1], t[2], t[3]);
      }
      rxq[j] = (p.vec && !p.onehot_idx) ? (second ? buf_load4(r1, xoff) : buf_load4(r0, xoff)) : xs;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<float4 *>(Ds + buf * 32 * LDT + (srow + 8 * j) * LDT + sq * 4) = rdq[j];
      *reinterpret_cast<float4 *>(Xs + buf * 32 * LDT + (srow + 8 * j) * LDT + sq * 4) = rxq[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fl = lane & 31, half = lane >> 5;
  float bsum = 0.f;  // column sum of dY for co0 + tid (threads < 128 of the k-tile-0 workgroups)
  const bool do_bias = p.db_partial != nullptr && blockIdx.y == 0 && tid < 128;
  if (chunk_begin < chunk_end) {
    load_chunk(chunk_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = chunk_begin; ch < chunk_end; ++ch) {
    const int buf = (ch - chunk_begin) & 1;
    if (ch + 1 < chunk_end) load_chunk(ch + 1);
    const float *d = Ds + buf * 32 * LDT + half * LDT + wm0 + fl;
    const float *x = Xs + buf * 32 * LDT + half * LDT + wn0 + fl;
#pragma unroll
    for (int t = 0; t < 16; ++t) {  // pixel pair (2t, 2t+1): k-slot = half
      const float a0 = d[2 * t * LDT], a1 = d[2 * t * LDT + 32];
      const float b0 = x[2 * t * LDT], b1 = x[2 * t * LDT + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (do_bias) {
      const float *dcol = Ds + buf * 32 * LDT + tid;
#pragma unroll
      for (int r = 0; r < 32; ++r) bsum += dcol[r * LDT];
    }
    if (ch + 1 < chunk_end) store_chunk(buf ^ 1);
    __syncthreads();
  }
  if (do_bias && co0 + tid < p.Cout) p.db_partial[(size_t)blockIdx.z * p.Cout + co0 + tid] = bsum;

  // ---- partial tile: rows = co, cols = k
  float *out = p.partial + (size_t)blockIdx.z * p.Cout * p.Kpad;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kcol = k0 + wn0 + j * 32 + fl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int corow = co0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (corow < p.Cout && kcol < p.Kpad) out[(size_t)corow * p.Kpad + kcol] = acc[i][j][r];
      }
    }
}

// ---- split-bf16 products (ISI_CONV_BF16X3 / ISI_CONV_BF16X6, see conv_igemm_f32.hip): the reduction runs over
// pixels, so both operands are transposed on the way into LDS -- a thread stages 4 consecutive pixels x 4
// consecutive channels and writes, per channel, the 4 pixels as one 8-byte row segment of the bf16 planes
// [channel][32 pixels] (64-B rows, 16-B slots XOR-swizzled with (row >> 2) & 3); fragments are then the same
// ds_read_b128 reads as in the forward kernel with "K" = pixels.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
namespace {
constexpr int LDB = 32;
__device__ __forceinline__ int bf_slot(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 8; }
// pieces of 4 values (one channel, 4 consecutive pixels): hi, lo(, mid); x = hi + mid + lo exactly when NP == 3
template <int NP>
__device__ __forceinline__ void split4(const float a, const float b, const float c, const float d, uint2 *out) {
  const f32x2 p0 = {a, b}, p1 = {c, d};
  const bf16x2 h0 = __builtin_convertvector(p0, bf16x2), h1 = __builtin_convertvector(p1, bf16x2);
  const f32x2 r0 = p0 - __builtin_convertvector(h0, f32x2), r1 = p1 - __builtin_convertvector(h1, f32x2);
  out[0] = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  if constexpr (NP == 3) {
    const bf16x2 m0 = __builtin_convertvector(r0, bf16x2), m1 = __builtin_convertvector(r1, bf16x2);
    const bf16x2 l0 = __builtin_convertvector(r0 - __builtin_convertvector(m0, f32x2), bf16x2);
    const bf16x2 l1 = __builtin_convertvector(r1 - __builtin_convertvector(m1, f32x2), bf16x2);
    out[2] = make_uint2(__builtin_bit_cast(unsigned, m0), __builtin_bit_cast(unsigned, m1));
    out[1] = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
  } else {
    out[1] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf16x2)),
                        __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2)));
  }
}
}  // namespace

// Tile variants (round 3): the 128 x 128 tile wastes 3/4 of its matrix work and LDS traffic on the residual blocks'
// 32-channel sides (3x3 128->32: Cout = 32; 1x1 32->128: K = 32), which are 16 of the step's 29 launches.  WR x WC
// waves (WR * WC = 4), each owning MT x NT 32 x 32 accumulator tiles: tile = (WR MT 32) channels x (WC NT 32) k.
// ONEHOT (vq_embed_sum): the "im2col" operand is the one-hot matrix of p.onehot_idx, generated while staging; it is
// exact in its hi piece, so with NP = 3 (dY = hi + mid + lo exactly) the three products against that piece give the
// exact fp32 terms -- the codebook's segment sums on the bf16 pipe instead of the fp32 one.
template <int NP, int WR, int MT, int NT, bool ONEHOT = false>   // NP 2: three-term split, 3: six-term split
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const WgradKArgs p) {
  constexpr int WC = 4 / WR, TCO = WR * MT * 32, TK = WC * NT * 32, JX = (TK + 127) / 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Dp = reinterpret_cast<unsigned short *>(smem);   // [NP][TCO][LDB]  dY^T pieces (0 hi, 1 lo, 2 mid)
  unsigned short *Xp = Dp + NP * TCO * LDB;                        // [NP][TK ][LDB]  im2col^T pieces
  float *bias_s = reinterpret_cast<float *>(Xp + NP * TK * LDB);   // [8][TCO] bias partials of the pixel groups
  constexpr int PSD = TCO * LDB, PSX = TK * LDB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WC) * (MT * 32), wn0 = (wave % WC) * (NT * 32);
  const int co0 = blockIdx.x * TCO, k0 = blockIdx.y * TK;
  const int per_z = (p.convT ? 4 : 1) * p.nsplit;
  const int zb = (int)blockIdx.z / per_z, zrem = (int)blockIdx.z - zb * per_z;
  const int phase = zrem / p.nsplit, split = zrem - phase * p.nsplit;
  const int py = p.convT ? phase >> 1 : 0, px = p.convT ? phase & 1 : 0;
  const int pad_y = p.convT ? 1 - py : p.pad, pad_x = p.convT ? 1 - px : p.pad;
  const int dy_off = p.convT ? py * p.dst_sh + px * p.dst_sw : 0;
  const float *x0p = p.x0 + (size_t)zb * p.zs_x0, *x1p = p.x1 + (size_t)zb * p.zs_x0, *dyp = p.dy + (size_t)zb * p.zs_dy;

  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x0p), 0, p.x0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x1p), 0, p.x1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dyp), 0, p.dy_bytes, 0x00020000);

  // staging role: pixel group pg (pixels 4 pg .. 4 pg + 3 of the 32-pixel chunk), channel quad sq of dY (when inside
  // the tile) and quads sq + 32 j of the im2col columns
  const int pg = tid >> 5, sq = tid & 31;
  const bool drole = sq * 4 < TCO;
  bool kvalid[JX], second[JX], xrole[JX];
  int kh[JX], kw[JX], cc[JX];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int kk = k0 + (sq + 32 * j) * 4;
    xrole[j] = (sq + 32 * j) * 4 < TK;
    kvalid[j] = xrole[j] && kk < p.K;
    int tap = 0, c = 0;
    kh[j] = 0; kw[j] = 0;
    if (kvalid[j]) { tap = kk / p.Cin; c = kk - tap * p.Cin; kh[j] = tap / p.KW; kw[j] = tap - kh[j] * p.KW; }
    second[j] = c >= p.C0;
    cc[j] = second[j] ? c - p.C0 : c;
  }
  const int co = co0 + sq * 4;
  const bool covalid = drole && co < p.Cout;

  const int nchunks_total = (p.M + 31) / 32;
  int cw_lo = 0, cw_hi = nchunks_total, cps = p.chunks_per_split;
  if (p.win_rpu) {     // the pixels on which this tile's channels of dY can be non-zero, shared out among the splits
    const int u_lo = max(0, p.wlo_slope * co0 + p.wlo_base);
    const int u_hi = min((p.M - 1) / p.win_rpu, p.whi_slope * (co0 + TCO - 1) + p.whi_base);
    cw_lo = min(nchunks_total, (u_lo * p.win_rpu) / 32);
    cw_hi = u_hi < u_lo ? cw_lo : min(nchunks_total, ((u_hi + 1) * p.win_rpu + 31) / 32);
    cps = (cw_hi - cw_lo + p.nsplit - 1) / p.nsplit;
  }
  const int chunk_begin = min(cw_hi, cw_lo + split * cps);
  const int chunk_end = min(cw_hi, chunk_begin + cps);

  float4 rdq[4], rxq[JX][4];
  // (batch, row, column) of this thread's four pixels, carried from chunk to chunk (chunks are visited in
  // order: + 32 pixels each) instead of two integer divisions per pixel and chunk
  int pb[4], py_[4], px4[4];
  const int adv_b = 32 / (p.OH * p.OW), adv_r = 32 - adv_b * (p.OH * p.OW);
  const int adv_y = adv_r / p.OW, adv_x = adv_r - adv_y * p.OW;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = chunk_begin * 32 + 4 * pg + j;
    pb[j] = m / (p.OH * p.OW);
    const int rem = m - pb[j] * (p.OH * p.OW);
    py_[j] = rem / p.OW;
    px4[j] = rem - py_[j] * p.OW;
  }
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = ch * 32 + 4 * pg + j;
      const bool mv = m < p.M;
      const int b = pb[j], oy = py_[j], ox = px4[j];
      unsigned doff = OOB;
      if (mv && covalid) doff = (unsigned)(dy_off + b * p.dn + oy * p.dh + ox * p.dw + co) * 4u;
      rdq[j] = buf_load4(rd, doff);
      if constexpr (ONEHOT) {
        const int code = mv ? (int)p.onehot_idx[m] : -1;
#pragma unroll
        for (int q = 0; q < JX; ++q) {
          const int kk = k0 + (sq + 32 * q) * 4;
          rxq[q][j] = make_float4(code == kk ? 1.f : 0.f, code == kk + 1 ? 1.f : 0.f, code == kk + 2 ? 1.f : 0.f,
                                  code == kk + 3 ? 1.f : 0.f);
        }
      } else
#pragma unroll
      for (int q = 0; q < JX; ++q) {
        unsigned xoff = OOB;
        const int iy = oy * p.stride - pad_y + kh[q], ix = ox * p.stride - pad_x + kw[q];
        if (mv && kvalid[q] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          xoff = second[q] ? (unsigned)(b * p.s1n + iy * p.s1h + ix * p.s1w + cc[q]) * 4u
                           : (unsigned)(b * p.s0n + iy * p.s0h + ix * p.s0w + cc[q]) * 4u;
        rxq[q][j] = second[q] ? buf_load4(r1, xoff) : buf_load4(r0, xoff);
      }
      // next chunk: + 32 pixels = adv_b images + adv_y rows + adv_x columns, one carry each (branch-free)
      px4[j] += adv_x;
      const int c1 = px4[j] >= p.OW ? 1 : 0;
      px4[j] -= c1 ? p.OW : 0;
      py_[j] += adv_y + c1;
      const int c2 = py_[j] >= p.OH ? 1 : 0;
      py_[j] -= c2 ? p.OH : 0;
      pb[j] += adv_b + c2;
    }
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};   // this thread's 4 channels of dY over its pixel group, all chunks
  auto pick = [](const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
  auto store_chunk = [&]() {
    bsum[0] += (rdq[0].x + rdq[1].x) + (rdq[2].x + rdq[3].x);
    bsum[1] += (rdq[0].y + rdq[1].y) + (rdq[2].y + rdq[3].y);
    bsum[2] += (rdq[0].z + rdq[1].z) + (rdq[2].z + rdq[3].z);
    bsum[3] += (rdq[0].w + rdq[1].w) + (rdq[2].w + rdq[3].w);
    if (drole) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = sq * 4 + e;
        const int wo = row * LDB + bf_slot(row, pg >> 1) + (pg & 1) * 4;
        uint2 pd[3];
        split4<NP>(pick(rdq[0], e), pick(rdq[1], e), pick(rdq[2], e), pick(rdq[3], e), pd);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(Dp + q * PSD + wo) = pd[q];
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      if (!xrole[j]) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (sq + 32 * j) * 4 + e;
        const int wo = row * LDB + bf_slot(row, pg >> 1) + (pg & 1) * 4;
        uint2 px_[3];
        split4<NP>(pick(rxq[j][0], e), pick(rxq[j][1], e), pick(rxq[j][2], e), pick(rxq[j][3], e), px_);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(Xp + q * PSX + wo) = px_[q];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fl = lane & 31, half = lane >> 5;
  if (chunk_begin < chunk_end) {
    load_chunk(chunk_begin);
    store_chunk();
  }
  __syncthreads();
  for (int ch = chunk_begin; ch < chunk_end; ++ch) {
    if (ch + 1 < chunk_end) load_chunk(ch + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 av[MT][NP], bv[NT][NP];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int ra = wm0 + i * 32 + fl;
        const int ao = ra * LDB + bf_slot(ra, s * 2 + half);
#pragma unroll
        for (int q = 0; q < NP; ++q) av[i][q] = *reinterpret_cast<const s16x8 *>(Dp + q * PSD + ao);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int rb = wn0 + i * 32 + fl;
        const int bo = rb * LDB + bf_slot(rb, s * 2 + half);
#pragma unroll
        for (int q = 0; q < NP; ++q) bv[i][q] = *reinterpret_cast<const s16x8 *>(Xp + q * PSX + bo);
      }
      auto term = [&](int qa, int qb) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i][qa]),
                                                                __builtin_bit_cast(bf16x8, bv[j][qb]), acc[i][j], 0, 0, 0);
      };
      if constexpr (ONEHOT) {    // the one-hot operand has no mid / lo piece
        static_assert(!ONEHOT || NP == 3, "exact segment sums need the three-piece split");
        term(1, 0); term(2, 0); term(0, 0);
      } else if constexpr (NP == 3) {   // smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
        term(1, 0); term(0, 1); term(2, 2); term(2, 0); term(0, 2); term(0, 0);
      } else {
        term(1, 0); term(0, 1); term(0, 0);
      }
    }
    __syncthreads();   // every wave has read the single stage
    if (ch + 1 < chunk_end) store_chunk();
    __syncthreads();
  }
  // ---- bias partials: fixed-order sum over the 8 pixel groups
  if (p.db_partial != nullptr && blockIdx.y == 0) {
    if (drole) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bias_s[pg * TCO + sq * 4 + e] = bsum[e];
    }
    __syncthreads();
    if (tid < TCO && co0 + tid < p.Cout) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += bias_s[g * TCO + tid];
      p.db_partial[(size_t)blockIdx.z * p.Cout + co0 + tid] = t;
    }
  }

  float *out = p.partial + (size_t)blockIdx.z * p.Cout * p.Kpad;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int kcol = k0 + wn0 + j * 32 + fl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int corow = co0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (corow < p.Cout && kcol < p.Kpad) out[(size_t)corow * p.Kpad + kcol] = acc[i][j][r];
      }
    }
}

// ---- halo-staged weight gradient (round 3).  The kernels above re-read the layer input once per TAP (a workgroup
// owns 128 im2col columns = one tap of a 128-channel layer) and dY once per 128-column tile: 2.4 GB of L2 traffic
// for a 3x3 128->128 layer whose tensors are 134 MB each -- tools/bench_wgrad.py: 190 TFLOP/s whatever the tile.
// Here a workgroup stages the input HALO of a pixel tile (R x 32 output pixels) once, as bf16 hi / lo planes in the
// tensor's own [pixel][channel] order (no transpose on the way in), and every tap reads a shifted window of it: the
// reduction runs over pixels, so the MFMA fragments are K-major in PIXELS, which `ds_read_b64_tr_b16` (gfx950's
// transposing LDS read: within 16 lanes, lane t receives element t & 3 of the segments 4 j + (t >> 2), j < 4 --
// tools/probes/tr_read_probe.hip) delivers from the pixel-major image: lane t of a group points at pixel row t / 4,
// channel quad t % 4 and gets 4 pixels of ONE channel.  A wave owns 32 output channels x (T taps x 32 input
// channels) = T accumulator tiles (T = 9 for 3x3, 8 = two kernel rows of k4s2; the four waves of a workgroup split
// over NCO channel groups x 4 / NCO input-channel slices), so the residual 3x3 (C -> 32) needs ONE pass over its
// input and the 128-channel layers re-read dY Cin / 32 times instead of 9 x the input.
//   LDS: X [slice][plane][halo pixel][32 ch], dY [group][plane][pixel][32 ch]: rows of 64 B, so the four pixel rows
//   a 32-lane read touches sit on banks 0 / 16 / 32 / 48; stride 2 keeps even and odd halo columns apart so that
//   consecutive output pixels stay on consecutive rows.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;

template <int NCO, int S, int KHG, int KWT, int R, int NG>
__global__ __launch_bounds__(256 * NG) void conv_wgrad_halo_kernel(const WgradKArgs p) {
  constexpr int NCI = 4 / NCO, T = KHG * KWT, OWT = 32;
  constexpr int HR = (R - 1) * S + KHG, HC = (OWT - 1) * S + KWT, HP = HR * HC, NPX = R * OWT;
  constexpr int HCH = (HC + 1) / 2;                       // stride 2: even columns first, then the odd ones
  constexpr int XI = NCI * HP * 8, DI = NCO * NPX * 8;    // float4 items per tile
  constexpr int NTH = 256 * NG;                           // NG = 2: 8 waves, two groups of four, see below
  constexpr int NXI = (XI + NTH - 1) / NTH, NDI = DI / NTH;
  constexpr int XPL = HP * 64, DPL = NPX * 64;            // bytes of one bf16 plane
  static_assert(DI % NTH == 0 && NPX == 64 && NDI * NG == 2 * NCO, "dY staging / bias reduction assume 64-pixel tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char *Xs = smem_b, *Ds = smem_b + NCI * 2 * XPL;

  // NG = 2: two groups of four waves.  Both groups own the same four (channel group, input-channel slice) roles; group 0
  // takes the first (T + 1) / 2 taps, group 1 the rest: half the accumulator registers per wave (two waves per SIMD fit:
  // their LDS reads and matrix instructions overlap -- one wave per SIMD measured 0.41 of the three-term ceiling), no
  // merge at the end, and 512 threads share the staging.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = NG == 2 ? wave >> 2 : 0, w4 = wave & 3;
  const int cg = w4 % NCO, cs = w4 / NCO;
  const int KH = p.K / (p.KW * p.Cin), NTG = KH / KHG, NCB = p.Cin / (32 * NCI);
  const int unit = blockIdx.x, split = blockIdx.y;
  const int tg = unit % NTG, cb = (unit / NTG) % NCB, ob = unit / (NTG * NCB);
  const int co0 = ob * (32 * NCO), ci0 = cb * (32 * NCI);

  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x0), 0, p.x0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x1), 0, p.x1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dy), 0, p.dy_bytes, 0x00020000);

  // staging items of this thread (tile-invariant part): element offset relative to the tile's (b, oy0 S, ox0 S)
  // origin, halo coordinates for the border test, LDS destination
  int xrel[NXI], xdst[NXI];
  short xhy[NXI], xhx[NXI];
  bool xsec[NXI];
#pragma unroll
  for (int j = 0; j < NXI; ++j) {
    const int item = tid + NTH * j;
    const int q = item & 7, hp = (item >> 3) % HP, sl = (item >> 3) / HP;
    const int hy = hp / HC, hx = hp % HC;
    const int c = ci0 + sl * 32 + q * 4;
    xsec[j] = c >= p.C0;
    const int dyy = hy + tg * KHG - p.pad, dxx = hx - p.pad;
    xhy[j] = (short)dyy; xhx[j] = (short)dxx;
    xrel[j] = item < XI ? (xsec[j] ? dyy * p.s1h + dxx * p.s1w + (c - p.C0) : dyy * p.s0h + dxx * p.s0w + c) : INT32_MIN;
    const int pos = S == 1 ? hx : (hx & 1) * HCH + (hx >> 1);
    xdst[j] = sl * 2 * XPL + (hy * HC + pos) * 64 + q * 8;
  }
  int drel[NDI], ddst[NDI];
#pragma unroll
  for (int j = 0; j < NDI; ++j) {
    const int item = tid + NTH * j;
    const int q = item & 7, px = (item >> 3) % NPX, gr = (item >> 3) / NPX;
    drel[j] = (px / OWT) * p.dh + (px % OWT) * p.dw + co0 + gr * 32 + q * 4;
    ddst[j] = gr * 2 * DPL + px * 64 + q * 8;
  }

  const int TPR = p.OW / OWT, TPI = (p.OH / R) * TPR;
  const int tile_begin = split * p.chunks_per_split;
  const int tile_end = min(p.M, tile_begin + p.chunks_per_split);   // M = number of pixel tiles here

  float4 rx[NXI], rdy[NDI];
  auto load_tile = [&](int tile) {
    const int b = tile / TPI, rem = tile - b * TPI;
    const int oy0 = (rem / TPR) * R, ox0 = (rem - (rem / TPR) * TPR) * OWT;
    const int iy0 = oy0 * S, ix0 = ox0 * S;
    const int xb0 = b * p.s0n + iy0 * p.s0h + ix0 * p.s0w, xb1 = b * p.s1n + iy0 * p.s1h + ix0 * p.s1w;
    const int db0 = b * p.dn + oy0 * p.dh + ox0 * p.dw;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const int iy = iy0 + xhy[j], ix = ix0 + xhx[j];
      const bool ok = xrel[j] != INT32_MIN && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)((xsec[j] ? xb1 : xb0) + xrel[j]) * 4u : OOB;
      rx[j] = xsec[j] ? buf_load4(r1, off) : buf_load4(r0, off);
    }
#pragma unroll
    for (int j = 0; j < NDI; ++j) rdy[j] = buf_load4(rd, (unsigned)(db0 + drel[j]) * 4u);
  };
  float bsum[NCO][4];
#pragma unroll
  for (int g = 0; g < NCO; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[g][e] = 0.f;
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      if (NTH * j + NTH - 1 >= XI && tid + NTH * j >= XI) continue;
      if (p.x_pair & (xsec[j] ? 2 : 1)) {
        // pair8 storage: the 16 bytes this thread loaded are the 8 hi pieces (even quad) or the 8 lo pieces (odd quad) of
        // its 8-channel group; the neighbouring lane (same item index +- 1: NTH is even) loaded the other half.  Each
        // hands the other the two dwords it needs (DPP quad_perm [1,0,3,2]) and decodes its own four channels.
        const bool odd = tid & 1;
        const unsigned a0 = __builtin_bit_cast(unsigned, rx[j].x), a1 = __builtin_bit_cast(unsigned, rx[j].y);
        const unsigned a2 = __builtin_bit_cast(unsigned, rx[j].z), a3 = __builtin_bit_cast(unsigned, rx[j].w);
        const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? a0 : a2), 0xB1, 0xF, 0xF, false);
        const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? a1 : a3), 0xB1, 0xF, 0xF, false);
        const unsigned h0 = odd ? r0 : a0, h1 = odd ? r1 : a1, l0 = odd ? a2 : r0, l1 = odd ? a3 : r1;
        constexpr float q4 = 1.f / f16s::kScaleA;
        rx[j].x = f16s::mix_sum<0>(h0, l0) * q4; rx[j].y = f16s::mix_sum<1>(h0, l0) * q4;
        rx[j].z = f16s::mix_sum<0>(h1, l1) * q4; rx[j].w = f16s::mix_sum<1>(h1, l1) * q4;
      }
      uint2 pc[3];
      split4<2>(rx[j].x, rx[j].y, rx[j].z, rx[j].w, pc);
      *reinterpret_cast<uint2 *>(Xs + xdst[j]) = pc[0];
      *reinterpret_cast<uint2 *>(Xs + xdst[j] + XPL) = pc[1];
    }
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
      uint2 pc[3];
      split4<2>(rdy[j].x, rdy[j].y, rdy[j].z, rdy[j].w, pc);
      *reinterpret_cast<uint2 *>(Ds + ddst[j]) = pc[0];
      *reinterpret_cast<uint2 *>(Ds + ddst[j] + DPL) = pc[1];
      constexpr int JG = 2 / NG;   // dY items per channel group and thread
      bsum[j / JG][0] += rdy[j].x; bsum[j / JG][1] += rdy[j].y; bsum[j / JG][2] += rdy[j].z; bsum[j / JG][3] += rdy[j].w;
    }
  };

  // fragment addressing: 16-lane group g = lane >> 4 covers channels 16 (g & 1) .. + 15 and the pixels
  // 8 (g >> 1) + {0..3 | 4..7} of a 16-pixel step; lane t of the group points at pixel row t / 4, channel quad t % 4
  const int fg = lane >> 4, ft = lane & 15;
  const int frow = 8 * (fg >> 1) + (ft >> 2), fch = (16 * (fg & 1) + 4 * (ft & 3)) * 2;
  const unsigned char *Ab = Ds + cg * 2 * DPL + frow * 64 + fch;
  const unsigned char *Bb = Xs + cs * 2 * XPL + frow * 64 + fch;
  auto frag = [](const unsigned char *q) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q + 4 * 64));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  constexpr int TG = NG == 2 ? (T + 1) / 2 : T;   // taps per wave group
  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // taps [T0, T0 + NTP) of this group; a product of a tap waits for its own accumulator, so the taps of a kernel row
  // are issued round-robin (NG = 1) / the SIMD's other wave fills the gaps (NG = 2)
  auto compute_taps = [&](auto t0_, auto ntp_) {
    constexpr int T0 = decltype(t0_)::value, NTP = decltype(ntp_)::value;
#pragma unroll
    for (int ks = 0; ks < 2 * R; ++ks) {
      const int r = ks >> 1, kc = ks & 1;
      const s16x8 ah = frag(Ab + (r * OWT + kc * 16) * 64), al = frag(Ab + DPL + (r * OWT + kc * 16) * 64);
      constexpr int CH = NG == 2 ? 1 : KWT;          // taps whose fragments are held together
#pragma unroll
      for (int c0 = 0; c0 < NTP; c0 += CH) {
        s16x8 bh[CH], bl[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (c0 + c >= NTP) continue;
          const int t = T0 + c0 + c, khl = t / KWT, kwl = t % KWT;
          const int col = S == 1 ? kc * 16 + kwl : (kwl & 1) * HCH + kc * 16 + (kwl >> 1);
          const int off = ((r * S + khl) * HC + col) * 64;
          bh[c] = frag(Bb + off);
          bl[c] = frag(Bb + XPL + off);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c0 + c < NTP) acc[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[c]), acc[c0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c0 + c < NTP) acc[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[c]), acc[c0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c0 + c < NTP) acc[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[c]), acc[c0 + c], 0, 0, 0);
      }
    }
  };
  auto compute_tile = [&]() {
    if (NG == 2 && grp == 1) compute_taps(std::integral_constant<int, TG>{}, std::integral_constant<int, T - TG>{});
    else compute_taps(std::integral_constant<int, 0>{}, std::integral_constant<int, TG>{});
  };

  if (tile_begin < tile_end) {
    load_tile(tile_begin);
    store_tile();
  }
  __syncthreads();
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    if (tile + 1 < tile_end) load_tile(tile + 1);
    compute_tile();
    __syncthreads();   // every wave has read the single stage
    if (tile + 1 < tile_end) store_tile();
    __syncthreads();
  }

  // ---- bias partials (units of the first input-channel block and tap group): fixed-order sum over the 64 staging
  // threads (pixels of a tile) of a channel quad
  if (p.db_partial != nullptr && cb == 0 && tg == 0) {
    constexpr int NPT = NTH / 8;                       // staging threads per channel quad
    float *red = reinterpret_cast<float *>(smem_b);   // [NCO * 32 channels][NPT + 1]
#pragma unroll
    for (int g = 0; g < NCO; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(g * 32 + (tid & 7) * 4 + e) * (NPT + 1) + (tid >> 3)] = bsum[g][e];
    __syncthreads();
    if (tid < NCO * 32) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < NPT; ++g) t += red[tid * (NPT + 1) + g];
      p.db_partial[(size_t)split * p.Cout + co0 + tid] = t;
    }
    __syncthreads();
  }

  float *out = p.partial + (size_t)split * p.Cout * p.Kpad;
  const int fl = lane & 31, half = lane >> 5;
#pragma unroll
  for (int tl = 0; tl < TG; ++tl) {
    const int t = grp * TG + tl;
    if (t >= T) break;
    const int tap = (tg * KHG + t / KWT) * KWT + t % KWT;
    const int kcol = tap * p.Cin + ci0 + cs * 32 + fl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int corow = co0 + cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)corow * p.Kpad + kcol] = acc[tl][r];
    }
  }
}

// ---- weight gradient of a LINEAR layer (round 3): dW[n][k] = sum_m dY[m][n] X[m][k], M = tokens.  The per-tap kernel
// above transposes both operands on the way into a single LDS stage (8-byte column pieces per channel) between two
// barriers per 32-row chunk.  Here both row chunks go to LDS in their own [row][channel] order (8-byte pieces of 4
// channels, slices of 32 channels = 64-byte rows like conv_wgrad_halo_kernel) and `ds_read_b64_tr_b16` delivers the
// row-major fragments the reduction over rows needs; two-stage ring, one barrier per chunk, 8 waves on a 128 x 128 tile
// (wave: 32 output features x 64 input features), 64 KB of LDS: two workgroups per CU.
__global__ __launch_bounds__(512, 2) void linear_wgrad_kernel(const WgradKArgs p) {
  // bytes of one 32-channel slice: hi plane, lo plane of [32 rows][64 B], + 128: a staging instruction's 64 lanes write two
  // rows of all four slices -- with slices a multiple of 256 bytes apart those were four-way bank conflicts (SQ counters:
  // 29 % of the LDS cycles); 128 bytes apart they tile the 64 banks exactly twice
  constexpr int SL = 2 * 32 * 64 + 128;
  constexpr int OPB = 4 * SL;                // one operand's chunk: four slices (128 channels)
  constexpr int STAGE = 2 * OPB;             // dY, then X
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 3, wk = wave >> 2;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128, split = blockIdx.z;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x0), 0, p.x0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dy), 0, p.dy_bytes, 0x00020000);

  // staging: float4 number tid + 512 j of a 32 x 128 chunk: row = that / 32, channel quad = tid % 32.  A chunk travels as
  // four pieces per thread (dY rows sm, sm + 16; X rows sm, sm + 16): piece g of chunk c + 1 is converted and written to
  // the other LDS stage behind the g-th group of matrix instructions of chunk c, and piece g of chunk c + 2 is requested
  // into the registers this frees (gemm_split_f32.hip has the measurements behind this order: no branch inside a chunk, so
  // that the compiler's wait counts stay exact; a request has a whole chunk to come back).
  const int sq = tid & 31, sm = tid >> 5;
  const int sdst = (sq >> 3) * SL + (sq & 7) * 8;
  const int chunk_begin = split * p.chunks_per_split;
  const int chunk_end = min((p.M + 31) / 32, chunk_begin + p.chunks_per_split);
  const unsigned vd0 = (unsigned)(sm * p.dw + n0 + sq * 4) * 4u, vd1 = vd0 + (unsigned)(16 * p.dw) * 4u;
  const unsigned vx0 = (unsigned)(sm * p.s0w + k0 + sq * 4) * 4u, vx1 = vx0 + (unsigned)(16 * p.s0w) * 4u;
  float4 pc4[4];
  auto load_piece = [&](int ch, int g) {
    // (the chunk's row offset travels in the scalar offset, which the range check ignores: rows beyond M -- the last
    // chunk's -- get the out-of-range vector offset by hand)
    const int rows = p.M - ch * 32;
    const bool ok = sm + 16 * (g >> 1) < rows;
    if (g & 1) pc4[g] = buf_load4(rx, ok ? (g >> 1 ? vx1 : vx0) : OOB, (unsigned)(ch * 32 * p.s0w) * 4u);
    else pc4[g] = buf_load4(rd, ok ? (g >> 1 ? vd1 : vd0) : OOB, (unsigned)(ch * 32 * p.dw) * 4u);
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto store_piece = [&](int stage, int g) {
    unsigned char *st = smem_b + stage * STAGE + (g & 1) * OPB;
    const int o = sdst + (sm + 16 * (g >> 1)) * 64;
    uint2 pc[3];
    split4<2>(pc4[g].x, pc4[g].y, pc4[g].z, pc4[g].w, pc);
    *reinterpret_cast<uint2 *>(st + o) = pc[0];
    *reinterpret_cast<uint2 *>(st + o + 2048) = pc[1];
    if (!(g & 1)) { bsum[0] += pc4[g].x; bsum[1] += pc4[g].y; bsum[2] += pc4[g].z; bsum[3] += pc4[g].w; }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // fragment addressing (conv_wgrad_halo_kernel): 16-lane group g covers channels 16 (g & 1) .. + 15 and rows
  // 8 (g >> 1) + {0..3 | 4..7} of a 16-row step; lane t points at row t / 4, channel quad t % 4
  const int fg = lane >> 4, ft = lane & 15;
  const int foff = (8 * (fg >> 1) + (ft >> 2)) * 64 + (16 * (fg & 1) + 4 * (ft & 3)) * 2;
  auto frag = [](const unsigned char *q) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q + 4 * 64));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  if (chunk_begin < chunk_end) {
#pragma unroll
    for (int g = 0; g < 4; ++g) load_piece(chunk_begin, g);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      store_piece(0, g);
      if (chunk_begin + 1 < chunk_end) load_piece(chunk_begin + 1, g);
    }
  }
  __syncthreads();
  // a chunk = 4 units (16-row step s, input-feature tile j) of 3 matrix instructions; the fragments of unit u + 1 are read
  // before unit u's matrix instructions are issued
  auto chunk = [&](const int ch, auto last) {
    constexpr bool LAST = decltype(last)::value;
    const int rel = ch - chunk_begin;
    const unsigned char *st = smem_b + (rel & 1) * STAGE;
    const unsigned char *ab = st + wn * SL + foff, *bb = st + OPB + (2 * wk) * SL + foff;
    const int ch_next = min(ch + 2, chunk_end - 1);
    s16x8 ah[2], al[2], bh[2], bl[2];
    ah[0] = frag(ab); al[0] = frag(ab + 2048);
    bh[0] = frag(bb); bl[0] = frag(bb + 2048);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s_ = u >> 1, j = u & 1;
      if (u + 1 < 4) {
        const int s1 = (u + 1) >> 1, j1 = (u + 1) & 1;
        if (j1 == 0) { ah[s1] = frag(ab + s1 * 16 * 64); al[s1] = frag(ab + 2048 + s1 * 16 * 64); }
        bh[(u + 1) & 1] = frag(bb + j1 * SL + s1 * 16 * 64);
        bl[(u + 1) & 1] = frag(bb + j1 * SL + 2048 + s1 * 16 * 64);
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[s_]), __builtin_bit_cast(bf16x8, bh[u & 1]), acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s_]), __builtin_bit_cast(bf16x8, bl[u & 1]), acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s_]), __builtin_bit_cast(bf16x8, bh[u & 1]), acc[j], 0, 0, 0);
      if (!LAST) {
        store_piece((rel + 1) & 1, u);   // last read one iteration ago: every wave is past its barrier
        load_piece(ch_next, u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  for (int ch = chunk_begin; ch + 1 < chunk_end; ++ch) chunk(ch, std::false_type{});
  if (chunk_begin < chunk_end) chunk(chunk_end - 1, std::true_type{});

  // ---- bias partials: fixed-order sum over the 16 staging threads of a channel quad
  if (p.db_partial != nullptr && blockIdx.y == 0) {
    float *red = reinterpret_cast<float *>(smem_b);   // [128 channels][16 + 1]
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(sq * 4 + e) * 17 + sm] = bsum[e];
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[tid * 17 + g];
      p.db_partial[(size_t)split * p.Cout + n0 + tid] = t;
    }
  }
  float *out = p.partial + (size_t)split * p.Cout * p.Kpad;
  const int fl = lane & 31, half = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kcol = k0 + (2 * wk + j) * 32 + fl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nrow = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)nrow * p.Kpad + kcol] = acc[j][r];
    }
  }
}

// out[i] = sum_s partial[s * stride + i]  (fixed order: 4 interleaved split groups, then a fixed tree).  A thread owns
// four consecutive elements (16-byte loads when the rows allow it) and keeps four loads in flight.
// `map` (optional): the summed [Cout][Kpad] matrix (k = tap * cin + ci) is written in torch's weight layout
// [Cout][keep][taps] instead, dropping the k padding and the channels ci >= keep (zero-padded input channels).
struct WgradOutMap { int K, Kpad, cin, taps, keep; };
// A second, small reduction rides in the same launch (the bias gradient behind a layer's weight gradient: blocks
// [nb_main, gridDim.x) -- 81 launches of ~5 us less per training step of the prior).
struct ReduceJob2 { const float *partial; float *out; int64_t n; int nsplit; int64_t stride; int nb_main; };
// one 256-thread block's share (64 quads starting at quad 64 bx) of a split reduction
template <bool VEC>
__device__ __forceinline__ void reduce_block(const float *__restrict__ partial, float *__restrict__ out, const int64_t n,
                                             const int nsplit, const int64_t stride, const int accumulate,
                                             const WgradOutMap map, const int bx, float4 (*red)[64]) {
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = ((int64_t)bx * 64 + e) * 4;
  auto ld = [&](int k) -> float4 {
    const float *q = partial + (size_t)k * stride + i;
    if constexpr (VEC) return *reinterpret_cast<const float4 *>(q);
    return make_float4(q[0], i + 1 < n ? q[1] : 0.f, i + 2 < n ? q[2] : 0.f, i + 3 < n ? q[3] : 0.f);
  };
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    int k = g;
    for (; k + 12 < nsplit; k += 16) {   // same order of additions as the one-at-a-time loop below
      const float4 a = ld(k), b = ld(k + 4), c = ld(k + 8), d = ld(k + 12);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < nsplit; k += 4) {
      const float4 a = ld(k);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    const float4 a = red[0][e], b = red[1][e], c = red[2][e], d = red[3][e];
    const float t[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                        (a.w + b.w) + (c.w + d.w)};
    if (map.Kpad) {
      const int co = (int)(i / map.Kpad), k = (int)(i - (int64_t)co * map.Kpad);   // Kpad % 4 == 0: one row per quad
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tap = (k + j) / map.cin, ci = (k + j) - tap * map.cin;
        if (k + j < map.K && ci < map.keep) out[((size_t)co * map.keep + ci) * map.taps + tap] = t[j];
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j < n) out[i + j] = accumulate ? out[i + j] + t[j] : t[j];
  }
}
template <bool VEC>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial,
                                                              float *__restrict__ out, int64_t n, int nsplit,
                                                              int64_t stride, int accumulate, int64_t zs_partial,
                                                              int64_t zs_out, WgradOutMap map, const ReduceJob2 j2) {
  __shared__ float4 red[4][64];
  int bx = blockIdx.x;
  if (j2.partial && bx >= j2.nb_main) {          // (uniform per workgroup)
    bx -= j2.nb_main;
    partial = j2.partial; out = j2.out; n = j2.n; nsplit = j2.nsplit; stride = j2.stride; accumulate = 0;
    map.Kpad = 0;
  } else {
    partial += (size_t)blockIdx.y * zs_partial;   // grid y: independent reductions
    out += (size_t)blockIdx.y * zs_out;
  }
  reduce_block<VEC>(partial, out, n, nsplit, stride, accumulate, map, bx, red);
}

// ---- deferred reductions (round 5).  A VQ-VAE training step ran 31 of these launches -- one behind every weight-gradient
// GEMM, ~9 us each for a few hundred KB -- although nothing reads a gradient before the step's end (or, data parallel, its
// bucket's all-reduce).  With a sink installed (isi_conv_wgrad_deferred_f32) the GEMM's launch returns the reduction as a
// JOB instead of launching it; isi_reduce_jobs_f32 runs up to 48 jobs per launch, the job table travelling by value in the
// kernel arguments (no device table to upload: the launch records into a HIP graph like any other).
constexpr int kJobsPerLaunch = 48;
struct ReduceJobPack { isi_reduce_job j[kJobsPerLaunch]; };
static_assert(sizeof(ReduceJobPack) <= 3800, "the job table must fit the kernel-argument segment");
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const ReduceJobPack pack) {
  __shared__ float4 red[4][64];
  const isi_reduce_job &j = pack.j[blockIdx.y];
  const WgradOutMap map{j.map_K, j.map_Kpad, j.map_cin, j.map_taps, j.map_keep};
  const int nb = (int)((j.n + 255) / 256);
  for (int bx = blockIdx.x; bx < nb; bx += gridDim.x) {     // (uniform per workgroup)
    if (j.vec) reduce_block<true>(j.partial, j.out, j.n, j.nsplit, j.stride, j.accumulate, map, bx, red);
    else reduce_block<false>(j.partial, j.out, j.n, j.nsplit, j.stride, j.accumulate, map, bx, red);
    __syncthreads();
  }
}
static thread_local std::vector<isi_reduce_job> *g_reduce_sink = nullptr;

// every row start 16-byte aligned and whole quads: the vector form
static bool reduce_vec_ok(const float *partial, int64_t n, int64_t stride, int64_t zs_partial) {
  return (n % 4 == 0) && (stride % 4 == 0) && (zs_partial % 4 == 0) && (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
}
static void launch_reduce_partials(const float *partial, float *out, int64_t n, int nsplit, int64_t stride,
                                   int accumulate, int64_t zs_partial, int64_t zs_out, int ny, hipStream_t stream,
                                   const WgradOutMap map = WgradOutMap{0, 0, 0, 0, 0},
                                   ReduceJob2 j2 = ReduceJob2{nullptr, nullptr, 0, 0, 0, 0}) {
  const bool vec = reduce_vec_ok(partial, n, stride, zs_partial);
  if (g_reduce_sink && ny == 1) {          // deferred: hand the reduction(s) to the caller's job list
    isi_reduce_job j;
    memset(&j, 0, sizeof j);
    j.partial = partial; j.out = out; j.n = n; j.stride = stride; j.nsplit = nsplit; j.accumulate = accumulate; j.vec = vec ? 1 : 0;
    j.map_K = map.K; j.map_Kpad = map.Kpad; j.map_cin = map.cin; j.map_taps = map.taps; j.map_keep = map.keep;
    g_reduce_sink->push_back(j);
    if (j2.partial) {
      memset(&j, 0, sizeof j);
      j.partial = j2.partial; j.out = j2.out; j.n = j2.n; j.stride = j2.stride; j.nsplit = j2.nsplit;
      j.vec = reduce_vec_ok(j2.partial, j2.n, j2.stride, 0) ? 1 : 0;
      g_reduce_sink->push_back(j);
    }
    return;
  }
  const unsigned nb = (unsigned)((n + 255) / 256);
  j2.nb_main = (int)nb;
  const dim3 grid(nb + (j2.partial ? (unsigned)((j2.n + 255) / 256) : 0u), ny);
  if (vec) hipLaunchKernelGGL(reduce_partials_kernel<true>, grid, dim3(256), 0, stream, partial, out, n, nsplit, stride,
                              accumulate, zs_partial, zs_out, map, j2);
  else hipLaunchKernelGGL(reduce_partials_kernel<false>, grid, dim3(256), 0, stream, partial, out, n, nsplit, stride,
                          accumulate, zs_partial, zs_out, map, j2);
}

int reduce_jobs_f32(const isi_reduce_job *jobs, int n_jobs, hipStream_t stream) {
  if (n_jobs < 0 || (n_jobs && !jobs)) return invalid("reduce_jobs: bad argument");
  for (int i0 = 0; i0 < n_jobs; i0 += kJobsPerLaunch) {
    ReduceJobPack pack;
    memset(&pack, 0, sizeof pack);
    const int m = std::min(kJobsPerLaunch, n_jobs - i0);
    int64_t nb_max = 1;
    for (int i = 0; i < m; ++i) {
      pack.j[i] = jobs[i0 + i];
      if (!pack.j[i].partial || !pack.j[i].out || pack.j[i].n <= 0 || pack.j[i].nsplit <= 0) return invalid("reduce_jobs: bad job");
      nb_max = std::max<int64_t>(nb_max, (pack.j[i].n + 255) / 256);
    }
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)std::min<int64_t>(nb_max, 64), m), dim3(256), 0, stream, pack);
    int rc = check_launch("reduce_jobs");
    if (rc) return rc;
  }
  return ISI_OK;
}

int conv_wgrad_deferred_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw, int cin_keep, float *db,
                            float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW, int stride,
                            int pad, int flags, hipStream_t stream, isi_reduce_job *jobs_out, int *n_jobs) {
  if (!jobs_out || !n_jobs) return invalid("conv_wgrad_deferred: null job list");
  std::vector<isi_reduce_job> sink;
  g_reduce_sink = &sink;
  const int rc = conv_wgrad_f32(s0, s1, dy, dw, db, workspace, workspace_floats, B, H, W, Cout, KH, KW, stride, pad, flags, stream, cin_keep);
  g_reduce_sink = nullptr;
  if (rc) return rc;
  if (sink.size() > 4) return unsupported("conv_wgrad_deferred: more reductions than a job list holds");
  *n_jobs = (int)sink.size();
  for (size_t i = 0; i < sink.size(); ++i) jobs_out[i] = sink[i];
  return ISI_OK;
}

static bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static int64_t extent4(int64_t n, int64_t sn, int64_t c, int64_t sc, int64_t h, int64_t sh, int64_t w, int64_t sw) {
  return (n - 1) * sn + (c - 1) * sc + (h - 1) * sh + (w - 1) * sw + 1;
}

// tile of the split-product kernel for a layer shape (the fp32 kernel's tile is always 128 x 128)
struct WgradTile { int tco, tk, small; };
static WgradTile wgrad_tile(int Cout, int Kpad) {
  // measured at B = 64 (tools/bench_wgrad.py): the 64 x 64 tile with its finer pixel split takes the 2-channel layers
  // from ~310 to ~245 us; 32 x 512 the residual 3x3 (C -> 32) from 340 to 309 (both stay bound by re-reading the input
  // once per tap); 128 x 32 for the residual 1x1 and 64 x 256 for the 64-channel layers were SLOWER (97 -> 153 us,
  // 116 -> 125 us: fewer, longer dY streams per CU) and are not selected
  if (Kpad <= 64 && Cout <= 64) return {64, 64, 1};   // the 2-channel spectrogram side (K = 16 x 4)
  if (Kpad <= 64) return {128, 64, 0};                // K = 64 operands (the attention backward's dE = G^T Q: 779 -> 770 us)
  if (Cout <= 32) return {32, 512, 0};                // residual 3x3 (C -> 32)
  return {128, 128, 0};
}
static int wgrad_nsplit_tile(int Cout, int Kpad, int M, int nphase, int nz, const WgradTile &t) {
  const int tiles = ((Cout + t.tco - 1) / t.tco) * ((Kpad + t.tk - 1) / t.tk) * nphase * nz;
  const int nchunks = (M + 31) / 32;
  // small tiles are memory-bound and light on LDS: several workgroups per CU hide the load latency
  // (K = 64 tiles -- the attention backward's dE = G^T Q, bound by the bytes of G: 136 tiles x 5 splits left a third of the
  // chip idle in a second round; measured at B8 H8 S1025: unmasked backward 653 -> 576 us from 768 to 1536, tools: ISI_WGRAD_SPLIT_TARGET)
  const int target = knobs().wgrad_split_target > 0 ? knobs().wgrad_split_target : (t.tk == 64 && !t.small ? 1536 : 768);
  const int nsplit = std::min(t.small ? 1024 : 256, std::max(1, (t.small ? 2048 : target) / tiles));
  return std::min(nsplit, std::max(1, nchunks / 8));
}
static int wgrad_nsplit(int Cout, int Kpad, int M, int nphase, int nz, bool split_kernel) {
  return wgrad_nsplit_tile(Cout, Kpad, M, nphase, nz, split_kernel ? wgrad_tile(Cout, Kpad) : WgradTile{128, 128, 0});
}
// pixel splits of the halo-staged kernel: its register budget (accumulators of 8-9 taps) admits one workgroup per
// CU, so one workgroup per CU over its `units` (channel block, tap group) pairs -- more splits only add partials
static int wgrad_halo_nsplit(int units, int ntiles) {
  return std::max(1, std::min(256 / std::max(1, std::min(units, 256)), ntiles / 4));
}
// splits of the row-major linear_wgrad_kernel (Cout, K multiples of 128): ONE definition for the launch and for the
// workspace sizer -- its count can exceed wgrad_nsplit's where (Cout/128)(K/128) lies in (384, 512)
static int linear_wgrad_nsplit(int Cout, int K, int M) {
  const int tiles = (Cout / 128) * (K / 128), nchunks = (M + 31) / 32;
  // one workgroup per CU for the small layers (a split costs a 64 KB partial tile written and read back: measured at M = 8200,
  // 512 x 512: 33 -> 29 us, 1024 x 512: 51 -> 45 us), two from 48 tiles on (1536 x 512: 66 vs 70 us with one)
  const int target = knobs().wgrad_split_target > 0 ? knobs().wgrad_split_target : (tiles <= 32 ? 256 : 512);
  return std::max(1, std::min(std::min(64, (target + tiles - 1) / tiles), nchunks / 8));
}
size_t conv_wgrad_batched_workspace_floats(int Cout, int K, int M, int nphase, int nz) {
  const int Kpad = (int)round_up(K, kBK);
  // which kernel runs depends on the operands' layout: size for any of them (halo kernel: a workgroup covers at most
  // 4 waves x 9 accumulator tiles of 32 x 32, which bounds its unit count from below and its splits from above)
  int nsplit = std::max(wgrad_nsplit(Cout, Kpad, M, nphase, nz, false), wgrad_nsplit(Cout, Kpad, M, nphase, nz, true));
  if (nphase == 1 && nz == 1 && Cout % 32 == 0 && K % 32 == 0)
    nsplit = std::max(nsplit, wgrad_halo_nsplit(((Cout / 32) * (K / 32) + 35) / 36, M / 16));
  if (nphase == 1 && nz == 1 && Cout % 128 == 0 && K % 128 == 0) nsplit = std::max(nsplit, linear_wgrad_nsplit(Cout, K, M));
  return (size_t)nz * ((size_t)nsplit * nphase * Cout * Kpad + (size_t)nsplit * nphase * Cout);
}
size_t conv_wgrad_workspace_floats(int Cout, int K, int M, int nphase) {
  return conv_wgrad_batched_workspace_floats(Cout, K, M, nphase, 1);
}

template <int NP, int WR, int MT, int NT, bool ONEHOT = false>
static void launch_wgrad_split(const WgradKArgs &a, int nzs, hipStream_t stream) {
  constexpr int WC = 4 / WR, TCO = WR * MT * 32, TK = WC * NT * 32;
  static_assert(TCO <= kBandTileMax && kBK == kBandChunk, "a banded launch reads at most this far beyond a row's band (kMargin)");
  constexpr size_t smem = (size_t)NP * (TCO + TK) * LDB * sizeof(unsigned short) + 8 * TCO * sizeof(float);
  static DeviceOnce attr_set;
  if (smem > 48 * 1024 && !attr_set.done()) {
    // (a failure here surfaces as the launch error the caller checks right after)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_split_kernel<NP, WR, MT, NT, ONEHOT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess)
      attr_set.mark();
  }
  dim3 grid((a.Cout + TCO - 1) / TCO, (a.Kpad + TK - 1) / TK, nzs);
  hipLaunchKernelGGL((conv_wgrad_split_kernel<NP, WR, MT, NT, ONEHOT>), grid, dim3(256), smem, stream, a);
}
template <int NP>
static void launch_wgrad_split_np(const WgradKArgs &a, const WgradTile &t, int nzs, hipStream_t stream) {
  if (t.tco == 128 && t.tk == 64) launch_wgrad_split<NP, 2, 2, 1>(a, nzs, stream);
  else if (t.tco == 64 && t.tk == 64) launch_wgrad_split<NP, 2, 1, 1>(a, nzs, stream);
  else if (t.tco == 32) launch_wgrad_split<NP, 1, 1, 4>(a, nzs, stream);
  else launch_wgrad_split<NP, 2, 2, 2>(a, nzs, stream);
}

// partial layout: [z][phase][split][Cout][Kpad] -> out [phase][Cout][Kpad]; bias partials [phase * split][Cout]
static int wgrad_reduce(const WgradKArgs &a, float *workspace, float *dw_packed, float *db, int nphase, int nsplit,
                        int nz, int64_t zs_dw, hipStream_t stream, int torch_keep = 0) {
  const int64_t per = (int64_t)a.Cout * a.Kpad;
  WgradOutMap map{0, 0, 0, 0, 0};
  if (torch_keep > 0) map = WgradOutMap{a.K, a.Kpad, a.Cin, a.K / a.Cin, torch_keep};
  // one phase, one operand set: the bias gradient's reduction rides in the weight gradient's launch (same vector form)
  const bool ride = db && nphase == 1 && nz == 1 &&
                    reduce_vec_ok(workspace, per, per, (int64_t)nsplit * per) == reduce_vec_ok(a.db_partial, a.Cout, a.Cout, 0);
  for (int ph = 0; ph < nphase; ++ph) {   // grid y = operand set
    launch_reduce_partials(workspace + (size_t)ph * nsplit * per, dw_packed + (size_t)ph * per, per, nsplit, per, 0,
                           (int64_t)nphase * nsplit * per, zs_dw, nz, stream, map,
                           ride ? ReduceJob2{a.db_partial, db, (int64_t)a.Cout, nsplit, (int64_t)a.Cout, 0} : ReduceJob2{nullptr, nullptr, 0, 0, 0, 0});
  }
  int rc = check_launch("reduce_partials");
  if (rc || !db || ride) return rc;
  // bias gradient: every (phase, split) partial covers a disjoint pixel set
  launch_reduce_partials(a.db_partial, db, (int64_t)a.Cout, nsplit * nphase, (int64_t)a.Cout, 0, 0, 0, 1, stream);
  return check_launch("reduce_partials(bias)");
}

template <int NCO, int S, int KHG, int KWT>
static int launch_wgrad_halo(const WgradKArgs &a, int units, int nsplit, hipStream_t stream) {
  constexpr int R = 2, NCI = 4 / NCO;
  constexpr int NG = 2;                  // two wave groups (kernel comment)
  constexpr int HP = ((R - 1) * S + KHG) * (31 * S + KWT);
  constexpr size_t smem = (size_t)NCI * 2 * HP * 64 + (size_t)NCO * 2 * R * 32 * 64;
  static_assert(smem >= (size_t)NCO * 32 * (32 * NG + 1) * 4, "bias reduction scratch");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_halo_kernel<NCO, S, KHG, KWT, R, NG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_wgrad_halo)");
    attr_set.mark();
  }
  hipLaunchKernelGGL((conv_wgrad_halo_kernel<NCO, S, KHG, KWT, R, NG>), dim3(units, nsplit), dim3(256 * NG), smem, stream, a);
  return check_launch("conv_wgrad_halo");
}

// dW packed like the forward weights: [nphase][Cout][Kpad].  x = layer input (two sources allowed), dy = gradient
// of the layer output, dense channels-last [B, OH(, *2), OW(, *2), Cout].
int conv_wgrad_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                   float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                   int stride, int pad, int transposed, hipStream_t stream, int torch_keep) {
  return conv_wgrad_batched_f32(s0, s1, dy, dw_packed, db, workspace, workspace_floats, B, H, W, Cout, KH, KW, stride,
                                pad, transposed, 1, 0, 0, 0, stream, torch_keep);
}

// nz independent weight-gradient GEMMs of one shape in a single launch: pair z reads x at s0->ptr + z * zs_x0
// and dY at dy + z * zs_dy and writes dw_packed + z * zs_dw (element strides).  One source, no bias when nz > 1.
int conv_wgrad_batched_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                           float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                           int stride, int pad, int transposed, int nz, int64_t zs_x0, int64_t zs_dy, int64_t zs_dw,
                           hipStream_t stream, int torch_keep, const WgradBand *band) {
  if (!s0 || !s0->ptr || !dy || !dw_packed || !workspace) return invalid("conv_wgrad: null pointer");
  if (torch_keep && ((transposed & 1) || nz != 1)) return unsupported("conv_wgrad: torch-layout output is for plain convolutions");
  if (nz < 1 || nz > 255) return invalid("conv_wgrad: bad batch count");
  if (nz > 1 && ((s1 && s1->ptr) || db)) return unsupported("conv_wgrad: batched launches take one source and no bias");
  if (nz > 1 && ((zs_x0 | zs_dy | zs_dw) & 3)) return invalid("conv_wgrad: batch strides must be multiples of 4 floats");
  if (zs_x0 < 0 || zs_dy < 0 || zs_x0 >= ((int64_t)1 << 31) || zs_dy >= ((int64_t)1 << 31))
    return unsupported("conv_wgrad: batch stride out of range");
  const int prec_flags = transposed & (ISI_CONV_BF16X3 | ISI_CONV_BF16X6);   // product mode rides in the flag word
  const int x_pair = ((transposed & ISI_CONV_IN0_PAIR) ? 1 : 0) | ((transposed & ISI_CONV_IN1_PAIR) ? 2 : 0);
  transposed &= 1;
  const bool two = s1 && s1->ptr;
  const int Cin = s0->C + (two ? s1->C : 0);
  int OH, OW, nphase = 1, K;
  if (transposed) {
    if (KH != 4 || KW != 4 || stride != 2 || pad != 1) return unsupported("conv_wgrad: transposed conv must be k4 s2 p1");
    OH = H; OW = W; nphase = 4; K = 4 * Cin;
  } else {
    OH = (H + 2 * pad - KH) / stride + 1; OW = (W + 2 * pad - KW) / stride + 1; K = KH * KW * Cin;
  }
  if (OH <= 0 || OW <= 0) return invalid("conv_wgrad: empty output");
  const int64_t M64 = (int64_t)B * OH * OW;
  if (M64 > INT32_MAX) return unsupported("conv_wgrad: too many pixels");
  WgradKArgs a;
  memset(&a, 0, sizeof a);
  const int64_t lim = (int64_t)1 << 30;
  const int64_t e0 = extent4(B, s0->sn, s0->C, s0->sc, H, s0->sh, W, s0->sw);
  const int64_t e1 = two ? extent4(B, s1->sn, s1->C, 1, H, s1->sh, W, s1->sw) : 1;
  const int64_t ed = transposed ? (int64_t)B * 2 * H * 2 * W * Cout : M64 * Cout;
  if (e0 > lim || e1 > lim || ed > lim) return unsupported("conv_wgrad: a tensor spans 4 GiB or more");
  a.x0 = s0->ptr; a.x1 = two ? s1->ptr : s0->ptr; a.dy = dy; a.partial = workspace;
  a.x0_bytes = (unsigned)(e0 * 4); a.x1_bytes = two ? (unsigned)(e1 * 4) : a.x0_bytes; a.dy_bytes = (unsigned)(ed * 4);
  a.C0 = s0->C; a.Cin = Cin;
  a.s0n = (int)s0->sn; a.s0c = (int)s0->sc; a.s0h = (int)s0->sh; a.s0w = (int)s0->sw;
  if (two) { a.s1n = (int)s1->sn; a.s1h = (int)s1->sh; a.s1w = (int)s1->sw; }
  bool vec = s0->sc == 1 && (s0->C % 4 == 0) && aligned16(s0->ptr) && (s0->sn % 4 == 0) && (s0->sh % 4 == 0) &&
             (s0->sw % 4 == 0);
  if (two) vec = vec && s1->sc == 1 && (s1->C % 4 == 0) && aligned16(s1->ptr) && (s1->sn % 4 == 0) &&
                 (s1->sh % 4 == 0) && (s1->sw % 4 == 0);
  if (two && s1->sc != 1) return unsupported("conv_wgrad: second source must be channels-last");
  a.vec = vec ? 1 : 0;
  a.dvec = ((Cout % 4) == 0 && aligned16(dy)) ? 1 : 0;
  a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.Cout = Cout; a.K = K; a.Kpad = (int)round_up(K, kBK);
  a.KW = transposed ? 2 : KW; a.stride = transposed ? 1 : stride; a.pad = pad; a.M = (int)M64;
  a.convT = transposed ? 1 : 0;
  if (transposed) {
    const int OWf = 2 * W;
    a.dn = 2 * H * OWf * Cout; a.dh = 2 * OWf * Cout; a.dw = 2 * Cout;
    a.dst_sh = OWf * Cout; a.dst_sw = Cout;
  } else {
    a.dn = OH * OW * Cout; a.dh = OW * Cout; a.dw = Cout;
  }
  const int nchunks = (a.M + 31) / 32;
  // split-bf16 products need the vectorised loaders (channels-last sources, Cout % 4 == 0)
  const bool use_split = prec_flags && a.vec && a.dvec;
  const WgradTile tile = wgrad_tile(Cout, a.Kpad);
  const int nsplit = wgrad_nsplit(Cout, a.Kpad, a.M, nphase, nz, use_split);
  a.nsplit = nsplit; a.chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  a.nz = nz; a.zs_x0 = (int)zs_x0; a.zs_dy = (int)zs_dy;
  if (band && band->win_rpu > 0 && band->lo_slope >= 0 && band->hi_slope >= 0) {   // (a hint: kernels without it read everything)
    a.win_rpu = band->win_rpu; a.wlo_slope = band->lo_slope; a.wlo_base = band->lo_base;
    a.whi_slope = band->hi_slope; a.whi_base = band->hi_base;
  }
  const size_t need = (size_t)nz * ((size_t)nsplit * nphase * Cout * a.Kpad + (size_t)nsplit * nphase * Cout);
  if (workspace_floats < need) { set_last_error("conv_wgrad: workspace too small"); return ISI_E_WORKSPACE; }
  a.db_partial = db ? workspace + (size_t)nsplit * nphase * Cout * a.Kpad : nullptr;
  const bool band_required = band && band->required;
  if (band_required && (!a.win_rpu || !use_split || transposed))
    return unsupported("conv_wgrad: the band of dY is only honoured by the split-product kernel");
  // a linear layer's weight gradient (rows of a dense matrix): the row-major kernel
  if (use_split && !band_required && !(prec_flags & ISI_CONV_BF16X6) && !transposed && nz == 1 && !two && KH == 1 && KW == 1 && stride == 1 &&
      pad == 0 && B == 1 && H == 1 && Cout % 128 == 0 && Cin % 128 == 0 && !knobs().no_gemm_kernel) {
    const int ns = linear_wgrad_nsplit(Cout, Cin, a.M);
    a.nsplit = ns; a.chunks_per_split = (nchunks + ns - 1) / ns;
    const size_t need_l = (size_t)ns * Cout * a.Kpad + (size_t)ns * Cout;
    if (workspace_floats < need_l) { set_last_error("conv_wgrad: workspace too small"); return ISI_E_WORKSPACE; }
    a.db_partial = db ? workspace + (size_t)ns * Cout * a.Kpad : nullptr;
    constexpr size_t smem_l = 2 * 2 * 4 * (2 * 32 * 64 + 128);
    static DeviceOnce attr_l;
    if (!attr_l.done()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(linear_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem_l) != hipSuccess)
        return check_launch("hipFuncSetAttribute(linear_wgrad)");
      attr_l.mark();
    }
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(Cout / 128, Cin / 128, ns), dim3(512), smem_l, stream, a);
    const int rc_l = check_launch("linear_wgrad");
    if (rc_l) return rc_l;
    return wgrad_reduce(a, workspace, dw_packed, db, 1, ns, 1, 0, stream, torch_keep);
  }
  // halo-staged kernel: 3x3 (s1 p1) and k4 (s2 p1) layers with 32-multiple channels and whole 2 x 32 pixel tiles
  const bool k3 = KH == 3 && KW == 3 && stride == 1 && pad == 1, k4 = KH == 4 && KW == 4 && stride == 2 && pad == 1;
  int nco = Cout % 128 == 0 ? 4 : Cout % 64 == 0 ? 2 : 1;
  while (nco < 4 && Cin % (32 * (4 / nco))) nco *= 2;   // fewer input-channel slices per workgroup when Cin is small
  const bool halo = use_split && !band_required && !(prec_flags & ISI_CONV_BF16X6) && !transposed && nz == 1 && (k3 || k4) &&
                    Cout % (32 * nco) == 0 && Cin % (32 * (4 / nco)) == 0 && s0->C % 32 == 0 && OW % 32 == 0 &&
                    OH % 2 == 0 && !(k4 && nco == 1) && !knobs().no_wgrad_halo;   // (k4, one channel group: does not fit its registers)
  if (x_pair && !halo)
    return unsupported("conv_wgrad: pair-format sources are read by the halo-staged kernel only (3x3 s1 p1 / k4 s2 p1 layers of "
                       "32-multiple channels, whole 2 x 32 pixel tiles, three-term products): isi_conv_wgrad_halo_route");
  if (halo) {
    a.x_pair = x_pair;
    const int units = (Cout / (32 * nco)) * (Cin / (32 * (4 / nco))) * (k3 ? 1 : 2);
    const int ntiles = B * (OH / 2) * (OW / 32);
    const int ns = wgrad_halo_nsplit(units, ntiles);
    a.nsplit = ns; a.chunks_per_split = (ntiles + ns - 1) / ns; a.M = ntiles;
    const size_t need_h = (size_t)ns * Cout * a.Kpad + (size_t)ns * Cout;
    if (workspace_floats < need_h) { set_last_error("conv_wgrad: workspace too small"); return ISI_E_WORKSPACE; }
    a.db_partial = db ? workspace + (size_t)ns * Cout * a.Kpad : nullptr;
    int rc_h = 0;
#define ISI_HALO(NCO_, S_, KHG_, KW_)                                                                                 \
  rc_h = launch_wgrad_halo<NCO_, S_, KHG_, KW_>(a, units, ns, stream)
    if (k3) { if (nco == 4) ISI_HALO(4, 1, 3, 3); else if (nco == 2) ISI_HALO(2, 1, 3, 3); else ISI_HALO(1, 1, 3, 3); }
    else { if (nco == 4) ISI_HALO(4, 2, 2, 4); else ISI_HALO(2, 2, 2, 4); }
#undef ISI_HALO
    if (rc_h) return rc_h;
    return wgrad_reduce(a, workspace, dw_packed, db, 1, ns, 1, 0, stream, torch_keep);
  }
  if (use_split) {
    if (prec_flags & ISI_CONV_BF16X6) launch_wgrad_split_np<3>(a, tile, nz * nphase * nsplit, stream);
    else launch_wgrad_split_np<2>(a, tile, nz * nphase * nsplit, stream);
  } else {
    constexpr size_t smem = (size_t)4 * 32 * LDT * sizeof(float);
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_f32_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return check_launch("hipFuncSetAttribute(conv_wgrad)");
      attr_set.mark();
    }
    dim3 grid((Cout + 127) / 128, (a.Kpad + 127) / 128, nz * nphase * nsplit);
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(256), smem, stream, a);
  }
  int rc = check_launch("conv_wgrad_f32");
  if (rc) return rc;
  return wgrad_reduce(a, workspace, dw_packed, db, nphase, nsplit, nz, zs_dw, stream, torch_keep);
}

// Would a plain convolution's weight gradient (three-term products, dense channels-last operands) run the halo-staged
// kernel -- the one that also reads pair-format sources?  (The launch conditions above, for callers that plan tensor formats.)
bool conv_wgrad_halo_route(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, int OH, int OW) {
  const int Cin = C0 + C1;
  const bool k3 = KH == 3 && KW == 3 && stride == 1 && pad == 1, k4 = KH == 4 && KW == 4 && stride == 2 && pad == 1;
  if (Cout % 32 || Cin % 32 || C0 % 32) return false;
  int nco = Cout % 128 == 0 ? 4 : Cout % 64 == 0 ? 2 : 1;
  while (nco < 4 && Cin % (32 * (4 / nco))) nco *= 2;
  return (k3 || k4) && Cout % (32 * nco) == 0 && Cin % (32 * (4 / nco)) == 0 && OW % 32 == 0 && OH % 2 == 0 && !(k4 && nco == 1) &&
         !knobs().no_wgrad_halo;
}

// embed_sum[d][k] = sum over vectors n with idx[n] == k of z[n][d]  ==  z^T @ onehot(idx)
// (bottleneck.py:83), as the same pixel-reduction GEMM with the one-hot operand generated on the fly.
// out: [D][K] like the reference's `embed_avg`; deterministic.
size_t vq_embed_sum_workspace_floats(int D, int K, int64_t N) {
  (void)N;
  return (size_t)256 * D * K;   // at most 256 pixel splits of [D][K] partials
}

int vq_embed_sum_f32(const float *z, const int64_t *idx, float *embed_sum_dk, float *workspace,
                     size_t workspace_floats, int64_t N, int D, int K, hipStream_t stream) {
  if (!z || !idx || !embed_sum_dk || !workspace || N <= 0 || N > INT32_MAX || D <= 0 || (D & 3) || K <= 0 || (K % 32))
    return invalid("vq_embed_sum: bad argument (D % 4 == 0, K % 32 == 0)");
  if ((int64_t)N * D > ((int64_t)1 << 30)) return unsupported("vq_embed_sum: tensor spans 4 GiB or more");
  WgradKArgs a;
  memset(&a, 0, sizeof a);
  a.x0 = z; a.x1 = z; a.dy = z; a.onehot_idx = idx; a.partial = workspace;
  a.x0_bytes = a.x1_bytes = a.dy_bytes = (unsigned)((size_t)N * D * 4);
  a.C0 = K; a.Cin = K; a.vec = 1; a.dvec = 1;
  a.dn = 0; a.dh = 0; a.dw = D;           // "pixels" = vectors: one row of N
  a.H = 1; a.W = (int)N; a.OH = 1; a.OW = (int)N; a.Cout = D; a.K = K; a.Kpad = K;
  a.KW = 1; a.stride = 1; a.pad = 0; a.M = (int)N;
  // split-bf16 kernel with exact pieces (see ONEHOT above); tile 64 x 256 for the usual 64-dimensional codes
  const int tco = D <= 64 ? 64 : 128, tk = D <= 64 ? 256 : 128;
  const int tiles = ((D + tco - 1) / tco) * ((K + tk - 1) / tk);
  const int nchunks = (a.M + 31) / 32;
  int nsplit = std::min(256, std::max(1, 768 / tiles));
  nsplit = std::min(nsplit, std::max(1, nchunks / 8));
  a.nsplit = nsplit; a.chunks_per_split = (nchunks + nsplit - 1) / nsplit; a.nz = 1;
  if (workspace_floats < (size_t)nsplit * D * K) { set_last_error("vq_embed_sum: workspace too small"); return ISI_E_WORKSPACE; }
  if (D <= 64) launch_wgrad_split<3, 1, 2, 2, true>(a, nsplit, stream);
  else launch_wgrad_split<3, 2, 2, 2, true>(a, nsplit, stream);
  int rc = check_launch("vq_embed_sum(wgrad)");
  if (rc) return rc;
  const int64_t per = (int64_t)D * K;
  launch_reduce_partials(workspace, embed_sum_dk, per, nsplit, per, 0, 0, 0, 1, stream);
  return check_launch("vq_embed_sum(reduce)");
}

}  // namespace isi
