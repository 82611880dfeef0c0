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

#include "isi_common.h"
#include "isi_internal.h"

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
};

namespace {
constexpr int LDT = 132;  // LDS row (floats): 128 + 4
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
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

template <int NP>   // 2: three-term split, 3: six-term split
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const WgradKArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Dp = reinterpret_cast<unsigned short *>(smem);   // [NP][128 co][LDB]  dY^T pieces (0 hi, 1 lo, 2 mid)
  unsigned short *Xp = Dp + NP * 128 * LDB;                        // [NP][128 k ][LDB]  im2col^T pieces
  float *bias_s = reinterpret_cast<float *>(Xp + NP * 128 * LDB);  // [8][128] bias partials of the pixel groups
  constexpr int PS = 128 * LDB;

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

  // staging role: pixel group pg (pixels 4 pg .. 4 pg + 3 of the 32-pixel chunk), channel quad sq
  const int pg = tid >> 5, sq = tid & 31;
  const int kk = k0 + sq * 4;
  const bool kvalid = kk < p.K;
  int tap = 0, c = 0, kh = 0, kw = 0;
  if (kvalid) { tap = kk / p.Cin; c = kk - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW; }
  const bool second = c >= p.C0;
  const int cc = second ? c - p.C0 : c;
  const int co = co0 + sq * 4;
  const bool covalid = co < p.Cout;

  const int chunk_begin = split * p.chunks_per_split;
  const int nchunks_total = (p.M + 31) / 32;
  const int chunk_end = min(nchunks_total, chunk_begin + p.chunks_per_split);

  float4 rdq[4], rxq[4];
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
      unsigned doff = OOB, xoff = OOB;
      if (m < p.M) {
        const int b = pb[j], oy = py_[j], ox = px4[j];
        if (covalid) doff = (unsigned)(dy_off + b * p.dn + oy * p.dh + ox * p.dw + co) * 4u;
        const int iy = oy * p.stride - pad_y + kh, ix = ox * p.stride - pad_x + kw;
        if (kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          xoff = second ? (unsigned)(b * p.s1n + iy * p.s1h + ix * p.s1w + cc) * 4u
                        : (unsigned)(b * p.s0n + iy * p.s0h + ix * p.s0w + cc) * 4u;
      }
      rdq[j] = buf_load4(rd, doff);
      rxq[j] = second ? buf_load4(r1, xoff) : buf_load4(r0, xoff);
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
  auto store_chunk = [&]() {
    bsum[0] += (rdq[0].x + rdq[1].x) + (rdq[2].x + rdq[3].x);
    bsum[1] += (rdq[0].y + rdq[1].y) + (rdq[2].y + rdq[3].y);
    bsum[2] += (rdq[0].z + rdq[1].z) + (rdq[2].z + rdq[3].z);
    bsum[3] += (rdq[0].w + rdq[1].w) + (rdq[2].w + rdq[3].w);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = sq * 4 + e;
      const int wo = row * LDB + bf_slot(row, pg >> 1) + (pg & 1) * 4;
      uint2 pd[3], px_[3];
      const float d0 = e == 0 ? rdq[0].x : e == 1 ? rdq[0].y : e == 2 ? rdq[0].z : rdq[0].w;
      const float d1 = e == 0 ? rdq[1].x : e == 1 ? rdq[1].y : e == 2 ? rdq[1].z : rdq[1].w;
      const float d2 = e == 0 ? rdq[2].x : e == 1 ? rdq[2].y : e == 2 ? rdq[2].z : rdq[2].w;
      const float d3 = e == 0 ? rdq[3].x : e == 1 ? rdq[3].y : e == 2 ? rdq[3].z : rdq[3].w;
      const float x0 = e == 0 ? rxq[0].x : e == 1 ? rxq[0].y : e == 2 ? rxq[0].z : rxq[0].w;
      const float x1 = e == 0 ? rxq[1].x : e == 1 ? rxq[1].y : e == 2 ? rxq[1].z : rxq[1].w;
      const float x2 = e == 0 ? rxq[2].x : e == 1 ? rxq[2].y : e == 2 ? rxq[2].z : rxq[2].w;
      const float x3 = e == 0 ? rxq[3].x : e == 1 ? rxq[3].y : e == 2 ? rxq[3].z : rxq[3].w;
      split4<NP>(d0, d1, d2, d3, pd);
      split4<NP>(x0, x1, x2, x3, px_);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        *reinterpret_cast<uint2 *>(Dp + q * PS + wo) = pd[q];
        *reinterpret_cast<uint2 *>(Xp + q * PS + wo) = px_[q];
      }
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
  if (chunk_begin < chunk_end) {
    load_chunk(chunk_begin);
    store_chunk();
  }
  __syncthreads();
  for (int ch = chunk_begin; ch < chunk_end; ++ch) {
    if (ch + 1 < chunk_end) load_chunk(ch + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 av[2][NP], bv[2][NP];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm0 + i * 32 + fl, rb = wn0 + i * 32 + fl;
        const int ao = ra * LDB + bf_slot(ra, s * 2 + half), bo = rb * LDB + bf_slot(rb, s * 2 + half);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          av[i][q] = *reinterpret_cast<const s16x8 *>(Dp + q * PS + ao);
          bv[i][q] = *reinterpret_cast<const s16x8 *>(Xp + q * PS + bo);
        }
      }
#define ISI_MF(i, j, qa, qb)                                                                                       \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i][qa]),                       \
                                                      __builtin_bit_cast(bf16x8, bv[j][qb]), acc[i][j], 0, 0, 0)
#define ISI_TERM(qa, qb) ISI_MF(0, 0, qa, qb); ISI_MF(0, 1, qa, qb); ISI_MF(1, 0, qa, qb); ISI_MF(1, 1, qa, qb)
      if constexpr (NP == 3) {   // smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
        ISI_TERM(1, 0); ISI_TERM(0, 1); ISI_TERM(2, 2); ISI_TERM(2, 0); ISI_TERM(0, 2); ISI_TERM(0, 0);
      } else {
        ISI_TERM(1, 0); ISI_TERM(0, 1); ISI_TERM(0, 0);
      }
#undef ISI_TERM
#undef ISI_MF
    }
    __syncthreads();   // every wave has read the single stage
    if (ch + 1 < chunk_end) store_chunk();
    __syncthreads();
  }
  // ---- bias partials: fixed-order sum over the 8 pixel groups
  if (p.db_partial != nullptr && blockIdx.y == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bias_s[pg * 128 + sq * 4 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && co0 + tid < p.Cout) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += bias_s[g * 128 + tid];
      p.db_partial[(size_t)blockIdx.z * p.Cout + co0 + tid] = t;
    }
  }

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

// out[i] = sum_s partial[s * stride + i]  (fixed order: 4 interleaved split groups, then a fixed tree)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial,
                                                              float *__restrict__ out, int64_t n, int nsplit,
                                                              int64_t stride, int accumulate, int64_t zs_partial,
                                                              int64_t zs_out) {
  __shared__ float red[4][64];
  partial += (size_t)blockIdx.y * zs_partial;   // grid y: independent reductions
  out += (size_t)blockIdx.y * zs_out;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < n)
    for (int k = g; k < nsplit; k += 4) s += partial[(size_t)k * stride + i];
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    const float t = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    out[i] = accumulate ? out[i] + t : t;
  }
}

static bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static int64_t extent4(int64_t n, int64_t sn, int64_t c, int64_t sc, int64_t h, int64_t sh, int64_t w, int64_t sw) {
  return (n - 1) * sn + (c - 1) * sc + (h - 1) * sh + (w - 1) * sw + 1;
}

static int wgrad_nsplit(int Cout, int Kpad, int M, int nphase, int nz) {
  const int tiles = ((Cout + 127) / 128) * ((Kpad + 127) / 128) * nphase * nz;
  const int nchunks = (M + 31) / 32;
  const int nsplit = std::min(256, std::max(1, 768 / tiles));
  return std::min(nsplit, std::max(1, nchunks / 8));
}
size_t conv_wgrad_batched_workspace_floats(int Cout, int K, int M, int nphase, int nz) {
  const int Kpad = (int)round_up(K, kBK);
  const int nsplit = wgrad_nsplit(Cout, Kpad, M, nphase, nz);
  return (size_t)nz * ((size_t)nsplit * nphase * Cout * Kpad + (size_t)nsplit * nphase * Cout);
}
size_t conv_wgrad_workspace_floats(int Cout, int K, int M, int nphase) {
  return conv_wgrad_batched_workspace_floats(Cout, K, M, nphase, 1);
}

// dW packed like the forward weights: [nphase][Cout][Kpad].  x = layer input (two sources allowed), dy = gradient
// of the layer output, dense channels-last [B, OH(, *2), OW(, *2), Cout].
int conv_wgrad_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                   float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                   int stride, int pad, int transposed, hipStream_t stream) {
  return conv_wgrad_batched_f32(s0, s1, dy, dw_packed, db, workspace, workspace_floats, B, H, W, Cout, KH, KW, stride,
                                pad, transposed, 1, 0, 0, 0, stream);
}

// nz independent weight-gradient GEMMs of one shape in a single launch: pair z reads x at s0->ptr + z * zs_x0
// and dY at dy + z * zs_dy and writes dw_packed + z * zs_dw (element strides).  One source, no bias when nz > 1.
int conv_wgrad_batched_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                           float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                           int stride, int pad, int transposed, int nz, int64_t zs_x0, int64_t zs_dy, int64_t zs_dw,
                           hipStream_t stream) {
  if (!s0 || !s0->ptr || !dy || !dw_packed || !workspace) return invalid("conv_wgrad: null pointer");
  if (nz < 1 || nz > 255) return invalid("conv_wgrad: bad batch count");
  if (nz > 1 && ((s1 && s1->ptr) || db)) return unsupported("conv_wgrad: batched launches take one source and no bias");
  if (nz > 1 && ((zs_x0 | zs_dy | zs_dw) & 3)) return invalid("conv_wgrad: batch strides must be multiples of 4 floats");
  if (zs_x0 < 0 || zs_dy < 0 || zs_x0 >= ((int64_t)1 << 31) || zs_dy >= ((int64_t)1 << 31))
    return unsupported("conv_wgrad: batch stride out of range");
  const int prec_flags = transposed & (ISI_CONV_BF16X3 | ISI_CONV_BF16X6);   // product mode rides in the flag word
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
  const int nsplit = wgrad_nsplit(Cout, a.Kpad, a.M, nphase, nz);
  a.nsplit = nsplit; a.chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  a.nz = nz; a.zs_x0 = (int)zs_x0; a.zs_dy = (int)zs_dy;
  const size_t need = (size_t)nz * ((size_t)nsplit * nphase * Cout * a.Kpad + (size_t)nsplit * nphase * Cout);
  if (workspace_floats < need) { set_last_error("conv_wgrad: workspace too small"); return ISI_E_WORKSPACE; }
  a.db_partial = db ? workspace + (size_t)nsplit * nphase * Cout * a.Kpad : nullptr;
  constexpr size_t smem = (size_t)4 * 32 * LDT * sizeof(float);
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_f32_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_wgrad)");
    attr_set.mark();
  }
  dim3 grid((Cout + 127) / 128, (a.Kpad + 127) / 128, nz * nphase * nsplit);
  // split-bf16 products need the vectorised loaders (channels-last sources, Cout % 4 == 0)
  if (prec_flags && a.vec && a.dvec) {
    const int np = (prec_flags & ISI_CONV_BF16X6) ? 3 : 2;
    const size_t smem_s = (size_t)2 * np * 128 * LDB * sizeof(unsigned short) + 8 * 128 * sizeof(float);
    if (np == 3) hipLaunchKernelGGL(conv_wgrad_split_kernel<3>, grid, dim3(256), smem_s, stream, a);
    else hipLaunchKernelGGL(conv_wgrad_split_kernel<2>, grid, dim3(256), smem_s, stream, a);
  } else {
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(256), smem, stream, a);
  }
  int rc = check_launch("conv_wgrad_f32");
  if (rc) return rc;
  // partial layout: [phase][split][Cout][Kpad] -> out [phase][Cout][Kpad]
  const int64_t per = (int64_t)Cout * a.Kpad;
  for (int ph = 0; ph < nphase; ++ph) {   // grid y = operand set: partials [z][phase][split][Cout][Kpad]
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((per + 63) / 64), nz), dim3(256), 0, stream,
                       workspace + (size_t)ph * nsplit * per, dw_packed + (size_t)ph * per, per, nsplit, per, 0,
                       (int64_t)nphase * nsplit * per, zs_dw);
  }
  rc = check_launch("reduce_partials");
  if (rc || !db) return rc;
  // bias gradient: every (phase, split) partial covers a disjoint pixel set
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((Cout + 63) / 64), dim3(256), 0, stream, a.db_partial, db,
                     (int64_t)Cout, nsplit * nphase, (int64_t)Cout, 0, (int64_t)0, (int64_t)0);
  return check_launch("reduce_partials(bias)");
}

// embed_sum[d][k] = sum over vectors n with idx[n] == k of z[n][d]  ==  z^T @ onehot(idx)
// (bottleneck.py:83), as the same pixel-reduction GEMM with the one-hot operand generated on the fly.
// out: [D][K] like the reference's `embed_avg`; deterministic.
size_t vq_embed_sum_workspace_floats(int D, int K, int64_t N) { return conv_wgrad_workspace_floats(D, K, (int)N, 1); }

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
  const int tiles = ((D + 127) / 128) * ((K + 127) / 128);
  const int nchunks = (a.M + 31) / 32;
  int nsplit = std::min(256, std::max(1, 768 / tiles));
  nsplit = std::min(nsplit, std::max(1, nchunks / 8));
  a.nsplit = nsplit; a.chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  if (workspace_floats < (size_t)nsplit * D * K) { set_last_error("vq_embed_sum: workspace too small"); return ISI_E_WORKSPACE; }
  constexpr size_t smem = (size_t)4 * 32 * LDT * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_f32_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return check_launch("hipFuncSetAttribute(conv_wgrad)");
  hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3((D + 127) / 128, (K + 127) / 128, nsplit), dim3(256), smem, stream, a);
  int rc = check_launch("vq_embed_sum(wgrad)");
  if (rc) return rc;
  const int64_t per = (int64_t)D * K;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, stream, workspace,
                     embed_sum_dk, per, nsplit, per, 0, (int64_t)0, (int64_t)0);
  return check_launch("vq_embed_sum(reduce)");
}

}  // namespace isi
