// Row-major GEMM of the prior's linear layers with split products, gfx950:
//
//   out[m][n] = sum_k a[m][k] w[n][k] + bias[n] (+ residual[m][n]) (ReLU)        a: [M][K] rows of lda, w: [N][K] dense
//
// fp32 in / out; every product is the three-term sum of two 16-bit pieces per operand on the 16-bit matrix pipe with
// fp32 accumulation -- bf16 pieces (ISI_CONV_BF16X3: ~2^-16 per product, no range limit; the input-gradient GEMMs) or
// f16 pieces of 4 a and 1024 w (ISI_CONV_F16X3: ~2^-22, |a| < 16384, |w| < 64; the forward GEMMs) -- exactly the
// products of conv_igemm_f32.hip's 1x1 case, which these launches used to run on.
//
// Why a kernel of its own.  The implicit-GEMM convolution kernel stages a 32-deep K chunk through ONE LDS stage between
// two barriers and hides the round trip with other workgroups of the CU; at the prior's shapes (M = B S = 8200 rows,
// K, N = 512 .. 2048: 260 .. 1000 tiles of 128 x 128) it runs at 0.16 - 0.21 of the three-term ceiling
// (tools/bench_linear.py: 33 us for the 4.3 GFLOP of a 512 x 512 projection).  Here
//   * a workgroup = 8 waves = 128 x (64 TN) tile, wave tile 32 x (32 TN); 40 - 64 KB of LDS, <= 128 VGPRs: two
//     workgroups = four waves per SIMD, so that one wave's conversions / LDS traffic run under another's MFMAs;
//   * K chunks of 32 through a TWO-stage LDS ring with one barrier per chunk: chunk c + 1 is in registers (global loads
//     issued before the MFMAs of chunk c), converted to pieces and written to the other stage after them;
//   * pieces in LDS as [row][32 k] planes of 64-byte rows, 16-byte slots XOR-swizzled with (row >> 2) & 3: a fragment
//     (8 consecutive k of one row) is one conflict-free ds_read_b128;
//   * no im2col arithmetic: rows are rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "split_bf16.h"
#include "split_f16.h"

namespace isi {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8g __attribute__((ext_vector_type(8)));

struct GemmArgs {
  const float *a, *w, *bias, *res;
  float *out;
  unsigned a_bytes, w_bytes, res_bytes, out_bytes;
  int M, N, K, lda, ldr, ldo, relu;
  const float *w32;       // fp32 weight [N][K] (the tail rows' operand; == w unless w is a pair copy)
  int tail;               // rows [M, M + tail) beyond the tiled rows: fp32 dot products, spread over the workgroups
  int zs_a, zs_w, zs_res, zs_out;                          // element strides between the products of a batch (grid y)
  int win_rpu, lo_slope, lo_base, hi_slope, hi_base;       // GemmExtra's band of non-zero A columns (win_rpu = 0: none)
  const float *gate;      // optional: out = gate[m ldg + n] > 0 ? v * gate_scale : 0
  int ldg;
  unsigned gate_bytes;
  float gate_scale;
  unsigned drop_thresh;   // fused inverted dropout: keep where dropout_keep(seed, m ldo + n, thresh); 0 = none
  float drop_scale;
  uint64_t drop_seed;
  const uint64_t *drop_base;   // device-resident term of the seed (dropout_seed_base(); nullptr: none)
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int BM = 128;
constexpr int kGemmTailRows = 16;   // M % BM up to this many rows: no tile row of their own (gemm_split_f32 below)
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned uniform_off = 0) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, uniform_off, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ int swz(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 16; }

template <bool F16>
__device__ __forceinline__ void split_pieces(const float4 v, const float scale, uint2 &hi, uint2 &lo) {
  if constexpr (F16) f16s::split4(v, scale, hi, lo);
  else split_f4(v, hi, lo);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma(const s16x8g a, const s16x8g b, const f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16s::f16x8, a), __builtin_bit_cast(f16s::f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// phase timestamps (-DISI_MEASURE builds; tools/stamps_gemm.py): workgroup 8, waves 0 and 1
#ifdef ISI_MEASURE
__device__ long long g_gemm_stamps[512];
#define ISI_GEMM_STAMP(i_) do { if (blockIdx.x == 8 && wave < 2 && lane == 0 && (i_) < 256) \
    g_gemm_stamps[wave * 256 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_GEMM_STAMP(i_) do { } while (0)
#endif

// TN: 32-column tiles per wave (tile width BN = 64 TN).  WPRE: `w` is the weight's pair copy in the product mode's
// pieces (ISI_CONV_W16: groups of 8 k as {hi[8] | lo[8]}, 32 bytes where the 8 floats were) -- a 16-byte piece IS a slot of
// the LDS planes, so the weight tile is staged by plain copies: every 128-row tile of the fp32 form converts its
// weight tile again, 65 times at M = 8200.
// TM: 32-row tiles per wave (tile height 128 TM).  TM = TN = 2: a 256 x 128 tile, wave tile 64 x 64 -- 24 matrix
// instructions per 16 fragment reads and chunk instead of 12 per 12 (the 128 x 128 form is co-limited by LDS traffic,
// staging conversions and the vector-memory path, each about as long as its matrix work); 96 KB of LDS, one workgroup per
// CU: the wide layers (N >= 1024) whose tile count still fills the chip.
// TM = 2, TN = 4: a 256 x 256 tile, wave tile 64 x 128 -- 48 matrix instructions per 24 fragment reads and chunk; 128 KB
// of LDS, the whole register file of two waves per SIMD: the widest layers when they come as ONE round of such tiles.
template <bool F16, int TN, bool WPRE = false, int TM = 1>
__global__ __launch_bounds__(512, (TM * TN >= 4 ? 2 : 4)) void gemm_split_kernel(GemmArgs p) {
  constexpr int BN = 64 * TN, BM = 128 * TM;
  if (blockIdx.y) {      // a batch of products: this one's operands
    const size_t z = blockIdx.y;
    p.a += z * p.zs_a; p.w += z * p.zs_w; p.w32 += z * p.zs_w; p.out += z * p.zs_out;
    if (p.res) p.res += z * p.zs_res;
  }
  constexpr int APL = BM * 64, BPL = BN * 64;                  // bytes of one plane (rows of 32 pieces = 64 B)
  constexpr int STAGE = 2 * APL + 2 * BPL;                     // A hi, A lo, B hi, B lo
  constexpr int NA = BM * 8 / 512, NB = BN * 8 / 512;          // float4 per thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ISI_GEMM_STAMP(0);
  const int wm0 = (wave & 3) * (32 * TM), wn0 = (wave >> 2) * (32 * TN);
  const int fr = lane & 31, fh = lane >> 5;
  // consecutive workgroups walk the M tiles of one column block: they share the block's weights in L2
  const int tiles_m = (p.M + BM - 1) / BM;
  const int m0 = ((int)blockIdx.x % tiles_m) * BM, n0 = ((int)blockIdx.x / tiles_m) * BN;

  // (the operands' extents as constants: rows beyond them carry the out-of-range offset OOB, every other row's chunks lie
  // inside -- with the byte counts read from the arguments the compiler re-loaded them, and waited for every outstanding
  // LDS read with them, in front of each group of loads)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, OOB, 0x00020000);

  // staging roles: float4 number tid + 512 j of the chunk: row = (tid + 512 j) / 8, quad = tid % 8 (8 lanes = one
  // 128-byte row segment)
  const int sq = tid & 7, sr = tid >> 3;
  unsigned aoff[NA], boff[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + sr + 64 * j;
    aoff[j] = m < p.M ? (unsigned)(m * p.lda + sq * 4) * 4u : OOB;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + sr + 64 * j;
    boff[j] = n < p.N ? (unsigned)(n * p.K + sq * 4) * 4u : OOB;
  }
  float4 pa[NA], pb[NB];
  // The next chunk travels in TN groups of pieces (group g: A pieces g NA / TN .., B piece g): requested one group after
  // each column tile's matrix instructions of the chunk's first half, converted and written to the other stage one group
  // after each column tile's of the second half.  (All eight loads of a 256 x 256 tile's thread issued back to back took 900
  // cycles to issue -- 16 waves x 8 KB through the 64 B/clk vector-memory path -- and the wave then waited ~2000 cycles for
  // the last of them in front of its LDS stores: 5600 cycles per chunk where the matrix work is 3072, tools/stamps_gemm.py.)
  constexpr int APG = NA / TN;
  static_assert(NA % TN == 0 && NB == TN, "pieces per group");
  auto load_group = [&](int k0, int g) {
#pragma unroll
    // (the chunk's offset travels in the instruction's scalar offset, which the range check ignores: a row beyond the
    // operand keeps its out-of-range vector offset, and no address is recomputed per load)
    for (int j = g * APG; j < (g + 1) * APG; ++j) pa[j] = buf_load4(ra, aoff[j], (unsigned)k0 * 4u);
    pb[g] = buf_load4(rw, boff[g], (unsigned)k0 * 4u);
  };
  auto store_group = [&](int stage, int g) {
    unsigned char *st = smem + stage * STAGE;
#pragma unroll
    for (int j = g * APG; j < (g + 1) * APG; ++j) {
      const int row = sr + 64 * j;
      uint2 hi, lo;
      split_pieces<F16>(pa[j], f16s::kScaleA, hi, lo);
      const int o = row * 64 + swz(row, sq >> 1) + (sq & 1) * 8;
      *reinterpret_cast<uint2 *>(st + o) = hi;
      *reinterpret_cast<uint2 *>(st + APL + o) = lo;
    }
    {
      const int row = sr + 64 * g;
      if constexpr (WPRE) {   // piece sq of the row's 128 bytes: group sq / 2, plane sq % 2
        *reinterpret_cast<float4 *>(st + 2 * APL + (sq & 1) * BPL + row * 64 + swz(row, sq >> 1)) = pb[g];
      } else {
        uint2 hi, lo;
        split_pieces<F16>(pb[g], f16s::kScaleB, hi, lo);
        const int o = row * 64 + swz(row, sq >> 1) + (sq & 1) * 8;
        *reinterpret_cast<uint2 *>(st + 2 * APL + o) = hi;
        *reinterpret_cast<uint2 *>(st + 2 * APL + BPL + o) = lo;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunk = p.K / 32;
  int c_lo = 0, c_hi = nchunk;
  if (p.win_rpu) {       // the K chunks that hold this tile's band of non-zero A columns (GemmExtra)
    const int u0 = m0 / p.win_rpu, u1 = min(p.M - 1, m0 + BM - 1) / p.win_rpu;
    const int klo = p.lo_slope * u0 + p.lo_base, khi = p.hi_slope * u1 + p.hi_base;
    c_lo = max(klo, 0) / 32;
    c_hi = khi < 0 ? c_lo : min(nchunk, khi / 32 + 1);
    c_lo = min(c_lo, c_hi);
  }
  if (c_lo < c_hi) {
#pragma unroll
    for (int g = 0; g < TN; ++g) load_group(32 * c_lo, g);
#pragma unroll
    for (int g = 0; g < TN; ++g) {
      store_group(0, g);
      if (c_lo + 1 < c_hi) load_group(32 * (c_lo + 1), g);
    }
  }
  __syncthreads();
  ISI_GEMM_STAMP(1);
  // A chunk = 2 TN units (k half s, column tile j) of 3 TM matrix instructions.  The fragments of unit u + 1 are read
  // from LDS before unit u's matrix instructions are issued (two fragment sets); behind unit (1, j) the next chunk's
  // group j of pieces is written to the other stage and, into the registers this frees, the chunk after that is requested:
  // a request has a whole chunk's matrix work to come back (with half a chunk, a 256 x 256 tile's second half waited:
  // 2500 cycles against 1300 for the first).  (With each unit reading its own fragments first, a
  // wave's chain per unit was LDS latency + matrix time: 2400 cycles per half chunk where the matrix work of the SIMD's two
  // waves is 1536, tools/stamps_gemm.py.)
  s16x8g ah[2][TM], al[2][TM], bh[2], bl[2];
  auto read_a = [&](const unsigned char *st, int s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int arow = wm0 + 32 * i + fr;
      const int ao = arow * 64 + swz(arow, 2 * s + fh);
      ah[s][i] = *reinterpret_cast<const s16x8g *>(st + ao);
      al[s][i] = *reinterpret_cast<const s16x8g *>(st + APL + ao);
    }
  };
  auto read_b = [&](const unsigned char *st, int s, int j, int slot) {
    const int brow = wn0 + 32 * j + fr;
    const int bo = brow * 64 + swz(brow, 2 * s + fh);
    bh[slot] = *reinterpret_cast<const s16x8g *>(st + 2 * APL + bo);
    bl[slot] = *reinterpret_cast<const s16x8g *>(st + 2 * APL + BPL + bo);
  };
  // (No branch inside a chunk: behind one the compiler's wait counts fall back to "every request", and each group then
  // waited for the loads issued just before it.  The last chunk is peeled; the chunk before it requests the last one once
  // more instead of nothing.)
  auto chunk = [&](const int c, auto last) {
    constexpr bool LAST = decltype(last)::value;
    ISI_GEMM_STAMP(4 + 4 * (c - c_lo));
    const unsigned char *st = smem + ((c - c_lo) & 1) * STAGE;
    const int k0_next = 32 * min(c + 2, c_hi - 1);
    read_a(st, 0);
    read_b(st, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2 * TN; ++u) {
      const int s = u / TN, j = u % TN;
      if (u + 1 < 2 * TN) {
        if ((u + 1) % TN == 0) read_a(st, 1);
        read_b(st, (u + 1) / TN, (u + 1) % TN, (u + 1) & 1);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        acc[i][j] = mfma<F16>(al[s][i], bh[u & 1], acc[i][j]);
        acc[i][j] = mfma<F16>(ah[s][i], bl[u & 1], acc[i][j]);
        acc[i][j] = mfma<F16>(ah[s][i], bh[u & 1], acc[i][j]);
      }
      if (s == 1 && !LAST) {
        store_group((c + 1 - c_lo) & 1, j);   // that stage was last read in iteration c - 1: every wave is past its barrier
        load_group(k0_next, j);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u == TN - 1) ISI_GEMM_STAMP(5 + 4 * (c - c_lo));
    }
    ISI_GEMM_STAMP(6 + 4 * (c - c_lo));
    ISI_GEMM_STAMP(7 + 4 * (c - c_lo));
    __syncthreads();
  };
  for (int c = c_lo; c + 1 < c_hi; ++c) chunk(c, std::false_type{});
  if (c_lo < c_hi) chunk(c_hi - 1, std::true_type{});
  ISI_GEMM_STAMP(2);

  const uint64_t drop_seed = p.drop_seed + (p.drop_thresh && p.drop_base ? *p.drop_base : 0);
  // ---- epilogue.  C layout of a 32 x 32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a
  // store instruction writes two 128-byte row segments
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res : p.a), 0, p.res ? p.res_bytes : 4u, 0x00020000);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + 32 * j + fr;
    const bool nok = n < p.N;
    const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
    float res[16];
    unsigned oo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const bool ok = nok && m < p.M;
      oo[r] = ok ? (unsigned)(m * p.ldo + n) * 4u : OOB;
      res[r] = 0.f;
      if (p.res) res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ok ? (unsigned)(m * p.ldr + n) * 4u : OOB, 0, 0));
    }
    float gt[16];
    if (p.gate) {       // (requested with the residual: all of an epilogue's loads are in flight before its stores)
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gate), 0, p.gate_bytes, 0x00020000);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        gt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, oo[r] == OOB ? OOB : (unsigned)(m * p.ldg + n) * 4u, 0, 0));
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = (F16 ? acc[i][j][r] * f16s::kUnscale : acc[i][j][r]) + bias + res[r];
      if (p.relu) v = fmaxf(v, 0.f) + (v - v);      // (a NaN stays a NaN: an operand beyond the f16 range must be loud)
      if (p.gate) v = gt[r] > 0.f ? v * p.gate_scale : 0.f;
      if (p.drop_thresh) v = dropout_keep(drop_seed, oo[r] >> 2, p.drop_thresh) ? v * p.drop_scale : 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ro, oo[r], 0, 0);
    }
  }
  ISI_GEMM_STAMP(3);
  // ---- the few rows beyond the last full tile row (gemm_split_f32): output columns dealt round-robin to the
  // workgroups, one row per wave, plain fp32 dot products (one memory round trip at the end of every workgroup
  // instead of a second round of mostly empty tiles)
  if (p.tail) {
    const int nq = p.K >> 2;
    for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
      const float4 *wr = reinterpret_cast<const float4 *>(p.w32 + (size_t)n * p.K);
      for (int r = wave; r < p.tail; r += 8) {
        const size_t m = (size_t)p.M + r;
        const float4 *xr = reinterpret_cast<const float4 *>(p.a + m * p.lda);
        int q_lo = 0, q_hi = nq;
        if (p.win_rpu) {      // the row's band of non-zero columns (the rest of a banded operand may be undefined)
          const int u = (int)(m / p.win_rpu);
          const int klo = p.lo_slope * u + p.lo_base, khi = p.hi_slope * u + p.hi_base;
          q_lo = max(klo, 0) >> 2;
          q_hi = khi < 0 ? q_lo : min(nq, (khi >> 2) + 1);
        }
        float s = 0.f;
        for (int qd = q_lo + lane; qd < q_hi; qd += 64) {
          const float4 wv = wr[qd], xv = xr[qd];
          s += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
        }
        s = wave64_sum(s);
        if (lane == 0) {
          float v = s + (p.bias ? p.bias[n] : 0.f) + (p.res ? p.res[m * p.ldr + n] : 0.f);
          if (p.relu) v = fmaxf(v, 0.f) + (v - v);
          if (p.gate) v = p.gate[m * p.ldg + n] > 0.f ? v * p.gate_scale : 0.f;
          if (p.drop_thresh) v = dropout_keep(drop_seed, (uint32_t)(m * p.ldo + n), p.drop_thresh) ? v * p.drop_scale : 0.f;
          p.out[m * p.ldo + n] = v;
        }
      }
    }
  }
}

template <bool F16, int TN, bool WPRE = false, int TM = 1>
int launch_gemm(const GemmArgs &a, hipStream_t stream, int nz = 1) {
  constexpr int BN = 64 * TN, BMT = 128 * TM;
  constexpr size_t smem = (size_t)2 * (2 * BMT * 64 + 2 * BN * 64);
  static DeviceOnce attr_set;
  if (smem > 48 * 1024 && !attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_split_kernel<F16, TN, WPRE, TM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(gemm_split)");
    attr_set.mark();
  }
  const int tiles = ((a.M + BMT - 1) / BMT) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_split_kernel<F16, TN, WPRE, TM>), dim3(tiles, nz), dim3(512), smem, stream, a);
  return check_launch("gemm_split_f32");
}

}  // namespace

int gemm_split_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), sizeof(long long) * (size_t)(n < 512 ? n : 512)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

// Shapes this kernel takes over from the 1x1 implicit GEMM (conv2d_batched_f32 asks before it plans its own launch):
// dense rows (channel stride 1), K a multiple of 32, 16-byte aligned operands, three-term products.
bool gemm_split_applicable(int M, int N, int K, int split_mode) {
  return (split_mode == 1 || split_mode == 3) && K % 32 == 0 && K >= 128 && N > 32 && M >= 256;   // (K, N: where the convolution kernel uses split products too)
}

int gemm_split_f32(const float *a, int64_t lda, const float *w, const float *bias, const float *res, int64_t ldr, float *out,
                   int64_t ldo, int M, int N, int K, int relu, int split_mode, hipStream_t stream, const float *w16,
                   const GemmExtra *extra) {
  // A few rows beyond a multiple of the tile height (the prior's B x (S + 1) = 8 x 1025 = 64 tiles + 8 rows) would add a
  // whole column of 128-row tiles: 520 instead of 512 for N = 512, and with two workgroups per CU the launch then
  // runs TWO rounds (28 vs 22 us at K = 512, 92 vs 66 us at K = 2048, measured).  The tiles cover the full tile rows
  // only and every workgroup finishes with its share of the remaining rows' dot products (fp32 FMA, exact operands;
  // the few-row kernel of the decode path as a second launch was measured too: 10 us, as much as it saved).
  GemmArgs g;
  g.a = a; g.w = w16 ? w16 : w; g.bias = bias; g.res = res; g.out = out;
  g.M = M; g.N = N; g.K = K; g.lda = (int)lda; g.ldr = (int)ldr; g.ldo = (int)ldo; g.relu = relu;
  const int64_t ea = (int64_t)(M - 1) * lda + K, eo = (int64_t)(M - 1) * ldo + N, er = res ? (int64_t)(M - 1) * ldr + N : 1;
  const int64_t lim = (int64_t)1 << 30;
  if (ea > lim || eo > lim || er > lim || (int64_t)N * K > lim) return unsupported("gemm: a tensor spans 4 GiB or more");
  g.a_bytes = (unsigned)(ea * 4); g.w_bytes = (unsigned)((int64_t)N * K * 4); g.out_bytes = (unsigned)(eo * 4);
  g.res_bytes = (unsigned)(er * 4);
  g.w32 = w; g.tail = 0;
  g.zs_a = g.zs_w = g.zs_res = g.zs_out = 0;
  g.win_rpu = g.lo_slope = g.lo_base = g.hi_slope = g.hi_base = 0;
  g.gate = nullptr; g.ldg = 0; g.gate_bytes = 4; g.gate_scale = 1.f;
  g.drop_thresh = 0; g.drop_scale = 1.f; g.drop_seed = 0; g.drop_base = nullptr;
  int nz = 1;
  if (extra) {
    nz = extra->nz;
    if (extra->drop_p > 0.f) {
      if (!(extra->drop_p < 1.f) || nz != 1 || eo > ((int64_t)1 << 30)) return unsupported("gemm: dropout needs 0 < p < 1 and a single product");
      const double t = (double)extra->drop_p * 4294967296.0;
      g.drop_thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (t < 1.0 ? 1u : (unsigned)t);
      g.drop_scale = 1.f / (1.f - extra->drop_p);
      g.drop_seed = extra->drop_seed;
      g.drop_base = dropout_seed_base();
    }
    if (extra->gate) {
      const int64_t eg = (int64_t)(M - 1) * extra->ldg + N;
      if (nz != 1 || eg > lim || extra->ldg < N) return unsupported("gemm: a gate goes with a single product of less than 4 GiB");
      g.gate = extra->gate; g.ldg = (int)extra->ldg; g.gate_bytes = (unsigned)(eg * 4);
      g.gate_scale = extra->gate_scale != 0.f ? extra->gate_scale : 1.f;
    }
    if (nz < 1 || nz > 65535) return invalid("gemm: bad batch count");
    const int64_t zmax = std::max(std::max(extra->zs_a, extra->zs_w), std::max(extra->zs_res, extra->zs_out));
    if (zmax >= ((int64_t)1 << 31) || (extra->zs_a | extra->zs_w | extra->zs_res | extra->zs_out) < 0)
      return unsupported("gemm: batch stride out of range");
    g.zs_a = (int)extra->zs_a; g.zs_w = (int)extra->zs_w; g.zs_res = (int)extra->zs_res; g.zs_out = (int)extra->zs_out;
    if (extra->win_rpu > 0) {
      if (extra->lo_slope < 0 || extra->hi_slope < 0) return invalid("gemm: band slopes must be non-negative");
      g.win_rpu = extra->win_rpu; g.lo_slope = extra->lo_slope; g.lo_base = extra->lo_base;
      g.hi_slope = extra->hi_slope; g.hi_base = extra->hi_base;
    }
  }
  static_assert(BM <= kBandTileMax, "banded launches (win_rpu) run 128-row tiles: what kMargin of the attention backward covers");
  const int rem = M % BM;
  if (rem > 0 && rem <= kGemmTailRows && M >= 8 * BM && (lda & 3) == 0) { g.tail = rem; g.M = M - rem; M = g.M; }
  // 128 x 64 tiles when 128 x 128 ones would leave CUs without a second workgroup
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  // (from one workgroup per CU on: M = 8200, N = 512, K = 2048 72 -> 66 us with 128 x 128 tiles, K = 512 the same either way)
  const int narrow_below = knobs().gemm_narrow_below > 0 ? knobs().gemm_narrow_below : current_device_cu_count();
  const bool narrow = tiles128 < narrow_below || N % 128 != 0;
  // 256 x 128 tiles (one workgroup per CU) where they come in whole rounds of the chip: the widest layers (measured at
  // M = 8200, K = 512, tools/bench_linear.py with ISI_GEMM_NO_WIDE: N = 2048 (two rounds) 79 -> 70 us, N = 1024 (one round)
  // 42 -> 40 us; N = 1536 (one and a half rounds) 63 -> 70 us: not taken)
  const long tiles256 = (long)((M + 255) / 256) * ((N + 127) / 128);
  const long tiles256x256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  if (!narrow && nz == 1 && !g.win_rpu && N % 256 == 0 && tiles256x256 == current_device_cu_count() && !knobs().gemm_no_wide) {
    if (split_mode == 3 && w16) return launch_gemm<true, 4, true, 2>(g, stream, nz);
    if (split_mode == 3) return launch_gemm<true, 4, false, 2>(g, stream, nz);
    if (w16) return launch_gemm<false, 4, true, 2>(g, stream, nz);
    return launch_gemm<false, 4, false, 2>(g, stream, nz);
  }
  if (!narrow && nz == 1 && !g.win_rpu && tiles256 % current_device_cu_count() == 0 && knobs().gemm_no_wide != 1) {
    if (split_mode == 3 && w16) return launch_gemm<true, 2, true, 2>(g, stream, nz);
    if (split_mode == 3) return launch_gemm<true, 2, false, 2>(g, stream, nz);
    if (w16) return launch_gemm<false, 2, true, 2>(g, stream, nz);
    return launch_gemm<false, 2, false, 2>(g, stream, nz);
  }
  if (split_mode == 3 && w16) return narrow ? launch_gemm<true, 1, true>(g, stream, nz) : launch_gemm<true, 2, true>(g, stream, nz);
  if (split_mode == 3) return narrow ? launch_gemm<true, 1>(g, stream, nz) : launch_gemm<true, 2>(g, stream, nz);
  if (w16) return narrow ? launch_gemm<false, 1, true>(g, stream, nz) : launch_gemm<false, 2, true>(g, stream, nz);
  return narrow ? launch_gemm<false, 1>(g, stream, nz) : launch_gemm<false, 2>(g, stream, nz);
}

}  // namespace isi
