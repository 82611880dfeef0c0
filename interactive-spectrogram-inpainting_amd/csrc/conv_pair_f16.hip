// Implicit-GEMM convolution of the split-f16 PAIR pipeline for gfx950 (the dominant kernel of VQVAE.forward).
//
//   out[m][n] = sum_k A[m][k] B[n][k]        M = B*OH*OW pixels, N = Cout, K = KH*KW*Cin
//   every product as hi.hi + hi.lo + lo.hi of two 11-bit f16 pieces (split_f16.h) on v_mfma_f32_32x32x16_f16
//
// Same arithmetic as conv_igemm_f32.hip's ISI_CONV_F16X3 | ISI_CONV_W16 path -- with the accumulator flush off the
// two kernels return the same bits -- restructured around what that kernel's counters showed (matrix pipe 46 % busy
// on 128x128 tiles, 25-30 % on 128x64; waves parked at two barriers per 32-deep chunk; an LDS pipe loaded by 16
// ds_write_b64 + 16 v_perm per thread and chunk):
//
//   * activations arrive in the pair8 format and weights in its k-blocked twin: a 16-byte piece of either IS an MFMA
//     operand fragment, so staging is a pure copy, done by the LDS-DMA path (`buffer_load_dwordx4 ... lds`) -- no
//     staging registers, no conversion, no ds_write.  The DMA is issued from inline asm and retired by COUNTED
//     `s_waitcnt vmcnt(N)`: hipcc would wait vmcnt(0) for it at every barrier.
//   * 256 x BN tile (BN = 128 / 64), 8 waves of 64 x BN/2, K walked in 32-channel chunks through a THREE-stage LDS
//     ring: chunk k+2 is in flight while chunk k is multiplied, ONE barrier per chunk, 24 (12) MFMAs per wave between
//     barriers but with two waves per SIMD covering each other's waits.
//   * LDS rows are 128 bytes (8 pieces: hi/lo of four channel groups), the piece index XOR-swizzled with
//     (row >> 1) & 7 ON THE SOURCE side of the DMA (the destination of a DMA is lane-linear): every 16-lane group of
//     a ds_read_b128 fragment read covers all 16 slots of the bank row.
//   * zero padding = out-of-range buffer offset (the DMA writes zeros, tools/probes/lds_dma_probe.hip).
//   * accuracy: the accumulator of a tile is FLUSHED into a second one every `flush` chunks (pairwise-style
//     summation: a K = 1152 product-sum is 216 sequential fp32 roundings at the full magnitude in one accumulator,
//     18 at a twelfth of the variance plus 12 with flush = 3), which takes the error against fp64 from 0.9e-6 of the
//     maximum to below torch-CPU's own.
//   * the MFMAs run with SWAPPED operands (weights = rows, pixels = columns: the same products in the same k order,
//     the same bits), so a lane's accumulators are 4 x 4 consecutive channels of ONE pixel per 32-channel tile: the
//     epilogue stores straight from the registers -- fp32 quads, or pair8 pieces after v_permlane32_swap has handed
//     the lower half-wave both halves' hi quads and the upper one the lo quads -- with no LDS transpose and no block
//     barrier (the transposed epilogue cost 7-9 k of a work item's ~97 k cycles; 3x3 128->128: 212 -> 194 us).
//     Tried on top of it and dropped: the stage ring running THROUGH the work items of a workgroup (next item's tables
//     written during the current K loop, its first two chunks requested by the last two memory phases, vmcnt(PER +
//     stores) behind the epilogue): correct, but its loop carries the issue context as loop-variant state and ran 7 %
//     slower per chunk, 2.4 % slower overall on the large layers (198.6 vs 193.9 us).
//
// Replaces (reference, torch.nn): nn.Conv2d / nn.ConvTranspose2d / nn.ReLU / torch.cat at
// vqvae/encoder_decoder.py:95-112,138,199-215 and vqvae/vqvae.py:193-201,260,270-272,282.
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
typedef short s16x8 __attribute__((ext_vector_type(8)));
using f16s::f16x8;

constexpr int BM = 256;            // pixels per workgroup
constexpr int NS = 3;              // LDS stages
constexpr int ROWB = 128;          // bytes of one tile row per chunk: 32 channels x {hi, lo} f16
constexpr unsigned OOB = 0x7FFFFFF0u;   // beyond every descriptor (tensors are below 1.75 GiB)
constexpr unsigned OOB_STORE = 0xFFFFFFF0u;

struct PairK {
  const float *in0, *in1, *w, *bias;
  float *out;
  unsigned in0_bytes, in1_bytes, w_bytes, out_bytes;
  int C0, C1;                      // channels per source (multiples of 32; C1 = 0: one source)
  int s0n, s0h, s0w, s1n, s1h, s1w;
  int on, oh, ow;                  // output strides in GEMM-grid pixels (channel stride 1)
  int H, W, OH, OW, Cout, Kpad, KH, KW, stride, pad, relu, M;
  int convT, w_phase_stride, dst_sh, dst_sw;
  int flush;                       // chunks between accumulator flushes (0: never)
  float *out2;                     // OUTP launches: dense fp32 twin of the output at the same element offsets, or null
  unsigned out2_bytes;
};

// one LDS-DMA: lane l's 16 bytes at rsrc.base + voff + soff land at lds_addr + 16 l; zeros when voff is out of range
// (the range check looks at voff alone: tools/probes/lds_dma_probe.hip).  M0 carries the wave-uniform LDS address; it
// is written here and never restored -- nothing else in this kernel uses M0 (no other m0 in the generated code).
__device__ __forceinline__ void dma16(const unsigned lds_addr, const unsigned voff, const i32x4 rsrc, const unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// x / d for 0 <= x < 2^24, 1 <= d < 2^24
__device__ __forceinline__ int udiv_small(const int x, const int d) {
  int q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) ++q;
  return q;
}

__device__ __forceinline__ i32x4 make_rsrc(const void *ptr, const unsigned bytes) {
  const unsigned long long b = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// phase timestamps of the instrumented variant (ISI_CONV_ABLATE=32; tools/ablate_conv.py reads them back)
__device__ long long g_conv_pair_stamps[256];

template <int BN, bool OUTP, int ABL = 0, bool TWIN = false>   // ABL: measurements only (ISI_CONV_ABLATE): 1 no MFMAs, 2 no DMA in the loop
__global__ __launch_bounds__(512, 2) void conv_pair_kernel(const PairK p) {   // TWIN: also an fp32 copy of a pair output (training)
  constexpr int TM = 2;                       // 32-row tiles per wave (4 waves along M)
  constexpr int TN = BN / 64;                 // 32-column tiles per wave (2 waves along N)
  constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB, STAGE = A_STAGE + B_STAGE;
  constexpr int NA = A_STAGE / 1024 / 8;      // A DMAs per wave and chunk (4)
  constexpr int NB = B_STAGE / 1024 / 8;      // B DMAs per wave and chunk (2 / 1)
  constexpr int PER = NA + NB;                // DMAs per wave and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *row_oo = reinterpret_cast<int *>(smem + NS * STAGE);   // [BM] output element offset or -1
  int *row_n = row_oo + BM;                                    // [BM] batch index or -1
  int *row_y = row_n + BM;                                     // [BM] top-left input y
  int *row_x = row_y + BM;                                     // [BM] top-left input x
  float *col_bias = reinterpret_cast<float *>(row_x + BM);     // [BN] bias of the tile's output channels

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- persistent over work items (phase, N tile, M tile): a workgroup's output stores drain while it already
  // stages / multiplies its next tile (a non-persistent workgroup holds its CU until its stores have been
  // acknowledged: with one workgroup per CU that exposed ~35 us of HBM write time per launch, tools/ablate_conv.py)
  long long t_entry = 0;
  if constexpr (ABL & 32) t_entry = __builtin_readcyclecounter();
  const int mtiles = (p.M + BM - 1) / BM, ntiles_n = p.Cout / BN;
  const int nitems = mtiles * ntiles_n * (p.convT ? 4 : 1);
  for (int item_i = blockIdx.x; item_i < nitems; item_i += gridDim.x) {
  // XCD-aware order: item_i's low bits (the XCD a workgroup runs on) select a contiguous range of items, so that
  // consecutive M tiles (shared halo rows) meet in one XCD's L2
  int item;
  {
    const int q = nitems / 8, r = nitems % 8, xcd = item_i % 8, idx = item_i / 8;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int phase = item / (mtiles * ntiles_n);
  const int rem_i = item - phase * (mtiles * ntiles_n);
  const int m0 = (rem_i % mtiles) * BM;
  const int n0 = (rem_i / mtiles) * BN;
  const int py = p.convT ? (phase >> 1) : 0;
  const int px = p.convT ? (phase & 1) : 0;
  const int pad_y = p.convT ? 1 - py : p.pad;
  const int pad_x = p.convT ? 1 - px : p.pad;
  const int w_off = p.convT ? phase * p.w_phase_stride : 0;
  const int out_off = p.convT ? py * p.dst_sh + px * p.dst_sw : 0;

  long long e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
  if constexpr (ABL & 32) e0 = __builtin_readcyclecounter();
  __syncthreads();   // the previous item's epilogue has finished with the LDS (transpose buffers, row tables)
  if constexpr (ABL & 32) e1 = __builtin_readcyclecounter();
  if (tid < BM) {
    const int m = m0 + tid;
    int b = -1, oy = 0, ox = 0;
    if (m < p.M) {
      // m / (OH OW) and the remainder's / OW: reciprocal estimate + one correction step (exact: M < 2^24 here, the
      // estimate is off by at most one); the two integer divisions were 40 % of a tile's 6 k-cycle set-up
      b = udiv_small(m, p.OH * p.OW);
      const int rem = m - b * (p.OH * p.OW);
      oy = udiv_small(rem, p.OW);
      ox = rem - oy * p.OW;
    }
    row_n[tid] = b;
    row_y[tid] = oy * p.stride - pad_y;
    row_x[tid] = ox * p.stride - pad_x;
    row_oo[tid] = b < 0 ? -1 : out_off + b * p.on + oy * p.oh + ox * p.ow;
  } else if (tid < BM + BN) {
    col_bias[tid - BM] = p.bias ? p.bias[n0 + tid - BM] : 0.f;
  }
  __syncthreads();

  // Descriptors start `margin` bytes (the padding's worth) BEFORE each source: a lane's voffset is then the
  // non-negative offset of its row's top-left tap incl. padding, constant over the K walk, and the chunk's tap /
  // channel offset rides in the SGPR offset.  (readfirstlane: these are uniform, and must be SGPRs for the asm.)
  const int upy = __builtin_amdgcn_readfirstlane(pad_y), upx = __builtin_amdgcn_readfirstlane(pad_x);
  const unsigned margin0 = (unsigned)(upy * p.s0h + upx * p.s0w) * 4u;
  const unsigned margin1 = (unsigned)(upy * p.s1h + upx * p.s1w) * 4u;
  const i32x4 rs0 = make_rsrc(reinterpret_cast<const char *>(p.in0) - margin0, p.in0_bytes + margin0);
  const i32x4 rs1 = make_rsrc(reinterpret_cast<const char *>(p.in1) - margin1, p.in1_bytes + margin1);
  const i32x4 rsw = make_rsrc(p.w, p.w_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  // ---- this lane's share of the staging: A DMA i of wave w covers tile rows (w NA + i) 8 .. + 8, lane l -> row
  // + (l >> 3), LDS piece position l & 7, i.e. SOURCE piece (l & 7) ^ ((row >> 1) & 7).  Per DMA: the voffset into
  // the current source (a_v; a_v1 holds the second source's until the walk reaches it) and a bit mask of the taps
  // that fall inside the image (zero padding = out-of-range voffset).
  unsigned a_v[NA], a_v1[NA], a_mask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave * NA + i) * 8 + (lane >> 3);
    const int b = row_n[r];
    const int y0 = row_y[r], x0 = row_x[r];
    const unsigned piece = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
    a_v[i] = (unsigned)(b * p.s0n + (y0 + upy) * p.s0h + (x0 + upx) * p.s0w) * 4u + piece;
    a_v1[i] = (unsigned)(b * p.s1n + (y0 + upy) * p.s1h + (x0 + upx) * p.s1w) * 4u + piece;
    // taps inside the image: kh in [max(0, -y0), min(KH, H - y0)), kw likewise; bit kh KW + kw (KH, KW <= 4)
    const int kw_lo = max(0, -x0), kw_hi = min(p.KW, p.W - x0);
    const unsigned colbits = kw_hi > kw_lo ? ((1u << (kw_hi - kw_lo)) - 1u) << kw_lo : 0u;
    const int kh_lo = max(0, -y0), kh_hi = min(p.KH, p.H - y0);
    unsigned mask = 0;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
      if (kh >= kh_lo && kh < kh_hi) mask |= colbits << (kh * p.KW);
    a_mask[i] = b >= 0 ? mask : 0u;
  }
  unsigned b_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave * NB + j) * 8 + (lane >> 3);
    b_off[j] = (unsigned)(w_off + (n0 + n) * p.Kpad) * 4u + (unsigned)(((lane & 7) ^ ((n >> 1) & 7)) * 16);
  }

  // K walk, slice-major (for each 32-channel slice all KH x KW taps back to back: the shifted re-reads of a slice are
  // adjacent in time on every workgroup of the XCD and hit its L2): state of the NEXT chunk to issue, all uniform
  const int Cin = p.C0 + p.C1;
  const int nk = p.KH * p.KW * (Cin / 32);
  int is_c = 0, is_kh = 0, is_kw = 0;            // channel offset (over both sources), tap
  i32x4 rs_cur = rs0;                            // source the walk is in
  int sh_cur = p.s0h, sw_cur = p.s0w, c_sub = 0;

  // piece Q (0 .. PER - 1) of the next chunk -> `stage`; the last piece also advances the walk
#define ISI_ISSUE_PIECE(stage_, Q)                                                                                      \
  do {                                                                                                                  \
    if ((Q) < NA) {                                                                                                     \
      const unsigned soff_ = (unsigned)(is_kh * sh_cur + is_kw * sw_cur + (is_c - c_sub)) * 4u;                          \
      const unsigned tapbit_ = 1u << (is_kh * p.KW + is_kw);                                                             \
      const unsigned v_ = (a_mask[(Q) < NA ? (Q) : 0] & tapbit_) ? a_v[(Q) < NA ? (Q) : 0] : OOB;                         \
      dma16(lds0 + (unsigned)((stage_) * STAGE + (wave * NA + (Q)) * 1024), v_, rs_cur, soff_);                          \
    } else {                                                                                                            \
      const unsigned koff_ = (unsigned)((is_kh * p.KW + is_kw) * Cin + is_c) * 4u;                                        \
      dma16(lds0 + (unsigned)((stage_) * STAGE + A_STAGE + (wave * NB + ((Q) - NA)) * 1024),                             \
            b_off[(Q) >= NA ? (Q) - NA : 0], rsw, koff_);                                                                \
    }                                                                                                                   \
    if ((Q) == PER - 1) {                                                                                               \
      if (++is_kw == p.KW) {                                                                                            \
        is_kw = 0;                                                                                                      \
        if (++is_kh == p.KH) {                                                                                          \
          is_kh = 0;                                                                                                    \
          is_c += 32;                                                                                                   \
          if (is_c == p.C0 && p.C1 > 0) {   /* the walk enters the second source */                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) a_v[i_] = a_v1[i_];                                        \
            rs_cur = rs1; sh_cur = p.s1h; sw_cur = p.s1w; c_sub = p.C0;                                                  \
          }                                                                                                             \
        }                                                                                                               \
      }                                                                                                                 \
    }                                                                                                                   \
  } while (0)
#define ISI_ISSUE_CHUNK(stage_)                                                                                         \
  do {                                                                                                                  \
    ISI_ISSUE_PIECE(stage_, 0); ISI_ISSUE_PIECE(stage_, 1); ISI_ISSUE_PIECE(stage_, 2); ISI_ISSUE_PIECE(stage_, 3);     \
    ISI_ISSUE_PIECE(stage_, 4);                                                                                         \
    if (PER > 5) ISI_ISSUE_PIECE(stage_, 5);                                                                            \
  } while (0)

  // ---- fragment addresses: tile row r = base + (lane & 31), k-block kb = lane >> 5; piece (4 s + 2 kb + pl) of the
  // row sits at slot (piece ^ ((r >> 1) & 7)); base is a multiple of 32, so the swizzle depends on the lane only
  const int frow = lane & 31, kb = lane >> 5;
  const int xs = (2 * kb) ^ ((frow >> 1) & 7);
  unsigned fa[2][2], fb[2][2];                   // [k-step][plane] byte offsets inside a stage
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      fa[s][pl] = (unsigned)((wm * 64 + frow) * ROWB + ((xs ^ (4 * s + pl)) << 4));
      fb[s][pl] = (unsigned)(A_STAGE + (wn * (BN / 2) + frow) * ROWB + ((xs ^ (4 * s + pl)) << 4));
    }

  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  bool fresh = true;   // uniform: the next MFMA of every accumulator starts a new partial sum (C operand = 0)

  constexpr int WCOLS = TN * 32;                 // columns of this wave's sub-tile
  if constexpr (ABL & 32) e2 = __builtin_readcyclecounter();
  // ---- prologue: two chunks in flight; chunk 0 landed (every wave waits for its own pieces, then the barrier)
  ISI_ISSUE_CHUNK(0);
  if (nk > 1) {
    ISI_ISSUE_CHUNK(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- main loop.  A chunk is handled in two barrier-separated phases per wave:
  //   memory phase  fragments of chunk kc: LDS -> registers (16 ds_read_b128); DMAs of chunk kc + 2 issued; waits
  //   matrix phase  the chunk's 24 (12) MFMAs out of registers, nothing else
  // and the two waves of a SIMD (waves w and w + 4 of the workgroup) run HALF AN ITERATION APART: waves 4-7 pass one
  // extra barrier first, so that while one wave of a SIMD occupies the matrix pipe the other one issues its DMAs (which
  // queue behind the CU's 64 B/clk address path: 16 clocks per 1-KiB piece, 48 pieces per chunk) and fragment reads.
  // In lockstep -- every wave issuing DMAs at the same time, then every wave multiplying -- the three cost terms
  // added up (tools/ablate_conv.py: 102 us skeleton + 54 us DMA + 106 us matrix = the measured 265 us).
  // Hand-off rules (T = barrier interval; group A = waves 0-3, group B = waves 4-7 one interval later):
  //   * a wave ends its memory phase with lgkmcnt(0) (its reads of the stage are done: the stage may be overwritten
  //     two intervals later) and vmcnt(PER) (everything but the pieces it has just issued has landed);
  //   * chunk c's pieces are issued during the memory phases of chunk c - 2 and are complete one memory phase
  //     later on both groups, i.e. before the barrier in front of the first memory phase that reads them.
  if constexpr (ABL & 32) e3 = __builtin_readcyclecounter();
  const bool group_b = wave >= 4;                 // uniform per wave
  if (group_b) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  int since_flush = 0;
  for (int kc = 0; kc < ((ABL & 4) ? 1 : nk); ++kc) {
    const int stage = kc % NS;
    const bool more = kc + 2 < nk;               // uniform
    const int nstage = (kc + 2) % NS;
    // ---------------- memory phase
    const bool stamp = (ABL & 32) && blockIdx.x == 8 && (wave & 3) == 0 && kc >= 8 && kc < 16 && lane == 0;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    if constexpr (ABL & 32) t0 = __builtin_readcyclecounter();
    const char *st = smem + stage * STAGE;
    s16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[s_][i] = *reinterpret_cast<const s16x8 *>(st + fa[s_][0] + i * 32 * ROWB);
        al[s_][i] = *reinterpret_cast<const s16x8 *>(st + fa[s_][1] + i * 32 * ROWB);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[s_][j] = *reinterpret_cast<const s16x8 *>(st + fb[s_][0] + j * 32 * ROWB);
        bl[s_][j] = *reinterpret_cast<const s16x8 *>(st + fb[s_][1] + j * 32 * ROWB);
      }
    }
    if constexpr (ABL & 32) t1 = __builtin_readcyclecounter();
    if (more && !(ABL & 2)) ISI_ISSUE_CHUNK(nstage);
    if constexpr (ABL & 32) t2 = __builtin_readcyclecounter();
    // the fragments are in registers (nobody reads this stage on this wave's behalf any more), and all but the newest
    // pieces of this wave have landed
    if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (measurement: DMA never waited for)
    else if (more && !(ABL & 2)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 32) t3 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (ABL & 32) t4 = __builtin_readcyclecounter();
    // ---------------- matrix phase: lo terms first, the dominant hi.hi last (order of conv_igemm_f32.hip)
    // The partner wave on this SIMD is in its memory phase: its address arithmetic, DMA and LDS issues would take
    // issue slots between this wave's MFMAs -- the matrix pipe's owner gets priority for the phase.
    __builtin_amdgcn_s_setprio(2);
    // first term of k-step 0: starts a new partial sum after a flush (zero C operand: no register clearing)
    if (fresh) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if constexpr (!(ABL & 1))
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[0][j]),
                                                               __builtin_bit_cast(f16x8, al[0][i]), zero, 0, 0, 0);
        }
      fresh = false;
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if constexpr (!(ABL & 1))
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[0][j]),
                                                               __builtin_bit_cast(f16x8, al[0][i]), acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int t = (s_ == 0 ? 1 : 0); t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const s16x8 av = t == 0 ? al[s_][i] : ah[s_][i];
            const s16x8 bv = t == 1 ? bl[s_][j] : bh[s_][j];
            if constexpr (!(ABL & 1))
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv), __builtin_bit_cast(f16x8, av),
                                                                 acc[i][j], 0, 0, 0);
            else asm volatile("" ::"v"(av), "v"(bv));   // keep the fragment reads alive
          }
    __builtin_amdgcn_s_setprio(0);
    if (p.flush && ++since_flush == p.flush) {
      since_flush = 0;
      fresh = true;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[i][j][r] += acc[i][j][r];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 32) {
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][TN - 1][15]));   // the stamp below follows the last MFMA's result
      t5 = __builtin_readcyclecounter();
    }
    if (!(group_b && kc == ((ABL & 4) ? 1 : nk) - 1)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if constexpr (ABL & 32) {
      if (stamp) {
        long long *d = g_conv_pair_stamps + ((wave >> 2) * 8 + (kc - 8)) * 8;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = t5; d[6] = __builtin_readcyclecounter();
      }
    }
  }

  // ---- epilogue, straight from the accumulators.  The MFMAs ran with SWAPPED operands (weights = rows, pixels =
  // columns; the same products in the same k order, so the same bits): lane (pixel = lane & 31, kb = lane >> 5) holds,
  // per 32-channel tile j, the channels (r & 3) + 8 (r >> 2) + 4 kb of ITS pixel -- per quad q = r >> 2 four
  // consecutive channels.  fp32 output: one 16-byte store per quad.  Pair output: the quad's hi and lo pieces are
  // exchanged between the two half-waves with v_permlane32_swap (tools/probes/permlane_probe.hip) so that the lower
  // lane holds the group's 8 hi pieces and the upper lane its 8 lo pieces: again one 16-byte store per quad.  No LDS,
  // no barrier: the stage ring is not touched.
  if constexpr (ABL & 32) { e4 = __builtin_readcyclecounter(); e5 = e4; }
  const __amdgpu_buffer_rsrc_t rso_b = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso2_b = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, TWIN ? p.out2_bytes : 0u, 0x00020000);
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int o = row_oo[wm * 64 + i * 32 + frow];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cg = wn * WCOLS + j * 32 + 8 * q;                     // first channel of the 8-group (tile-relative)
        const float4 bq = *reinterpret_cast<const float4 *>(col_bias + cg + 4 * kb);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          float t = (fresh ? tot[i][j][r] : tot[i][j][r] + acc[i][j][r]) * f16s::kUnscale;
          t += e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
          // NaN-propagating rectifier: an operand beyond the f16 range (or an fp32 overflow) stays loud (torch.relu)
          if (p.relu) t = t < 0.f ? 0.f : t;   // NaN < 0 is false: a NaN stays
          v[e] = t;
        }
        uint4 w;
        unsigned off;
        if constexpr (OUTP) {
          uint2 hi, lo;
          f16s::split4(make_float4(v[0], v[1], v[2], v[3]), f16s::kScaleA, hi, lo);
          const u32x2v sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
          const u32x2v sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
          w = make_uint4(sx.x, sy.x, sx.y, sy.y);
          off = o >= 0 ? (unsigned)(o + n0 + cg) * 4u + (unsigned)kb * 16u : OOB_STORE;
          if constexpr (TWIN) {   // training: the tape keeps the fp32 values next to the pairs the next layer reads
            const uint4 wf = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                        __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, wf), rso2_b,
                                                   o >= 0 ? (unsigned)(o + n0 + cg + 4 * kb) * 4u : OOB_STORE, 0, 0);
          }
        } else {
          w = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                         __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
          off = o >= 0 ? (unsigned)(o + n0 + cg + 4 * kb) * 4u : OOB_STORE;
        }
        if constexpr (ABL & 8) { asm volatile("" ::"v"(w.x ^ w.y ^ w.z ^ w.w), "v"(off)); continue; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w), rso_b, off, 0, 0);
      }
    }
  }
  if constexpr (ABL & 32) e6 = __builtin_readcyclecounter();
  if constexpr (ABL & 32) {
    e7 = __builtin_readcyclecounter();
    if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && item_i == (int)blockIdx.x + (int)gridDim.x) {
      long long *d = g_conv_pair_stamps + 130 + (wave >> 2) * 10;
      d[0] = e0; d[1] = e1; d[2] = e2; d[3] = e3; d[4] = e4; d[5] = e5; d[6] = e6; d[7] = e7;
    }
  }
  }   // work items
  if constexpr (ABL & 32) {
    if (blockIdx.x == 8 && tid == 0) {
      g_conv_pair_stamps[126] = t_entry;
      g_conv_pair_stamps[127] = __builtin_readcyclecounter();
    }
  }
#undef ISI_ISSUE_CHUNK
#undef ISI_ISSUE_PIECE
}

// ---- the same convolution on a 128-pixel tile with FOUR waves, TWO workgroups per CU (round 3).  The 256-row kernel
// above keeps one workgroup per CU: its per-item overhead -- entry barrier, row tables, prologue, epilogue: ~15 % of an
// item (tools/stamps_conv.py) -- is serial, and a launch with one item per CU (the top-resolution layers) never overlaps
// anything.  Here a workgroup is 128 x BN with one wave per SIMD (each wave 64 x BN/2 as above), a two-stage ring
// (64 KiB at BN = 128): two workgroups share a CU with independent barriers, so one's epilogue / set-up / memory phases
// run beside the other's MFMAs without any stagger logic inside the kernel.  Per chunk:
//   wait own pieces of chunk k, barrier | fragments -> registers, lgkmcnt(0), barrier (stage free) | DMAs of chunk k + 2
//   | 24 (12) MFMAs
// Same products in the same order as the 256-row kernel (k walk, term order, flush): the same bits.
template <int BN, bool OUTP, bool TWIN = false>
__global__ __launch_bounds__(256, 2) void conv_pair128_kernel(const PairK p) {
  constexpr int BM2 = 128, NS2 = 2, NWV = 4;
  constexpr int TM = 2;                       // 32-row tiles per wave (2 waves along M)
  constexpr int TN = BN / 64;                 // 32-column tiles per wave (2 waves along N)
  constexpr int A_STAGE = BM2 * ROWB, B_STAGE = BN * ROWB, STAGE = A_STAGE + B_STAGE;
  constexpr int NA = A_STAGE / 1024 / NWV;    // A DMAs per wave and chunk (4)
  constexpr int NB = B_STAGE / 1024 / NWV;    // B DMAs per wave and chunk (4 / 2)
  constexpr int PER = NA + NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *row_oo = reinterpret_cast<int *>(smem + NS2 * STAGE);   // [BM2] output element offset or -1
  int *row_n = row_oo + BM2;                                    // [BM2] batch index or -1
  int *row_y = row_n + BM2;                                     // [BM2] top-left input y
  int *row_x = row_y + BM2;                                     // [BM2] top-left input x
  float *col_bias = reinterpret_cast<float *>(row_x + BM2);     // [BN]

  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int mtiles = (p.M + BM2 - 1) / BM2, ntiles_n = p.Cout / BN;
  const int nitems = mtiles * ntiles_n;
  for (int item_i = blockIdx.x; item_i < nitems; item_i += gridDim.x) {
  int item;
  {
    const int q = nitems / 8, r = nitems % 8, xcd = item_i % 8, idx = item_i / 8;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (item % mtiles) * BM2;
  const int n0 = (item / mtiles) * BN;
  int lane = lane0;
  asm volatile("" : "+v"(lane));               // (per-item opaque: keeps lane-derived offsets from being hoisted into registers)
  __syncthreads();   // the previous item is done with the LDS (row tables, ring)
  if (tid < BM2) {
    const int m = m0 + tid;
    int b = -1, oy = 0, ox = 0;
    if (m < p.M) {
      b = udiv_small(m, p.OH * p.OW);
      const int rem = m - b * (p.OH * p.OW);
      oy = udiv_small(rem, p.OW);
      ox = rem - oy * p.OW;
    }
    row_n[tid] = b;
    row_y[tid] = oy * p.stride - p.pad;
    row_x[tid] = ox * p.stride - p.pad;
    row_oo[tid] = b < 0 ? -1 : b * p.on + oy * p.oh + ox * p.ow;
  } else if (tid < BM2 + BN) {
    col_bias[tid - BM2] = p.bias ? p.bias[n0 + tid - BM2] : 0.f;
  }
  __syncthreads();

  const unsigned margin0 = (unsigned)(p.pad * p.s0h + p.pad * p.s0w) * 4u;
  const unsigned margin1 = (unsigned)(p.pad * p.s1h + p.pad * p.s1w) * 4u;
  const i32x4 rs0 = make_rsrc(reinterpret_cast<const char *>(p.in0) - margin0, p.in0_bytes + margin0);
  const i32x4 rs1 = make_rsrc(reinterpret_cast<const char *>(p.in1) - margin1, p.in1_bytes + margin1);
  const i32x4 rsw = make_rsrc(p.w, p.w_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  unsigned a_v[NA], a_v1[NA], a_mask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave * NA + i) * 8 + (lane >> 3);
    const int b = row_n[r];
    const int y0 = row_y[r], x0 = row_x[r];
    const unsigned piece = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
    a_v[i] = (unsigned)(b * p.s0n + (y0 + p.pad) * p.s0h + (x0 + p.pad) * p.s0w) * 4u + piece;
    a_v1[i] = (unsigned)(b * p.s1n + (y0 + p.pad) * p.s1h + (x0 + p.pad) * p.s1w) * 4u + piece;
    const int kw_lo = max(0, -x0), kw_hi = min(p.KW, p.W - x0);
    const unsigned colbits = kw_hi > kw_lo ? ((1u << (kw_hi - kw_lo)) - 1u) << kw_lo : 0u;
    const int kh_lo = max(0, -y0), kh_hi = min(p.KH, p.H - y0);
    unsigned mask = 0;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
      if (kh >= kh_lo && kh < kh_hi) mask |= colbits << (kh * p.KW);
    a_mask[i] = b >= 0 ? mask : 0u;
  }
  unsigned b_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave * NB + j) * 8 + (lane >> 3);
    b_off[j] = (unsigned)((n0 + n) * p.Kpad) * 4u + (unsigned)(((lane & 7) ^ ((n >> 1) & 7)) * 16);
  }
  const int Cin = p.C0 + p.C1;
  const int nk = p.KH * p.KW * (Cin / 32);
  int is_c = 0, is_kh = 0, is_kw = 0;
  i32x4 rs_cur = rs0;
  int sh_cur = p.s0h, sw_cur = p.s0w, c_sub = 0;
  auto issue_chunk = [&](const int stage_) {
    const unsigned soff_ = (unsigned)(is_kh * sh_cur + is_kw * sw_cur + (is_c - c_sub)) * 4u;
    const unsigned tapbit_ = 1u << (is_kh * p.KW + is_kw);
#pragma unroll
    for (int q = 0; q < NA; ++q)
      dma16(lds0 + (unsigned)(stage_ * STAGE + (wave * NA + q) * 1024), (a_mask[q] & tapbit_) ? a_v[q] : OOB, rs_cur, soff_);
    const unsigned koff_ = (unsigned)((is_kh * p.KW + is_kw) * Cin + is_c) * 4u;
#pragma unroll
    for (int q = 0; q < NB; ++q)
      dma16(lds0 + (unsigned)(stage_ * STAGE + A_STAGE + (wave * NB + q) * 1024), b_off[q], rsw, koff_);
    if (++is_kw == p.KW) {
      is_kw = 0;
      if (++is_kh == p.KH) {
        is_kh = 0;
        is_c += 32;
        if (is_c == p.C0 && p.C1 > 0) {   // the walk enters the second source
#pragma unroll
          for (int i_ = 0; i_ < NA; ++i_) a_v[i_] = a_v1[i_];
          rs_cur = rs1; sh_cur = p.s1h; sw_cur = p.s1w; c_sub = p.C0;
        }
      }
    }
  };

  const int frow = lane & 31, kb = lane >> 5;
  const int xs = (2 * kb) ^ ((frow >> 1) & 7);
  unsigned fa[2][2], fb[2][2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      fa[s_][pl] = (unsigned)((wm * 64 + frow) * ROWB + ((xs ^ (4 * s_ + pl)) << 4));
      fb[s_][pl] = (unsigned)(A_STAGE + (wn * (BN / 2) + frow) * ROWB + ((xs ^ (4 * s_ + pl)) << 4));
    }
  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  bool fresh = true;
  constexpr int WCOLS = TN * 32;

  // ---- prologue: chunks 0 and 1 in flight
  issue_chunk(0);
  if (nk > 1) issue_chunk(1);
  int since_flush = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const int stage = kc & 1;
    // chunk kc has landed: this wave's pieces (those of chunk kc + 1 may stay in flight), then everyone's
    if (kc + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *st = smem + stage * STAGE;
    s16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[s_][i] = *reinterpret_cast<const s16x8 *>(st + fa[s_][0] + i * 32 * ROWB);
        al[s_][i] = *reinterpret_cast<const s16x8 *>(st + fa[s_][1] + i * 32 * ROWB);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[s_][j] = *reinterpret_cast<const s16x8 *>(st + fb[s_][0] + j * 32 * ROWB);
        bl[s_][j] = *reinterpret_cast<const s16x8 *>(st + fb[s_][1] + j * 32 * ROWB);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (kc + 2 < nk) {
      // the stage is free once every wave holds its fragments: chunk kc + 2 goes there
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_chunk(stage);
    }
    __builtin_amdgcn_s_setprio(2);
    if (fresh) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[0][j]), __builtin_bit_cast(f16x8, al[0][i]), zero, 0, 0, 0);
        }
      fresh = false;
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[0][j]), __builtin_bit_cast(f16x8, al[0][i]), acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int t = (s_ == 0 ? 1 : 0); t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const s16x8 av = t == 0 ? al[s_][i] : ah[s_][i];
            const s16x8 bv = t == 1 ? bl[s_][j] : bh[s_][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv), __builtin_bit_cast(f16x8, av), acc[i][j], 0, 0, 0);
          }
    __builtin_amdgcn_s_setprio(0);
    if (p.flush && ++since_flush == p.flush) {
      since_flush = 0;
      fresh = true;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[i][j][r] += acc[i][j][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue, straight from the accumulators (as in the 256-row kernel)
  const __amdgpu_buffer_rsrc_t rso_b = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso2_b = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, TWIN ? p.out2_bytes : 0u, 0x00020000);
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int o = row_oo[wm * 64 + i * 32 + frow];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cg = wn * WCOLS + j * 32 + 8 * q;
        const float4 bq = *reinterpret_cast<const float4 *>(col_bias + cg + 4 * kb);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          float t = (fresh ? tot[i][j][r] : tot[i][j][r] + acc[i][j][r]) * f16s::kUnscale;
          t += e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
          if (p.relu) t = t < 0.f ? 0.f : t;
          v[e] = t;
        }
        uint4 w;
        unsigned off;
        if constexpr (OUTP) {
          uint2 hi, lo;
          f16s::split4(make_float4(v[0], v[1], v[2], v[3]), f16s::kScaleA, hi, lo);
          const u32x2v sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
          const u32x2v sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
          w = make_uint4(sx.x, sy.x, sx.y, sy.y);
          off = o >= 0 ? (unsigned)(o + n0 + cg) * 4u + (unsigned)kb * 16u : OOB_STORE;
          if constexpr (TWIN) {   // training: the tape keeps the fp32 values next to the pairs the next layer reads
            const uint4 wf = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                        __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, wf), rso2_b,
                                                   o >= 0 ? (unsigned)(o + n0 + cg + 4 * kb) * 4u : OOB_STORE, 0, 0);
          }
        } else {
          w = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                         __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
          off = o >= 0 ? (unsigned)(o + n0 + cg + 4 * kb) * 4u : OOB_STORE;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w), rso_b, off, 0, 0);
      }
    }
  }
  }   // work items
}

template <int BN, bool OUTP, bool TWIN = false>
int launch_pair128(const PairK &a, double flops, double bytes, hipStream_t stream) {
  auto kern = conv_pair128_kernel<BN, OUTP, TWIN>;
  constexpr size_t smem = (size_t)2 * (128 + BN) * ROWB + 4 * 128 * sizeof(int) + BN * sizeof(float);
  static_assert(2 * smem <= 160 * 1024, "two workgroups per CU");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_pair128)");
    attr_set.mark();
  }
  const int nitems = ((a.M + 127) / 128) * (a.Cout / BN);
  const int n_cu = current_device_cu_count();
  constexpr int per_cu = BN == 64 ? 3 : 2;            // 48-KiB / 64-KiB rings
  dim3 grid(nitems < per_cu * n_cu ? nitems : per_cu * n_cu);   // persistent over the items
  prof::Scope scope(prof::K_CONV_F16X3, flops, bytes, stream);
  ISI_PROF_LAUNCH(scope, kern, grid, dim3(256), smem, stream, a);
  return check_launch("conv_pair128_f16");
}

template <int BN>
constexpr size_t pair_smem_bytes() {
  return (size_t)NS * (BM + BN) * ROWB + 4 * BM * sizeof(int) + BN * sizeof(float);
}

template <int BN, bool OUTP, int ABL = 0, bool TWIN = false>
int launch_pair(const PairK &a, int nphase, double flops, double bytes, hipStream_t stream) {
  auto kern = conv_pair_kernel<BN, OUTP, ABL, TWIN>;
  constexpr size_t smem = pair_smem_bytes<BN>();
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_pair)");
    attr_set.mark();
  }
  const int nitems = ((a.M + BM - 1) / BM) * (a.Cout / BN) * nphase;
  const int n_cu = current_device_cu_count();
  dim3 grid(nitems < n_cu ? nitems : n_cu);   // one 144-KiB workgroup per CU, persistent over the items
  // the forward's single dominant kernel is timed under its own id (bench.py adds it back to the family's line)
  prof::Scope scope(BN == 128 && OUTP && ABL == 0 && !TWIN ? prof::K_CONV_PAIR_128_PAIROUT : prof::K_CONV_F16X3, flops, bytes, stream);
  ISI_PROF_LAUNCH(scope, kern, grid, dim3(512), smem, stream, a);
  return check_launch("conv_pair_f16");
}

}  // namespace

int conv_pair_debug_stamps(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_pair_stamps), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -2;
}

// Shapes the DMA kernel takes: pair8 sources of 32-channel multiples (both, when there are two), whole 64-column
// output tiles, channels-last output.  (conv_igemm_f32.hip runs everything else, including fp32 sources.)
bool conv_pair_kernel_ok(int C0, int C1, int Cout, int taps) {
  const bool off = knobs().no_conv_pair_kernel != 0;
  return !off && C0 > 0 && C0 % 32 == 0 && C1 % 32 == 0 && Cout % 64 == 0 && taps >= 1 && taps <= 16;   // 16-bit tap masks
}

// Which tile height for this launch (measured, tools/bench_conv_pair.py; the two kernels return the same bits)?
// B = 64: k4s2 64->128 at 64 x 256: 210 (256 rows) vs 230 us (128 rows); 3x3 128->128 at 32 x 128: 213 vs 217; 3x3 64->128 at
// 16 x 64: 35.3 vs 35.8; k4s2 128->64 at 32 x 128 (64 output columns): 68 vs 63 -- the narrow tile is the one case where the
// second workgroup per CU pays (its 48-KiB ring leaves room for three): the 128-row form runs the 64-column launches.
static bool conv_pair_prefers_128(int M, int Cout, int K) {
  (void)M; (void)K;
  return Cout % 128 != 0;
}

int conv_pair_f16(const PairConvArgs &c, hipStream_t stream) {
  // voffsets stay below the out-of-range marker 0x7FFFFFF0 with room for the SGPR offset
  if (c.in0_bytes >= 0x70000000u || (c.in1 && c.in1_bytes >= 0x70000000u))
    return unsupported("conv_pair: a pair-format source spans 1.75 GiB or more");
  PairK a;
  memset(&a, 0, sizeof a);
  a.in0 = c.in0; a.in1 = c.in1 ? c.in1 : c.in0; a.w = c.w16; a.bias = c.bias; a.out = c.out;
  a.in0_bytes = c.in0_bytes; a.in1_bytes = c.in1 ? c.in1_bytes : c.in0_bytes; a.w_bytes = c.w_bytes; a.out_bytes = c.out_bytes;
  a.C0 = c.C0; a.C1 = c.C1;
  a.s0n = c.s0n; a.s0h = c.s0h; a.s0w = c.s0w; a.s1n = c.s1n; a.s1h = c.s1h; a.s1w = c.s1w;
  a.on = c.on; a.oh = c.oh; a.ow = c.ow;
  a.H = c.H; a.W = c.W; a.OH = c.OH; a.OW = c.OW; a.Cout = c.Cout; a.Kpad = c.Kpad; a.KH = c.KH; a.KW = c.KW;
  a.stride = c.stride; a.pad = c.pad; a.relu = c.relu; a.M = c.M;
  a.convT = c.convT; a.w_phase_stride = c.w_phase_stride; a.dst_sh = c.dst_sh; a.dst_sw = c.dst_sw;
  // accumulator flush period: every 3 chunks (one kernel row of a 3x3 / three taps) unless overridden; 0 = never
  // (read per call: the tests compare flush = 0 -- the bits of conv_igemm_f32.hip -- with the default)
  a.flush = knobs().conv_flush;
  if (c.twin) {
    // the fp32 twin shares the pair tensor's element offsets: plain (non-transposed) dense channels-last outputs only
    if (!c.out_pair || c.convT) return unsupported("conv_pair: an fp32 twin accompanies a pair-format output of a plain convolution");
    a.out2 = c.twin; a.out2_bytes = c.out_bytes;
  }
  const int nphase = c.convT ? 4 : 1;
  const double K = (double)c.KH * c.KW * (c.C0 + c.C1);
  const double flops = 2.0 * c.M * nphase * c.Cout * K;
  const double in_px = nphase == 1 ? (double)c.M / (c.OH * c.OW) * c.H * c.W : (double)c.M;
  const double bytes = 4.0 * (in_px * (c.C0 + c.C1) + (double)c.M * nphase * c.Cout + nphase * c.Cout * K);
  const bool wide = c.Cout % 128 == 0;
#ifdef ISI_MEASURE   // measurements (tools/ablate_conv.py): wrong results by design, not in the default build
  const int abl = knobs().conv_ablate;
  if (abl == 1) return wide ? launch_pair<128, true, 1>(a, nphase, flops, bytes, stream) : launch_pair<64, true, 1>(a, nphase, flops, bytes, stream);
  if (abl == 2) return wide ? launch_pair<128, true, 2>(a, nphase, flops, bytes, stream) : launch_pair<64, true, 2>(a, nphase, flops, bytes, stream);
  if (abl == 3) return wide ? launch_pair<128, true, 3>(a, nphase, flops, bytes, stream) : launch_pair<64, true, 3>(a, nphase, flops, bytes, stream);
  if (abl == 7) return launch_pair<128, true, 7>(a, nphase, flops, bytes, stream);     // + no main loop
  if (abl == 15) return launch_pair<128, true, 15>(a, nphase, flops, bytes, stream);   // + no output stores
  if (abl == 11) return launch_pair<128, true, 11>(a, nphase, flops, bytes, stream);   // skeleton loop, no output stores
  if (abl == 32) return launch_pair<128, true, 32>(a, nphase, flops, bytes, stream);   // phase timestamps
  if (abl == 16) return launch_pair<128, true, 16>(a, nphase, flops, bytes, stream);   // DMA issued, never waited for
  if (abl == 17) return launch_pair<128, true, 17>(a, nphase, flops, bytes, stream);   // same, no MFMAs
  if (abl == 8) return launch_pair<128, true, 8>(a, nphase, flops, bytes, stream);     // everything but the output stores
#endif
  // tile height: 128-row workgroups (two per CU) or the 256-row kernel; ISI_CONV_PAIR_BM forces one (tests, measurements)
  const int bm = knobs().conv_pair_bm;
  const bool use128 = !c.convT && (bm == 128 || (bm == 0 && conv_pair_prefers_128(c.M, c.Cout, (int)K)));
  if (a.out2) {   // pair output + fp32 twin (training forward)
    if (use128) return wide ? launch_pair128<128, true, true>(a, flops, bytes, stream) : launch_pair128<64, true, true>(a, flops, bytes, stream);
    return wide ? launch_pair<128, true, 0, true>(a, nphase, flops, bytes, stream) : launch_pair<64, true, 0, true>(a, nphase, flops, bytes, stream);
  }
  if (use128) {
    if (c.out_pair) return wide ? launch_pair128<128, true>(a, flops, bytes, stream) : launch_pair128<64, true>(a, flops, bytes, stream);
    return wide ? launch_pair128<128, false>(a, flops, bytes, stream) : launch_pair128<64, false>(a, flops, bytes, stream);
  }
  if (c.out_pair) return wide ? launch_pair<128, true>(a, nphase, flops, bytes, stream)
                              : launch_pair<64, true>(a, nphase, flops, bytes, stream);
  return wide ? launch_pair<128, false>(a, nphase, flops, bytes, stream)
              : launch_pair<64, false>(a, nphase, flops, bytes, stream);
}

}  // namespace isi
