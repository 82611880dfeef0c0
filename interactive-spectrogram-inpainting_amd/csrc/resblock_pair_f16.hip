// Fused RosinalityResBlock of the split-f16 PAIR pipeline for gfx950.
//
//   out = [relu]( r + W2 * relu(W1 (*) r + b1) + b2 )        r = rectified input (reference
//   vqvae/encoder_decoder.py:22-35 with its in-place first ReLU already applied by the producer of r)
//
// Same arithmetic as resblock_f32.hip's ISI_CONV_F16X3 | ISI_CONV_W16 path (hi.hi + hi.lo + lo.hi of two 11-bit f16
// pieces per operand, fp32 accumulation), restructured around what bounded that kernel: it was LDS-bound (one 32x32
// tile per wave in the first GEMM: every operand fragment fed three MFMAs; 70 KB of ds_write staging per slice) and
// barrier-bound (two barriers per 54 MFMAs), at 0.24 of the matrix ceiling.  Here:
//
//   * input, hidden weights and output in the pair format (split_f16.h): staging is a plain copy by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`, counted s_waitcnt; see conv_pair_f16.hip), no conversion, no ds_write;
//   * a workgroup owns TH rows x 64 pixels, one wave per row: a wave's first GEMM is [64 px x 9 C] x [9 C x 32]
//     = TWO 32x32 tiles that share every weight fragment (6 fragment reads per 6 MFMAs instead of 8);
//   * K is walked in 16-channel stages -- the (TH + 2) x 66 halo of the slice and the [9 taps][32][16] slice of W1,
//     43.8 KB (TH = 4) / 60.7 KB (TH = 8) -- through a ring of 3 / 2 stages: the next stage's DMAs are issued
//     between the MFMAs of the current one, ONE barrier per 54 MFMAs per wave, the nine taps read shifted windows of
//     the resident halo (16-byte pieces XOR-swizzled on the DMA's source side: conflict-free ds_read_b128);
//   * both GEMMs run with SWAPPED operands (weights = MFMA rows, pixels = columns), so a lane's accumulators are
//     channels of ONE pixel: GEMM 1's accumulators (+ b1, ReLU, f16 split) are GEMM 2's B fragments as they stand, and
//     GEMM 2's (rows permuted so that a lane holds whole pair8 groups) are stored straight from the registers;
//   * the skip connection never touches memory again: the centre-tap fragments of every K stage are this lane's pieces
//     of r for exactly the groups it stores, and are kept in registers (64 of them at C = 128);
//   * W2 (as MFMA fragments) and the biases live in LDS for the whole launch: the tail contains NO vector-memory load
//     (a load waits in order behind earlier stores -- one vmcnt counts both), no LDS transpose and no barrier.
//     History (B = 64, 32 x 128, C = 128): LDS planes + transposes + per-pass W2 / skip loads 101 us; loads hoisted in
//     front of the stores 94 us (tools/stamps_resblock.py: the launch is then bound by its 440 MB of traffic, the 134 MB
//     skip re-read arriving as one burst); this form reads the input once.
//
// Requirements (else resblock_f32.hip): C % 32 == 0, C <= 128, R == 32, dense channels-last tensors.
#include <cstdlib>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_f16.h"

#ifndef ISI_RESPAIR_STORE_AUX
#define ISI_RESPAIR_STORE_AUX 0   // cache policy of the output stores.  Measured (round 6, the forward's eight launches): default 0.435 ms,
                                  // sc0 (1) 0.435, nt (2) 0.777 -- the write bursts of an item's tail live on the L2 / Infinity Cache absorbing them
#endif

namespace isi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
using f16s::f16x8;

constexpr int TW = 64, HWD = TW + 2;
constexpr int ROWB = 64;                       // bytes of a row per 16-channel stage: {hi g0, lo g0, hi g1, lo g1}
constexpr unsigned OOB = 0x7FFFFFF0u, OOB_ST = 0xFFFFFFF0u;

#ifdef ISI_MEASURE
#define ISI_RESPAIR_ABLBIT(p, b) ((p).ablate & (b))
#else
#define ISI_RESPAIR_ABLBIT(p, b) (0)  // the ablations are not compiled into the default build
#endif
struct ResPairK {
  const float *in, *w1, *b1, *w2, *b2;        // w1 / w2: blocked pair copies of the packed weights
  float *out;
  unsigned in_bytes, w1_bytes, w2_bytes;
  int C, H, W, B, relu, out_pair;
  int tiles_x, tiles_y;
  int ablate;   // measurements only (ISI_RESPAIR_ABL): 1 one stage instead of C / 16, 2 no second GEMM / epilogue, 4 no skip re-read, 8 no output stores
  // training forward (null otherwise): dense fp32 twin of a pair-format output, and the block's hidden activation
  // relu(conv3x3(r) + b1) as dense channels-last fp32 [B,H,W,32] -- what the hand-written backward reads
  float *out2, *hidden;
};

__device__ __forceinline__ void dma16(const unsigned lds_addr, const unsigned voff, const i32x4 rsrc, const unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void *ptr, const unsigned bytes) {
  const unsigned long long b = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// phase timestamps (-DISI_MEASURE builds; tools/stamps_resblock.py): workgroup 8, waves 0 and 4, its second work item
#ifdef ISI_MEASURE
__device__ long long g_respair_stamps[128];
#define ISI_STAMP(i_) do { if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && (item_i == (int)blockIdx.x + (int)gridDim.x || nitems <= (int)gridDim.x)) \
    g_respair_stamps[(wave >> 2) * 64 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_STAMP(i_) do { } while (0)
#endif
#define ISI_MH(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)

// WPR = 2 (round 6, the 4-row tiles of the top resolution: ONE item per workgroup): TWO waves per tile row, 32 pixels each --
// 8 waves per workgroup, two per SIMD; a weight fragment feeds one pixel tile instead of two (4 LDS reads per 3 matrix
// instructions; the LDS is at a sixth of its rate).  tools/stamps_resblock.py 16 64, cycles of the item: 41.9 k -> 35-37 k --
// the K loop stays at ~20 k (it was bound by the matrix pipe with one wave per SIMD already: the DMA ring hides the memory),
// set-up + prologue 6.5 -> 4 k and GEMM 2 + epilogue 13 -> 10 k are shared by twice the threads.  31 -> 28 us per launch.
template <int TH, int NT, int WPR = 1>   // NT = C / 32
__global__ __launch_bounds__(TH * WPR * 64) void resblock_pair_kernel(const ResPairK p) {
  constexpr int NW = TH * WPR;                             // waves: WPR per tile row
  constexpr int NPT = 2 / WPR;                             // 32-pixel tiles per wave
  constexpr int HPIX = (TH + 2) * HWD;                     // halo pixels
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;            // padded to whole 1-KiB DMAs (16 rows of 64 B)
  constexpr int NA = A_ROWS / 16, NWD = 9 * 32 / 16;       // DMAs per stage: halo, W1 slice
  constexpr int A_BYTES = A_ROWS * ROWB, W_BYTES = 9 * 32 * ROWB, STAGE = A_BYTES + W_BYTES + 1024;   // + a dump slot for padding pieces
  constexpr int NS = TH >= 8 ? 2 : 3;                      // ring stages
  constexpr int NDMA = NA + NWD;                           // DMAs per stage
  constexpr int PER = (NDMA + NW - 1) / NW;                // per wave (the last wave(s) may have fewer: they pad with
                                                           // out-of-range pieces so that every wave's count is PER)
  constexpr int kRingBytes = NS * STAGE;
  constexpr int C = NT * 32, NSTAGE = C / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const i32x4 rsi = make_rsrc(p.in, p.in_bytes), rsw = make_rsrc(p.w1, p.w1_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  // ---- once per workgroup: W2 as FRAGMENTS and the biases into LDS, behind the ring.  (Nothing in the tail may be a
  // vector-memory load: such a load is waited for IN ORDER behind every store issued before it -- one vmcnt counts
  // both --; the per-pass W2 / skip loads of the first version of this tail cost 11 k cycles per pass, 47 % of an item,
  // tools/stamps_resblock.py.)
  // GEMM 2 runs with swapped operands, O^T[channel][pixel] = W2 h^T: MFMA row m of output tile j is channel
  //   sigma(m) = 16 (m >> 4) + 8 ((m >> 2) & 1) + 4 ((m >> 3) & 1) + (m & 3)            (+ 32 j)
  // so that lane (pixel, kb)'s 16 accumulators -- rows (r & 3) + 8 (r >> 2) + 4 kb -- are the two pair8 groups
  // 4 j + kb and 4 j + 2 + kb of its pixel, 8 consecutive channels each: exactly the groups whose pieces of r the
  // same lane read as its centre-tap fragments in stages 2 j and 2 j + 1 of the K loop (the skip connection stays in
  // registers), and exactly what a 32-byte pair8 / fp32 store writes.  The k index (hidden channel) of k-step s,
  // slot 8 kb + e is
  //   pi(s, kb, e) = 16 s + 4 kb + e (e < 4),  16 s + 8 + 4 kb + (e - 4) (e >= 4)
  // -- the order in which GEMM 1's swapped accumulators hold a pixel's hidden channels (quads 2 s and 2 s + 1).
  // Fragment (j, s, plane) of lane l = (m = l & 31, kb = l >> 5): 16 bytes at ((j 2 + s) 2 + plane) 1024 + 16 l.
  char *w2s = smem + kRingBytes;
  {
    const unsigned short *w2g = reinterpret_cast<const unsigned short *>(p.w2);   // [C][4 groups]{hi[8] | lo[8]} of 1024 w
    unsigned short *w2o = reinterpret_cast<unsigned short *>(w2s);
    for (int i = tid; i < NT * 2 * 2 * 64 * 8; i += NW * 64) {
      const int e = i & 7, l = (i >> 3) & 63, plane = (i >> 9) & 1, s_ = (i >> 10) & 1, j = i >> 11;
      const int m = l & 31, kbl = l >> 5;
      const int n = 32 * j + 16 * (m >> 4) + 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3);
      const int k = e < 4 ? 16 * s_ + 4 * kbl + e : 16 * s_ + 8 + 4 * kbl + (e - 4);
      w2o[i] = w2g[n * 64 + (k >> 3) * 16 + plane * 8 + (k & 7)];
    }
  }
  float *b2s = reinterpret_cast<float *>(w2s + NT * 4096);      // [C] 4 b2, then [32] 4 b1
  float *b1s = b2s + C;
  for (int i = tid; i < C + 32; i += NW * 64) b2s[i] = f16s::kScaleA * (i < C ? p.b2[i] : p.b1[i - C]);   // in units of 4 x, like everything in the tail

  const __amdgpu_buffer_rsrc_t rso_b = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso2_b = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, p.out2 ? p.in_bytes : 0u, 0x00020000);
  // ---- work distribution.  (Every workgroup reaches its store-heavy tail at the same moment, so the launch alternates
  // between a K-loop phase -- 173 MB of halo reads -- and a store burst at the fabric's write rate, 256 KB per CU at
  // once: 17-21 k of an item's 63 k cycles, tools/stamps_resblock.py.  Running the odd workgroups half a tile out of
  // phase -- their first tile split into an upper half done first and a lower half done last, only the waves of those
  // rows multiplying -- was built and measured: 88 against 79 us; a half item costs ~0.7 of a whole one.  At ~4 TB/s
  // in both phases the block is within 25 % of its memory floor; what is left is traffic, i.e. fusing the stack.)
  const int nitems = p.tiles_x * p.tiles_y * p.B;
  for (int item_i = blockIdx.x; item_i < nitems; item_i += gridDim.x) {
    int item;
    {   // XCD-aware order: vertically / horizontally adjacent tiles (shared halo rows) on one XCD
      const int q = nitems / 8, r = nitems % 8, xcd = item_i % 8, idx = item_i / 8;
      item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = item / (p.tiles_x * p.tiles_y);
    const int rem = item - b * (p.tiles_x * p.tiles_y);
    const int y0 = (rem / p.tiles_x) * TH, x0 = (rem % p.tiles_x) * TW;
    // the lane index is made opaque per item: everything derived from it (dozens of LDS / global offsets that are
    // invariant across items) would otherwise be hoisted out of the item loop and held in ~100 registers for the whole
    // kernel
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    ISI_STAMP(0);
    const int frow = lane & 31, kb = lane >> 5;
    __syncthreads();   // the previous item is done with the ring
    ISI_STAMP(1);

    // ---- this lane's DMA pieces (constant over the stages but for the channel offset, which rides in soffset):
    // piece q of this wave is DMA number wave + NW q of the stage (halo DMAs first, then the W1 slice)
    unsigned dvo[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int d = wave + NW * q;                          // uniform
      const int row = (d < NA ? d : d - NA) * 16 + (lane >> 2);
      unsigned piece = (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);   // weight rows: swizzled with the row
      if (d < NA) {                                         // halo pixel `row`: swizzled with its COLUMN (below)
        const int hy = row / HWD, hx = row - hy * HWD;
        piece = (unsigned)(((lane & 3) ^ ((hx >> 2) & 3)) * 16);
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = row < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        dvo[q] = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * C) * 4u + piece : OOB;
      } else if (d < NDMA) {                                // W1 row: tap * 32 + hidden channel
        const int t = row >> 5, r = row & 31;
        dvo[q] = (unsigned)(r * 9 * C + t * C) * 4u + piece;
      } else {
        dvo[q] = OOB;                                       // padding piece (keeps every wave's DMA count at PER)
      }
    }
    auto issue = [&](const int stage, const int c16, const int q) {   // piece q of channel slice c16 -> ring stage
      const int d = wave + NW * q;                          // uniform
      const bool is_w = d >= NA;
      // halo DMAs first, then the W1 slice, then the stage's dump slot for padding pieces
      const unsigned dst = d < NDMA ? (unsigned)(d * 1024) : (unsigned)(A_BYTES + W_BYTES);   // (A_BYTES = NA KiB)
      dma16(lds0 + (unsigned)(stage * STAGE) + dst, dvo[q], is_w && d < NDMA ? rsw : rsi, d < NDMA ? (unsigned)(c16 * 64) : 0u);
    };

    // ---- fragment addresses inside a stage.  A: halo row (ry + dy) * 66 + hx, hx = 32 i + frow + dx; piece
    // (2 kb + pl) at position ^ ((hx >> 2) & 3): swizzling with the COLUMN instead of the linear row is equally
    // conflict-free and makes the per-lane part of a window's address depend on dx only -- 3 x 2 registers, (dy, i)
    // ride in the instruction's immediate offset.  B: row t * 32 + frow: lane-only swizzle term.
    const int ry = wave / WPR;                              // this wave's tile row
    const int px0 = WPR == 2 ? (wave & 1) : 0;              // ... and its first 32-pixel tile of the row
    unsigned abase[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = frow + dx;
      abase[dx][0] = (unsigned)((ry * HWD + hx) * ROWB + (((2 * kb) ^ ((hx >> 2) & 3)) << 4));
      abase[dx][1] = abase[dx][0] ^ 16u;
    }
    const unsigned bbase = (unsigned)(A_BYTES + frow * ROWB + (((2 * kb) ^ ((frow >> 2) & 3)) << 4));
    const unsigned bbase1 = bbase ^ 16u;

    // GEMM 1, operands swapped (W1 = MFMA rows, pixels = columns): H^T[hidden][pixel]
    f32x16 acc[NPT][2];   // [pixel tile][chain]: two chains per tile (alternating taps) keep dependent MFMAs apart
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][ch][r] = 0.f;
    // the skip connection: the centre-tap fragments of every stage ARE this lane's pieces of r -- pixel 32 i + frow,
    // channels 16 c + 8 kb .. + 7 as {hi | lo} f16 of 4 r -- and are simply kept (8 NT x 8 registers) instead of
    // re-read from memory in the tail (134 MB per launch at B = 64, which the launch waited for in a burst)
    s16x8 skh[NSTAGE][NPT], skl[NSTAGE][NPT];

    ISI_STAMP(2);
    // ---- prologue: NS - 1 stages in flight
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < NSTAGE) {
#pragma unroll
        for (int q = 0; q < PER; ++q) issue(s_, s_, q);
      }

    ISI_STAMP(3);
#pragma unroll
    for (int c = 0; c < NSTAGE; ++c) {
      constexpr int dummy = 0; (void)dummy;
      const int stage = c % NS;
      // this wave's pieces of stage c have landed (those of the NS - 2 newer stages may stay in flight) ...
      if (NS == 3 && c + 1 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and everyone's; everyone has also finished with the stage that slice c + NS - 1 goes to
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool more = c + NS - 1 < NSTAGE;
      const int nst = (c + NS - 1) % NS;
      const char *st = smem + stage * STAGE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        s16x8 ah[NPT], al[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          ah[i] = *reinterpret_cast<const s16x8 *>(st + abase[dx][0] + (dy * HWD + 32 * (px0 + i)) * ROWB);
          al[i] = *reinterpret_cast<const s16x8 *>(st + abase[dx][1] + (dy * HWD + 32 * (px0 + i)) * ROWB);
        }
        if (t == 4) {
#pragma unroll
          for (int i = 0; i < NPT; ++i) { skh[c][i] = ah[i]; skl[c][i] = al[i]; }
        }
        const s16x8 bh = *reinterpret_cast<const s16x8 *>(st + bbase + t * 32 * ROWB);
        const s16x8 bl = *reinterpret_cast<const s16x8 *>(st + bbase1 + t * 32 * ROWB);
        // lo terms first, hi.hi last (the order of the other split-f16 kernels)
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i][t & 1] = ISI_MH(bh, al[i], acc[i][t & 1]);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i][t & 1] = ISI_MH(bl, ah[i], acc[i][t & 1]);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i][t & 1] = ISI_MH(bh, ah[i], acc[i][t & 1]);
        // one or two DMAs of the slice NS - 1 ahead behind each tap's MFMAs
        if (more) {
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if (q * 9 / PER == t) issue(nst, c + NS - 1, q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ISI_STAMP(4);

    // ---- hidden activations: lane (pixel = frow of tile i, kb) holds hidden channels 8 q + 4 kb + (r & 3), q = r >> 2.
    // h = relu(acc + b1), split into the f16 pieces of 4 h: quads 2 s and 2 s + 1 are GEMM 2's B fragment of k-step s.
    s16x8 hh[NPT][2], hl[NPT][2];   // [pixel tile][k-step]
    {
      float4 b1q[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b1q[q] = *reinterpret_cast<const float4 *>(b1s + 8 * q + 4 * kb);
#pragma unroll
      for (int i = 0; i < NPT; ++i)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          float hv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = 4 * (2 * s_ + (e >> 2)) + (e & 3);
            const float4 bq = b1q[2 * s_ + (e >> 2)];
            const float bv = (e & 3) == 0 ? bq.x : (e & 3) == 1 ? bq.y : (e & 3) == 2 ? bq.z : bq.w;
            const float hpre = __builtin_fmaf(acc[i][0][r] + acc[i][1][r], f16s::kUnscale * f16s::kScaleA, bv);   // 4 h (b1 staged as 4 b1)
            hv[e] = hpre < 0.f ? 0.f : hpre;            // NaN-propagating rectifier
          }
          uint4 ph, pl;
          f16s::split2_scaled(hv[0], hv[1], ph.x, pl.x);
          f16s::split2_scaled(hv[2], hv[3], ph.y, pl.y);
          f16s::split2_scaled(hv[4], hv[5], ph.z, pl.z);
          f16s::split2_scaled(hv[6], hv[7], ph.w, pl.w);
          hh[i][s_] = __builtin_bit_cast(s16x8, ph);
          hl[i][s_] = __builtin_bit_cast(s16x8, pl);
        }
    }
    if (p.hidden) {   // (uniform) lane (pixel, kb) holds 4 h of hidden channels 16 s + 8 (e >> 2) + 4 kb + (e & 3)
      const __amdgpu_buffer_rsrc_t rsh_b = __builtin_amdgcn_make_buffer_rsrc(p.hidden, 0, (unsigned)((size_t)p.B * p.H * p.W * 32 * 4), 0x00020000);
      const int gyh = y0 + ry;
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int gx = x0 + 32 * (px0 + i) + frow;
        const bool okh = gyh < p.H && gx < p.W;
        const unsigned oh_ = (unsigned)(((b * p.H + gyh) * p.W + gx) * 32 + 4 * kb) * 4u;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          // the f16 pieces of 4 h hold h to 22 bits; the backward wants the fp32 value GEMM 2 effectively consumed:
          // decode hi + lo (exactly what the products saw), scaled back by 1 / 4
          const uint4 ph = __builtin_bit_cast(uint4, hh[i][s_]), pl = __builtin_bit_cast(uint4, hl[i][s_]);
          const unsigned phw[4] = {ph.x, ph.y, ph.z, ph.w}, plw[4] = {pl.x, pl.y, pl.z, pl.w};
          float hv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = ((e & 1) ? f16s::mix_sum<1>(phw[e >> 1], plw[e >> 1]) : f16s::mix_sum<0>(phw[e >> 1], plw[e >> 1])) * (1.f / f16s::kScaleA);
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const uint4 wv = make_uint4(__builtin_bit_cast(unsigned, hv[4 * hq]), __builtin_bit_cast(unsigned, hv[4 * hq + 1]),
                                        __builtin_bit_cast(unsigned, hv[4 * hq + 2]), __builtin_bit_cast(unsigned, hv[4 * hq + 3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, wv), rsh_b,
                                                   okh ? oh_ + (unsigned)((16 * s_ + 8 * hq) * 4) : OOB_ST, 0, 0);
          }
        }
      }
    }
    ISI_STAMP(5);
    if (ISI_RESPAIR_ABLBIT(p, 2)) { if (hh[0][0][0] == 123 && hl[NPT - 1][1][3] == 7) p.out[0] = 1.f; continue; }

    // ---- GEMM 2 (K = 32, W2 fragments from LDS) and the epilogue, per output tile j and pixel tile i, all in
    // registers: out = [relu](O + b2 + r), r decoded from the kept centre-tap pieces, 32-byte stores per lane.
    const int gy = y0 + ry;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s16x8 wh[2], wl[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        wh[s_] = *reinterpret_cast<const s16x8 *>(w2s + ((j * 2 + s_) * 2 + 0) * 1024 + lane * 16);
        wl[s_] = *reinterpret_cast<const s16x8 *>(w2s + ((j * 2 + s_) * 2 + 1) * 1024 + lane * 16);
      }
      float4 b2q[2][2];   // [half h][first / second quad]: channels 32 j + 16 h + 8 kb + 0 .. 7
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        b2q[h][0] = *reinterpret_cast<const float4 *>(b2s + 32 * j + 16 * h + 8 * kb);
        b2q[h][1] = *reinterpret_cast<const float4 *>(b2s + 32 * j + 16 * h + 8 * kb + 4);
      }
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        f32x16 o2;
#pragma unroll
        for (int r = 0; r < 16; ++r) o2[r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          o2 = ISI_MH(wh[s_], hl[i][s_], o2);
          o2 = ISI_MH(wl[s_], hh[i][s_], o2);
          o2 = ISI_MH(wh[s_], hh[i][s_], o2);
        }
        const int gx = x0 + 32 * (px0 + i) + frow;
        const bool ok = gy < p.H && gx < p.W;
        const unsigned o = (unsigned)(((b * p.H + gy) * p.W + gx) * C + 32 * j + 8 * kb) * 4u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // pair8 group 4 j + 2 h + kb: accumulator quads 2 h, 2 h + 1; skip of stage 2 j + h
          // Everything in units of 4 x (the pair format's scale; powers of two commute with rounding):
          //   x4 = relu(fma(O, 4 / 4096, 4 b2) + (hi + lo))        hi + lo = 4 r exactly (two 11-bit pieces)
          // is 4 out with the reference's single rounding of (f(r) + b2) + r, and its f16 pieces hi' = f16(x4),
          // lo' = f16(x4 - hi') are the pair format of out.  Mixed-precision fmas read the f16 pieces directly
          // (v_fma_mix_f32): 7 VALU instructions per element instead of 14 (decode: 2 cvt + add + mul; encode: mul,
          // cvt, cvt back, sub, cvt) -- the tail was VALU-bound (16-21 k cycles of an item's 63 k).
          const uint4 sh = __builtin_bit_cast(uint4, skh[2 * j + h][i]), sl = __builtin_bit_cast(uint4, skl[2 * j + h][i]);
          const unsigned shw[4] = {sh.x, sh.y, sh.z, sh.w}, slw[4] = {sl.x, sl.y, sl.z, sl.w};
          const float bb[8] = {b2q[h][0].x, b2q[h][0].y, b2q[h][0].z, b2q[h][0].w, b2q[h][1].x, b2q[h][1].y, b2q[h][1].z, b2q[h][1].w};
          float x4[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float r4 = (e & 1) ? f16s::mix_sum<1>(shw[e >> 1], slw[e >> 1]) : f16s::mix_sum<0>(shw[e >> 1], slw[e >> 1]);
            float t_ = __builtin_fmaf(o2[8 * h + e], f16s::kUnscale * f16s::kScaleA, bb[e]) + r4;
            if (p.relu) t_ = t_ < 0.f ? 0.f : t_;     // NaN-propagating rectifier
            x4[e] = t_;
          }
          uint4 w0, w1;
          if (p.out_pair) {
            f16s::split2_scaled(x4[0], x4[1], w0.x, w1.x);
            f16s::split2_scaled(x4[2], x4[3], w0.y, w1.y);
            f16s::split2_scaled(x4[4], x4[5], w0.z, w1.z);
            f16s::split2_scaled(x4[6], x4[7], w0.w, w1.w);
          } else {
            constexpr float q = 1.f / f16s::kScaleA;
            w0 = make_uint4(__builtin_bit_cast(unsigned, x4[0] * q), __builtin_bit_cast(unsigned, x4[1] * q),
                            __builtin_bit_cast(unsigned, x4[2] * q), __builtin_bit_cast(unsigned, x4[3] * q));
            w1 = make_uint4(__builtin_bit_cast(unsigned, x4[4] * q), __builtin_bit_cast(unsigned, x4[5] * q),
                            __builtin_bit_cast(unsigned, x4[6] * q), __builtin_bit_cast(unsigned, x4[7] * q));
          }
          const unsigned off = (ok && !ISI_RESPAIR_ABLBIT(p, 8)) ? o + (unsigned)(64 * h) : OOB_ST;
          // (round 5, measured and dropped: handing the kb = 0 lane both lanes' first halves with v_permlane32_swap, so that a
          // store instruction writes 32-byte instead of 16-byte runs per pixel -- the form that took the first layer from 86 to
          // 77 us -- changed nothing here: the tail's burst is bound by bytes)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w0), rso_b, off, 0, ISI_RESPAIR_STORE_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w1), rso_b, off == OOB_ST ? OOB_ST : off + 16u, 0, ISI_RESPAIR_STORE_AUX);
          if (p.out2) {   // (uniform) fp32 twin of a pair-format output: the same 8 channels at the same offsets
            constexpr float q2 = 1.f / f16s::kScaleA;
            const uint4 f0 = make_uint4(__builtin_bit_cast(unsigned, x4[0] * q2), __builtin_bit_cast(unsigned, x4[1] * q2),
                                        __builtin_bit_cast(unsigned, x4[2] * q2), __builtin_bit_cast(unsigned, x4[3] * q2));
            const uint4 f1 = make_uint4(__builtin_bit_cast(unsigned, x4[4] * q2), __builtin_bit_cast(unsigned, x4[5] * q2),
                                        __builtin_bit_cast(unsigned, x4[6] * q2), __builtin_bit_cast(unsigned, x4[7] * q2));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f0), rso2_b, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f1), rso2_b, off == OOB_ST ? OOB_ST : off + 16u, 0, 0);
          }
        }
      }
      ISI_STAMP(8 + j);
    }
    ISI_STAMP(7);
  }
}
#undef ISI_MH

template <int TH, int NT, int WPR = 1>
int launch_res_pair(const ResPairK &a, hipStream_t stream) {
  constexpr int HPIX = (TH + 2) * HWD;
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;
  constexpr int NS = TH >= 8 ? 2 : 3;
  constexpr size_t ring = (size_t)NS * (A_ROWS * ROWB + 9 * 32 * ROWB + 1024);
  constexpr size_t smem = ring + (size_t)NT * 4096 + (NT * 32 + 32) * sizeof(float);   // + W2 fragments, biases
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = resblock_pair_kernel<TH, NT, WPR>;
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(resblock_pair)");
    attr_set.mark();
  }
  const int n_cu = current_device_cu_count();
  const int nitems = a.tiles_x * a.tiles_y * a.B;
  const double Cc = a.C, M = (double)a.B * a.H * a.W;
  prof::Scope scope(prof::K_RESBLOCK, 2.0 * M * 32 * 9 * Cc + 2.0 * M * Cc * 32, 4.0 * (2.0 * M * Cc + 10.0 * Cc * 32), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3(nitems < n_cu ? nitems : n_cu), dim3(TH * WPR * 64), smem, stream, a);
  return check_launch("resblock_pair_f16");
}

}  // namespace

int resblock_pair_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_respair_stamps), sizeof(long long) * (size_t)(n < 128 ? n : 128)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

// Does the DMA kernel beat resblock_f32.hip on this launch?  Measured (tools/bench_resblock.py) at B = 64: yes at the
// bottom resolution (32 x 128: 101 vs 114 us), no at the top one (16 x 64: 38 vs 32 us).  The two kernels accumulate in
// different orders, so the choice depends on the per-sample geometry ONLY: a sample's result must not depend on the
// batch it is in (tests: batch independence is bit-exact).
bool resblock_pair_preferred(int B, int H, int W, int C, int R) {
  (void)B;
  if (!resblock_pair_ok(C, R)) return false;
  if (knobs().respair_th) return true;
  // (round 3, after the rewrite: 4-row tiles also win at the top resolution -- 16 x 64 at B = 64: 28.1 us against
  // 33.6 us for resblock_f32_kernel on the same pair tensors; below that the launch does not fill the chip either way)
  return (long)H * W >= 1024;
}

bool resblock_pair_ok(int C, int R) {
  const bool off = knobs().no_resblock_pair_kernel != 0;
  return !off && R == 32 && (C == 64 || C == 128);     // instantiated channel counts (others: resblock_f32.hip)
}

// in / out: dense channels-last pair-format [B,H,W,C] (out: fp32 unless out_pair); w1_16 / w2_16: the blocked pair
// copies behind the packed 3x3 [32][9C] and 1x1 [C][32] weights.
int resblock_pair_f16(const float *in, const float *w1_16, const float *b1, const float *w2_16, const float *b2, float *out,
                      int B, int H, int W, int C, int relu, int out_pair, hipStream_t stream, float *twin, float *hidden) {
  ResPairK a;
  memset(&a, 0, sizeof a);
  if (twin && !out_pair) return unsupported("resblock_pair: an fp32 twin accompanies a pair-format output");
  a.out2 = twin; a.hidden = hidden;
  a.in = in; a.w1 = w1_16; a.b1 = b1; a.w2 = w2_16; a.b2 = b2; a.out = out;
  const int64_t elems = (int64_t)B * H * W * C;
  if (elems * 4 >= 0x70000000ll) return unsupported("resblock_pair: tensor spans 1.75 GiB or more");
  a.in_bytes = (unsigned)(elems * 4);
  a.w1_bytes = (unsigned)((size_t)32 * 9 * C * 4);
  a.w2_bytes = (unsigned)((size_t)C * 32 * 4);
  a.C = C; a.H = H; a.W = W; a.B = B; a.relu = relu; a.out_pair = out_pair;
  a.tiles_x = (W + TW - 1) / TW;
  a.ablate = knobs().respair_abl;   // 0 outside -DISI_MEASURE builds
  // 8-row tiles (8 waves, two per SIMD) when they still fill the chip, 4-row tiles otherwise
  const int forced = knobs().respair_th;
  const long tiles8 = (long)a.tiles_x * ((H + 7) / 8) * B;
  const bool th8 = forced ? forced == 8 : tiles8 >= 256;
  if (th8) {
    a.tiles_y = (H + 7) / 8;
    return C == 128 ? launch_res_pair<8, 4>(a, stream) : launch_res_pair<8, 2>(a, stream);
  }
  a.tiles_y = (H + 3) / 4;
  if (!knobs().respair_one_wave_per_row) return C == 128 ? launch_res_pair<4, 4, 2>(a, stream) : launch_res_pair<4, 2, 2>(a, stream);
  return C == 128 ? launch_res_pair<4, 4>(a, stream) : launch_res_pair<4, 2>(a, stream);
}

}  // namespace isi
