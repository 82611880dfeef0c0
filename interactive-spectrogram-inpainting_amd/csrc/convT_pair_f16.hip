// ConvTranspose2d(k4, s2, p1) of the split-f16 PAIR pipeline for gfx950, the output phases FUSED on one staged input tile.
//
//   out[b][2y+py][2x+px][n] = bias[n] + sum_{ty,tx in {0,1}} sum_c in[b][y-1+py+ty][x-1+px+tx][c] W[c][n][3-py-2ty][3-px-2tx]
//
// i.e. four stride-1 2x2 "phase" convolutions (reference: nn.ConvTranspose2d at vqvae/encoder_decoder.py:196-216 and
// vqvae/vqvae.py:183-201).  The phase-per-launch form (conv_igemm_f32.hip, blockIdx.z = phase) staged every input
// pixel once per phase AND per tap -- 16 shifted reads of each pixel, 0.22-0.34 of the split-f16 matrix ceiling, 2.0x
// the algorithmic HBM traffic, 32 % of its LDS cycles in bank conflicts.  Here a work item is
//
//     (sample, TH x 64 input pixels, py)  ->  out rows 2y+py, BOTH px phases, 64 output channels      (N = 2 x 64)
//
//   * the (TH+1) x 66 input halo of a 16-channel stage is brought into LDS ONCE by LDS-DMA (`buffer_load ... lds`,
//     counted s_waitcnt: see conv_pair_f16.hip) and the six taps (ty, sx = px + tx) read shifted windows of it; the
//     eight (px, ty, tx) weight slices [64 n][16 c] of the stage arrive the same way.  Ring of 2 stages, the next
//     stage's DMAs issued between the MFMAs of the current one, ONE barrier per 96 MFMAs per wave.
//   * a wave owns one row of 64 pixels: 2 pixel tiles x 2 px phases x 2 channel tiles = 8 accumulator tiles; each
//     A fragment (pixels) feeds up to 12 MFMAs, each B fragment (weights) 6.
//   * products as everywhere in the pipeline: hi.hi + hi.lo + lo.hi of two f16 pieces (split_f16.h), lo terms first.
//   * MFMA operands swapped (weights = rows, pixels = columns) so that a lane's accumulators are quads of consecutive
//     channels of ONE output pixel: bias (from LDS: a vector-memory load would wait in order behind the previous
//     item's stores), ReLU and the stores (fp32 quads or pair8 pieces after v_permlane32_swap) run straight from the
//     registers -- no LDS transpose, no barrier (conv_pair_f16.hip's epilogue).
//   * 16-byte pieces of a 64-byte LDS row XOR-swizzled on the DMA's SOURCE side -- weight rows with (row >> 2) & 3,
//     halo rows with (column >> 2) & 3: conflict-free ds_read_b128, and a tap's window address is an immediate away
//     from three per-lane registers.
//
// The K order differs from the phase-per-launch kernel's (16-channel stages, taps inside), so the two agree to
// rounding, not bit for bit.  Requirements (else the phase-per-launch kernel): pair8 channels-last dense input with
// Cin % 16 == 0, Cout % 64 == 0, channels-last dense output.
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
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
using f16s::f16x8;

constexpr int TW = 64, HWD = TW + 2;
constexpr int ROWB = 64;                       // bytes of a row per 16-channel stage: {hi g0, lo g0, hi g1, lo g1}
constexpr int NPAIR = 8;                       // (px, ty, tx) weight slices per stage
constexpr unsigned OOB = 0x7FFFFFF0u, OOB_ST = 0xFFFFFFF0u;

struct ConvTPairK {
  const float *in, *w16, *bias;
  float *out;
  unsigned in_bytes, w_bytes, out_bytes;
  int Cin, Cout, Kpad, H, W, B, relu;
  int tiles_x, tiles_y, ntn;                   // ntn = Cout / 64
  // YP mode (the decoder's tail, see below): packed fp32 weight [n2][64] of the few-channel transposed convolution that
  // follows this one; `out` then receives Y' [B][2H][2W][32] fp32 instead of this layer's activation
  const float *w2;
  int n2;
  float *out2;                                 // OUTP launches: dense fp32 twin of the output (training tape), or null
};

__device__ __forceinline__ void dma16(const unsigned lds_addr, const unsigned voff, const i32x4 rsrc, const unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void *ptr, const unsigned bytes) {
  const unsigned long long b = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// phase timestamps (-DISI_MEASURE builds; tools/stamps_convT.py): workgroup 8, waves 0 and 4, its second work item
#ifdef ISI_MEASURE
__device__ long long g_convT_pair_stamps[128];
#define ISI_STAMP(i_) do { if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && item_i == (int)blockIdx.x + (int)gridDim.x) \
    g_convT_pair_stamps[(wave >> 2) * 64 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_STAMP(i_) do { } while (0)
#endif
#define ISI_MH(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)

// YP (round 3, "decoder tail"): this layer (ConvT C -> 64, + bias, ReLU) is followed by ConvTranspose2d(64 -> n2 / 16 <= 2)
// (vqvae/encoder_decoder.py:196-209).  A transposed convolution is a per-pixel projection Y' = W2^T u onto its 16 taps x
// outputs followed by a col2im sum; the projection needs nothing but the pixel's own 64 channels, which at the end of this
// kernel's K loop sit in the lane's registers.  So instead of writing u (256 bytes per pixel, the largest tensor of the
// network, read once more by the next kernel) the epilogue rectifies and splits u in registers, multiplies by W2 (12 MFMAs
// per 32-pixel group; the accumulator quads ARE the B fragments once W2's k order is permuted to match) and writes the
// 32 floats of Y' per pixel (128 bytes): half the bytes out of this kernel, half the bytes into the next, which is left
// with the col2im gather (convT_small_f32.hip: convT_gather_kernel).  Same products as the two-kernel path (u split into
// the same f16 pieces, three-term products with W2's pieces), other summation order.
template <int TH, bool OUTP, bool YP = false>
__global__ __launch_bounds__(TH * 64) void convT_pair_kernel(const ConvTPairK p) {
  constexpr int NW = TH;                                   // waves: one per tile row
  constexpr int HPIX = (TH + 1) * HWD;                     // halo pixels: rows y0-1+py .. y0+TH-1+py, columns x0-1 .. x0+64
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;            // padded to whole 1-KiB DMAs (16 rows of 64 B)
  constexpr int NA = A_ROWS / 16, NWD = NPAIR * 64 / 16;   // DMAs per stage: halo, weight slices
  constexpr int A_BYTES = A_ROWS * ROWB, W_BYTES = NPAIR * 64 * ROWB, STAGE = A_BYTES + W_BYTES + 1024;   // + dump slot
  constexpr int NS = 2;                                   // ring stages (three do not fit beside the 32-KB weight slices)
  constexpr int NDMA = NA + NWD;
  constexpr int PER = (NDMA + NW - 1) / NW;                // per wave (padding pieces keep every wave's count at PER)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, kb = lane >> 5;
  const int Cin = p.Cin, nstage = Cin / 16;
  const i32x4 rsi = make_rsrc(p.in, p.in_bytes), rsw = make_rsrc(p.w16, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rso_b = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso2_b = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, p.out2 ? p.out_bytes : 0u, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int OW = 2 * p.W, OH = 2 * p.H;

  // bias -> LDS once per workgroup: a vector-memory load in the epilogue would have to be waited for IN ORDER behind
  // the previous stores (vmcnt counts both) and turn their latency into epilogue time (measured: 19.5 k of an item's
  // 98 k cycles); a ds_read is counted separately
  float *bias_s = reinterpret_cast<float *>(smem + NS * STAGE);
  for (int i = tid; i < p.Cout; i += NW * 64) bias_s[i] = p.bias ? p.bias[i] : 0.f;
  // YP: W2 as MFMA A fragments (row n = tap * outputs + co, zero beyond n2), piece (k-step s, plane) of lane (n, kb) at
  // ((s 2 + plane) 64 + lane) 16 B; k-slot (s, kb, e) is channel 32 (s >> 1) + 16 (s & 1) + (e < 4 ? 4 kb + e : 8 + 4 kb + e - 4)
  // -- the order in which a lane's accumulator quads 2 (s & 1), 2 (s & 1) + 1 of channel tile s >> 1 hold its pixel
  unsigned short *w2s = reinterpret_cast<unsigned short *>(bias_s + 64);
  if constexpr (YP) {
    for (int i = tid; i < 4 * 2 * 64 * 8; i += NW * 64) {
      const int e = i & 7, l = (i >> 3) & 63, plane = (i >> 9) & 1, s_ = i >> 10;
      const int n = l & 31, kbl = l >> 5;
      const int ch = 32 * (s_ >> 1) + 16 * (s_ & 1) + (e < 4 ? 4 * kbl + e : 8 + 4 * kbl + (e - 4));
      const float w = n < p.n2 ? p.w2[n * 64 + ch] * f16s::kScaleB : 0.f;
      const _Float16 hi = (_Float16)w;
      const _Float16 v = plane ? (_Float16)(w - (float)hi) : hi;
      w2s[i] = __builtin_bit_cast(unsigned short, v);
    }
  }

  const int per_b = p.tiles_x * p.tiles_y * 2 * p.ntn;
  const int nitems = per_b * p.B;
  for (int item_i = blockIdx.x; item_i < nitems; item_i += gridDim.x) {
    int item;
    {   // XCD-aware order: the items that share halo rows / the same input tile (py, n tile) on one XCD
      const int q = nitems / 8, r = nitems % 8, xcd = item_i % 8, idx = item_i / 8;
      item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = item / per_b;
    int rem = item - b * per_b;
    const int tile = rem / (2 * p.ntn);
    rem -= tile * (2 * p.ntn);
    const int py = rem / p.ntn, n0 = (rem - py * p.ntn) * 64;
    const int y0 = (tile / p.tiles_x) * TH, x0 = (tile % p.tiles_x) * TW;
    ISI_STAMP(0);
    __syncthreads();   // the previous item is done with the LDS
    ISI_STAMP(1);

    // ---- this lane's DMA pieces (constant over the stages but for the channel offset, which rides in soffset):
    // piece q of this wave is DMA number wave + NW q of the stage (halo DMAs first, then the weight slices)
    unsigned dvo[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int d = wave + NW * q;                          // uniform
      const int row = (d < NA ? d : d - NA) * 16 + (lane >> 2);
      if (d < NA) {                                         // halo pixel `row`: swizzled with its COLUMN (below)
        const int hy = row / HWD, hx = row - hy * HWD;
        const unsigned piece = (unsigned)(((lane & 3) ^ ((hx >> 2) & 3)) * 16);
        const int gy = y0 - 1 + py + hy, gx = x0 - 1 + hx;
        const bool ok = row < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        dvo[q] = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * Cin) * 4u + piece : OOB;
      } else if (d < NDMA) {                                // weight row: slice (px, ty, tx) * 64 + output channel
        const unsigned piece = (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
        const int sl = row >> 6, n = row & 63;
        const int px = sl >> 2, tap = sl & 3;               // tap = ty * 2 + tx (the packed k order)
        dvo[q] = (unsigned)((((py * 2 + px) * p.Cout + n0 + n) * p.Kpad) + tap * Cin) * 4u + piece;
      } else {
        dvo[q] = OOB;                                       // padding piece
      }
    }
    auto issue = [&](const int stage, const int c16, const int q) {   // piece q of channel slice c16 -> ring stage
      const int d = wave + NW * q;                          // uniform
      const bool is_w = d >= NA;
      const unsigned dst = d < NDMA ? (unsigned)(d * 1024) : (unsigned)(A_BYTES + W_BYTES);   // (A_BYTES = NA KiB)
      dma16(lds0 + (unsigned)(stage * STAGE) + dst, dvo[q], is_w && d < NDMA ? rsw : rsi, d < NDMA ? (unsigned)(c16 * 64) : 0u);
    };

    // ---- fragment addresses inside a stage.  A: halo row (ry + ty) * 66 + hx, hx = 32 i + frow + sx; piece
    // (2 kb + pl) sits at position ^ ((hx >> 2) & 3): swizzling with the COLUMN instead of the linear row keeps the
    // reads conflict-free (rows of a 16-lane group that share a bank quarter still differ in hx >> 2) and makes the
    // per-lane part of the address depend on sx only -- 3 x 2 address registers, (ty, i) ride in the instruction's
    // immediate offset (the linear-row form needed 24).  B: row slice * 64 + 32 j + frow: lane-only swizzle term.
    const int ry = wave;                                    // this wave's tile row
    unsigned abase[3][2];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const int hx = frow + sx;
      abase[sx][0] = (unsigned)((ry * HWD + hx) * ROWB + (((2 * kb) ^ ((hx >> 2) & 3)) << 4));
      abase[sx][1] = abase[sx][0] ^ 16u;
    }
    const unsigned bbase = (unsigned)(A_BYTES + frow * ROWB + (((2 * kb) ^ ((frow >> 2) & 3)) << 4));
    const unsigned bbase1 = bbase ^ 16u;

    f32x16 acc[2][2][2];   // [px][pixel tile i][channel tile j]
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a_][i][j][r] = 0.f;

    ISI_STAMP(2);
    // ---- prologue: NS - 1 stages in flight
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < nstage) {
#pragma unroll
        for (int q = 0; q < PER; ++q) issue(s_, s_, q);
      }

    ISI_STAMP(3);
    for (int c = 0; c < nstage; ++c) {
      const int stage = c % NS;
      ISI_STAMP(8 + 3 * c);
      // this wave's pieces of stage c have landed (those of the NS - 2 newer stages may stay in flight) ...
      if (NS == 3 && c + 1 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and everyone's; everyone has also finished with the stage that slice c + NS - 1 goes to
      ISI_STAMP(9 + 3 * c);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ISI_STAMP(10 + 3 * c);
      const bool more = c + NS - 1 < nstage;
      const int nst = (c + NS - 1) % NS;
      const char *st = smem + stage * STAGE;
      // The stage's eight (ty, sx, px) steps.  (Requesting step k + 1's fragments ahead of step k's MFMAs from a
      // second register set was measured: same time -- the partner wave of the SIMD covers a step's LDS latency --
      // at 256 VGPRs with spills in the item set-up.)  The stage's DMAs go out during the first steps: they then have
      // most of a stage to land before the vmcnt(0) at the top of the next one.
      int step = 0;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          s16x8 ah[2], al[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const s16x8 *>(st + abase[sx][0] + (ty * HWD + 32 * i) * ROWB);
            al[i] = *reinterpret_cast<const s16x8 *>(st + abase[sx][1] + (ty * HWD + 32 * i) * ROWB);
          }
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int tx = sx - px;
            if (tx < 0 || tx > 1) continue;
            const int sl = px * 4 + ty * 2 + tx;
            s16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              bh[j] = *reinterpret_cast<const s16x8 *>(st + bbase + (sl * 64 + 32 * j) * ROWB);
              bl[j] = *reinterpret_cast<const s16x8 *>(st + bbase1 + (sl * 64 + 32 * j) * ROWB);
            }
            // operands swapped: weights are the MFMA's rows, pixels its columns; lo terms first, hi.hi last
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[px][i][j] = ISI_MH(bh[j], al[i], acc[px][i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[px][i][j] = ISI_MH(bl[j], ah[i], acc[px][i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[px][i][j] = ISI_MH(bh[j], ah[i], acc[px][i][j]);
            if (more) {
#pragma unroll
              for (int q = 0; q < PER; ++q)
                if (q / 2 == step) issue(nst, c + NS - 1, q);   // two per step: all out by step (PER - 1) / 2
            }
            ++step;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }

    ISI_STAMP(4);
    // ---- epilogue, straight from the accumulators: lane (pixel = frow of tile i, kb) holds, per channel tile j and
    // quad q = r >> 2, the channels 32 j + 8 q + 4 kb + (r & 3) of output pixel (2 y + py, 2 x + px).
    // Where an item's time goes (tools/stamps_convT.py, 128 -> 64 at 32 x 128, B = 64: 86 k cycles per item): K loop
    // 61 k; the 256 KB of stores 10-15 k plus 4-6 k during which the next item's first DMAs queue behind them -- every
    // CU reaches its epilogue at the same time and the burst runs at the fabric's write rate while HBM idles during the
    // K loops.  Regrouping the stores into whole 128-byte lines through a wave-private LDS transpose (the ring stage
    // the last K stage leaves idle; no block barrier) was built and measured: 190.7 against 188.3 us -- the burst is
    // bound by bytes, not by lines per instruction.  What removes it is not writing this tensor at all (fusing the
    // 64 -> 2 transposed convolution behind it: DESIGN.md section 8).
    const int gy = y0 + ry;
    if constexpr (YP) {
      // u = relu(acc + bias) in units of 4 x, split into f16 pieces: the B fragments of Y'^T = W2^T u^T
      s16x8 wfh[4], wfl[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        wfh[s_] = *reinterpret_cast<const s16x8 *>(w2s + ((s_ * 2 + 0) * 64 + lane) * 8);
        wfl[s_] = *reinterpret_cast<const s16x8 *>(w2s + ((s_ * 2 + 1) * 64 + lane) * 8);
      }
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x16 y;
#pragma unroll
          for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) {
            const int j = s_ >> 1, qa = 2 * (s_ & 1);
            float u4[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int q = qa + (e >> 2);
              const float bv = bias_s[32 * j + 8 * q + 4 * kb + (e & 3)];
              float t = __builtin_fmaf(acc[px][i][j][4 * q + (e & 3)], f16s::kUnscale * f16s::kScaleA, bv * f16s::kScaleA);
              if (p.relu) t = t < 0.f ? 0.f : t;
              u4[e] = t;
            }
            uint4 uh, ul;
            f16s::split2_scaled(u4[0], u4[1], uh.x, ul.x);
            f16s::split2_scaled(u4[2], u4[3], uh.y, ul.y);
            f16s::split2_scaled(u4[4], u4[5], uh.z, ul.z);
            f16s::split2_scaled(u4[6], u4[7], uh.w, ul.w);
            const s16x8 uhv = __builtin_bit_cast(s16x8, uh), ulv = __builtin_bit_cast(s16x8, ul);
            y = ISI_MH(wfh[s_], ulv, y);
            y = ISI_MH(wfl[s_], uhv, y);
            y = ISI_MH(wfh[s_], uhv, y);
          }
          const int gx = x0 + 32 * i + frow;
          const bool ok = gy < p.H && gx < p.W;
          const unsigned o = (unsigned)(((b * OH + 2 * gy + py) * OW + 2 * gx + px) * 32);   // Y': 32 floats per pixel
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 w = make_uint4(__builtin_bit_cast(unsigned, y[4 * q] * f16s::kUnscale), __builtin_bit_cast(unsigned, y[4 * q + 1] * f16s::kUnscale),
                                       __builtin_bit_cast(unsigned, y[4 * q + 2] * f16s::kUnscale), __builtin_bit_cast(unsigned, y[4 * q + 3] * f16s::kUnscale));
            const unsigned off = ok ? (o + (unsigned)(8 * q + 4 * kb)) * 4u : OOB_ST;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w), rso_b, off, 0, 0);
          }
        }
    } else {
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gx = x0 + 32 * i + frow;
        const bool ok = gy < p.H && gx < p.W;
        const unsigned o = (unsigned)(((b * OH + 2 * gy + py) * OW + 2 * gx + px) * p.Cout + n0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cg = 32 * j + 8 * q;
            const float4 bq = *reinterpret_cast<const float4 *>(bias_s + n0 + cg + 4 * kb);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = __builtin_fmaf(acc[px][i][j][4 * q + e], f16s::kUnscale, e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w);
              if (p.relu) t = t < 0.f ? 0.f : t;   // NaN < 0 is false: a NaN stays (torch.relu)
              v[e] = t;
            }
            uint4 w;
            unsigned off;
            if constexpr (OUTP) {
              uint2 hi, lo;
              f16s::split4(make_float4(v[0], v[1], v[2], v[3]), f16s::kScaleA, hi, lo);
              const u32x2v sx_ = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
              const u32x2v sy_ = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
              w = make_uint4(sx_.x, sy_.x, sx_.y, sy_.y);
              off = ok ? (o + (unsigned)cg) * 4u + (unsigned)kb * 16u : OOB_ST;
              if (p.out2) {   // (uniform) the same values as fp32 for the backward's tape
                const uint4 wf = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                            __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, wf), rso2_b,
                                                       ok ? (o + (unsigned)(cg + 4 * kb)) * 4u : OOB_ST, 0, 0);
              }
            } else {
              w = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                             __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
              off = ok ? (o + (unsigned)(cg + 4 * kb)) * 4u : OOB_ST;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w), rso_b, off, 0, 0);
          }
      }
    }
    ISI_STAMP(5);
  }
}
#undef ISI_MH

template <int TH, bool OUTP, bool YP = false>
int launch_convT_pair(const ConvTPairK &a, hipStream_t stream) {
  constexpr int HPIX = (TH + 1) * HWD;
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;
  constexpr int NS = 2;
  constexpr size_t ring = (size_t)NS * (A_ROWS * ROWB + NPAIR * 64 * ROWB + 1024);
  static_assert(ring + 4096 <= 160 * 1024, "LDS budget");
  const size_t smem = ring + (size_t)(a.Cout > 64 ? a.Cout : 64) * sizeof(float) + (YP ? 8192 : 0);   // + the bias, W2 fragments
  auto kern = convT_pair_kernel<TH, OUTP, YP>;
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(convT_pair)");
    attr_set.mark();
  }
  const int n_cu = current_device_cu_count();
  const int nitems = a.tiles_x * a.tiles_y * 2 * a.ntn * a.B;
  const double M = (double)a.B * a.H * a.W;
  const double flops = 2.0 * M * 4 * a.Cout * 4.0 * a.Cin;
  const double bytes = 4.0 * (M * a.Cin + 4.0 * M * a.Cout + 16.0 * a.Cin * a.Cout);
  prof::Scope scope(prof::K_CONV_F16X3, flops, bytes, stream);
  ISI_PROF_LAUNCH(scope, kern, dim3(nitems < n_cu ? nitems : n_cu), dim3(TH * 64), smem, stream, a);
  return check_launch("convT_pair_f16");
}

}  // namespace

bool decoder_tail_ok(int Cin, int Cmid, int Cout) {
  return !knobs().no_tail_fusion && convT_pair_ok(Cin, Cmid) && Cmid == 64 && Cout >= 1 && Cout <= 2;
}

int decoder_tail_f32(const float *in_pair, const float *packed_w1, const float *bias1, const float *packed_w2,
                     const float *bias2, float *yprime_ws, const isi_dst *dst, int B, int H, int W, int Cin, int Cmid,
                     int Cout, hipStream_t stream) {
  if (!in_pair || !packed_w1 || !packed_w2 || !yprime_ws || !dst || !dst->ptr) return invalid("decoder_tail: null pointer");
  if (B <= 0 || H <= 0 || W <= 0) return invalid("decoder_tail: bad shape");
  if (!convT_pair_ok(Cin, Cmid) || Cmid != 64 || Cout < 1 || Cout > 2) return unsupported("decoder_tail: Cin % 16 == 0, Cmid == 64, Cout <= 2");
  const size_t Kpad = round_up((size_t)4 * Cin, kBK);
  int rc = convT_pair_f16(in_pair, packed_w1 + (size_t)4 * Cmid * Kpad, bias1, yprime_ws, B, H, W, Cin, Cmid, /*relu*/ 1,
                          /*out_pair*/ 0, stream, packed_w2, 16 * Cout);
  if (rc) return rc;
  return convT_gather_f32(yprime_ws, bias2, dst->ptr, B, 2 * H, 2 * W, Cout, (int)dst->sn, (int)dst->sc, (int)dst->sh,
                          (int)dst->sw, 0, stream);
}

int convT_pair_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_convT_pair_stamps), sizeof(long long) * (size_t)(n < 128 ? n : 128)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

bool convT_pair_ok(int Cin, int Cout) {
  return !knobs().no_convt_pair_kernel && Cin >= 16 && Cin % 16 == 0 && Cout % 64 == 0 && Cout <= 1024;
}

// in: dense channels-last pair-format [B,H,W,Cin]; out: dense channels-last [B,2H,2W,Cout], fp32 or pair format;
// w16: the blocked pair copy behind the packed phase matrices [4][Cout][Kpad] (pack_convT_k4s2_weight + ISI_CONV_W16).
// w2 != nullptr: the decoder-tail form (YP): Cout must be 64, `out` receives Y' [B][2H][2W][32] fp32 for the n2 <= 32
// rows of the packed weight w2 [n2][64] of the few-channel transposed convolution behind this layer
int convT_pair_f16(const float *in, const float *w16, const float *bias, float *out, int B, int H, int W, int Cin,
                   int Cout, int relu, int out_pair, hipStream_t stream, const float *w2, int n2, float *twin) {
  if (!convT_pair_ok(Cin, Cout)) return unsupported("convT_pair: shape outside the fused kernel");
  if (twin && (!out_pair || w2)) return unsupported("convT_pair: an fp32 twin accompanies a pair-format output");
  if (w2 && (Cout != 64 || n2 < 1 || n2 > 32 || out_pair)) return unsupported("convT_pair: the tail form needs Cout == 64 and n2 <= 32");
  const int64_t ein = (int64_t)B * H * W * Cin, eout = (int64_t)B * 4 * H * W * (w2 ? 32 : Cout);
  if (ein * 4 >= 0x70000000ll || eout * 4 >= 0xF0000000ll) return unsupported("convT_pair: tensor too large for 32-bit offsets");
  ConvTPairK a;
  memset(&a, 0, sizeof a);
  a.in = in; a.w16 = w16; a.bias = bias; a.out = out;
  a.Cin = Cin; a.Cout = Cout; a.Kpad = (int)round_up((size_t)4 * Cin, kBK);
  a.in_bytes = (unsigned)(ein * 4); a.out_bytes = (unsigned)(eout * 4);
  a.w_bytes = (unsigned)((size_t)4 * Cout * a.Kpad * 4);
  a.H = H; a.W = W; a.B = B; a.relu = relu;
  a.tiles_x = (W + TW - 1) / TW;
  a.ntn = Cout / 64;
  a.w2 = w2; a.n2 = n2; a.out2 = twin;
  // 8-row tiles (8 waves, two per SIMD) when they still give every CU an item, 4-row tiles otherwise
  const int forced = knobs().convt_pair_th;
  const long items8 = (long)a.tiles_x * ((H + 7) / 8) * 2 * a.ntn * B;
  const bool th8 = forced == 8 || (forced != 4 && items8 >= 256);
  if (th8) {
    a.tiles_y = (H + 7) / 8;
    if (w2) return launch_convT_pair<8, false, true>(a, stream);
    return out_pair ? launch_convT_pair<8, true>(a, stream) : launch_convT_pair<8, false>(a, stream);
  }
  a.tiles_y = (H + 3) / 4;
  if (w2) return launch_convT_pair<4, false, true>(a, stream);
  return out_pair ? launch_convT_pair<4, true>(a, stream) : launch_convT_pair<4, false>(a, stream);
}

}  // namespace isi
