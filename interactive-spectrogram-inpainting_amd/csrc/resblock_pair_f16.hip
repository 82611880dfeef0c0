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
//   * the hidden activations go through the wave's own LDS rows into the second GEMM, whose weight fragments
//     (16 KB, L2-resident) are loaded straight into registers;
//   * epilogue per 32-pixel tile through an LDS transpose: a lane owns 8 channels of one pixel, reads r's two
//     16-byte pieces ((hi + lo) / 4: the skip connection), adds b2, rectifies and stores fp32 or pair pieces.
//
// Requirements (else resblock_f32.hip): C % 32 == 0, C <= 128, R == 32, dense channels-last tensors.
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
};

__device__ __forceinline__ void dma16(const unsigned lds_addr, const unsigned voff, const i32x4 rsrc, const unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void *ptr, const unsigned bytes) {
  const unsigned long long b = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
#define ISI_MH(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)

template <int TH>
__global__ __launch_bounds__(TH * 64) void resblock_pair_kernel(const ResPairK p) {
  constexpr int NW = TH;                                   // waves: one per tile row
  constexpr int HPIX = (TH + 2) * HWD;                     // halo pixels
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;            // padded to whole 1-KiB DMAs (16 rows of 64 B)
  constexpr int NA = A_ROWS / 16, NWD = 9 * 32 / 16;       // DMAs per stage: halo, W1 slice
  constexpr int A_BYTES = A_ROWS * ROWB, W_BYTES = 9 * 32 * ROWB, STAGE = A_BYTES + W_BYTES + 1024;   // + a dump slot for padding pieces
  constexpr int NS = TH >= 8 ? 2 : 3;                      // ring stages
  constexpr int NDMA = NA + NWD;                           // DMAs per stage
  constexpr int PER = (NDMA + NW - 1) / NW;                // per wave (the last wave(s) may have fewer: they pad with
                                                           // out-of-range pieces so that every wave's count is PER)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, kb = lane >> 5;
  const int C = p.C, nstage = ISI_RESPAIR_ABLBIT(p, 1) ? 1 : C / 16;
  const i32x4 rsi = make_rsrc(p.in, p.in_bytes), rsw = make_rsrc(p.w1, p.w1_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

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
    __syncthreads();   // the previous item is done with the LDS

    // ---- this lane's DMA pieces (constant over the stages but for the channel offset, which rides in soffset):
    // piece q of this wave is DMA number wave + NW q of the stage (halo DMAs first, then the W1 slice)
    unsigned dvo[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int d = wave + NW * q;                          // uniform
      const int row = (d < NA ? d : d - NA) * 16 + (lane >> 2);
      const unsigned piece = (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
      if (d < NA) {                                         // halo pixel `row`
        const int hy = row / HWD, hx = row - hy * HWD;
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

    // ---- fragment addresses inside a stage.  A: halo row (ry + dy) * 66 + 32 i + frow + dx; piece (2 kb + pl) sits
    // at position ^ ((row >> 2) & 3).  B: row t * 32 + frow: the swizzle term depends on the lane only.
    const int ry = wave;                                    // this wave's tile row
    const unsigned bbase = (unsigned)(A_BYTES + frow * ROWB + (((2 * kb) ^ ((frow >> 2) & 3)) << 4));

    f32x16 acc[2][2];   // [pixel tile][chain]: two chains per tile (alternating taps) keep dependent MFMAs apart
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][ch][r] = 0.f;

    // ---- prologue: NS - 1 stages in flight
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < nstage) {
#pragma unroll
        for (int q = 0; q < PER; ++q) issue(s_, s_, q);
      }

    for (int c = 0; c < nstage; ++c) {
      const int stage = c % NS;
      // this wave's pieces of stage c have landed (those of the NS - 2 newer stages may stay in flight) ...
      if (NS == 3 && c + 1 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and everyone's; everyone has also finished with the stage that slice c + NS - 1 goes to
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool more = c + NS - 1 < nstage;
      const int nst = (c + NS - 1) % NS;
      const char *st = smem + stage * STAGE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        s16x8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (ry + dy) * HWD + 32 * i + frow + dx;
          const unsigned ao = (unsigned)(row * ROWB + (((2 * kb) ^ ((row >> 2) & 3)) << 4));
          ah[i] = *reinterpret_cast<const s16x8 *>(st + ao);
          al[i] = *reinterpret_cast<const s16x8 *>(st + (ao ^ 16u));
        }
        const s16x8 bh = *reinterpret_cast<const s16x8 *>(st + bbase + t * 32 * ROWB);
        const s16x8 bl = *reinterpret_cast<const s16x8 *>(st + (bbase ^ 16u) + t * 32 * ROWB);
        // lo terms first, hi.hi last (the order of the other split-f16 kernels)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][t & 1] = ISI_MH(al[i], bh, acc[i][t & 1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][t & 1] = ISI_MH(ah[i], bl, acc[i][t & 1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][t & 1] = ISI_MH(ah[i], bh, acc[i][t & 1]);
        // one or two DMAs of the slice NS - 1 ahead behind each tap's MFMAs
        if (more) {
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if (q * 9 / PER == t) issue(nst, c + NS - 1, q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    if (ISI_RESPAIR_ABLBIT(p, 2)) { if (acc[0][0][0] == 123.f && acc[1][1][3] == 7.f) p.out[0] = acc[0][1][5] + acc[1][0][2]; continue; }
    // ---- hidden activations h = relu(acc + b1) -> this wave's LDS rows as pair planes [64 px][32] (A operand of GEMM 2)
    __syncthreads();                     // every wave is done with the ring
    constexpr int HROW = 128;            // bytes per pixel: 8 pieces of 16 B, piece 2 g + plane of hidden-channel group g
    char *hb = smem + wave * 64 * HROW;  // [64 px][8 pieces of 16 B], piece = 2 g + plane, position ^ ((px >> 1) & 7)
    {
      const float b1v = p.b1[frow];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kb;
          const float hpre = (acc[i][0][r] + acc[i][1][r]) * f16s::kUnscale + b1v;
          const float hv = (hpre < 0.f ? 0.f : hpre) * f16s::kScaleA;   // NaN-propagating rectifier
          const _Float16 hh = (_Float16)hv;
          const _Float16 hl = (_Float16)(hv - (float)hh);
          const int g = frow >> 3, e = frow & 7;   // hidden channel frow = 8 g + e
          const int sw = (px >> 1) & 7;
          *reinterpret_cast<_Float16 *>(hb + px * HROW + (((2 * g) ^ sw) << 4) + e * 2) = hh;
          *reinterpret_cast<_Float16 *>(hb + px * HROW + (((2 * g + 1) ^ sw) << 4) + e * 2) = hl;
        }
    }
    // (Tried instead of the LDS planes / transposes of this tail, both correct, neither faster: (a) everything in
    // registers -- both GEMMs with swapped operands so that GEMM 1's accumulators are GEMM 2's operand fragments and
    // GEMM 2's accumulators are stored as 16-byte pieces per lane, v_permlane32_swap for the skip and pair pieces as in
    // conv_pair_f16.hip: 106 us against 104, every load / store instruction then touches 32 cache lines instead of 8 and
    // the memory part of the tail grows from 21 to 31 us, which is what the saved LDS work gains; (b) registers for the
    // hidden activations only, LDS transposes kept: 111 us, the 8-byte W2 fragment reads of the permuted k order cost
    // more than the 2-byte LDS writes they replace.)
    // W2 fragments come straight from memory (B operand of GEMM 2: row n = 32 j + frow, k-step s, k-block kb; 16 KB,
    // L2-resident).  (Holding all of them, or all of a tile's skip pieces, in registers at once was measured: the
    // kernel then needs 256 VGPRs with spills and runs 25 % slower.)
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w2), 0, p.w2_bytes, 0x00020000);
    const int ntile = C / 32;
    const __amdgpu_buffer_rsrc_t rsi_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso_b = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.in_bytes, 0x00020000);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- per 32-pixel tile and pair of 32-channel output tiles: GEMM 2 (K = 32), transpose through LDS, skip +
    // bias + ReLU, 16-byte stores
    constexpr int LDT = 68;                                                             // [32 px][64 + 4] floats
    float *tb = reinterpret_cast<float *>(smem + NW * 64 * HROW) + wave * 32 * LDT;
    const int gy = y0 + ry;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      s16x8 hh[2], hl[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const int px = 32 * i + frow, sw = (px >> 1) & 7;
        hh[s_] = *reinterpret_cast<const s16x8 *>(hb + px * HROW + (((2 * (2 * s_ + kb)) ^ sw) << 4));
        hl[s_] = *reinterpret_cast<const s16x8 *>(hb + px * HROW + (((2 * (2 * s_ + kb) + 1) ^ sw) << 4));
      }
      for (int j0 = 0; j0 < ntile; j0 += 2) {
        const int nj = ntile - j0 < 2 ? 1 : 2;               // uniform
        for (int jj = 0; jj < nj; ++jj) {
          const int j = j0 + jj;
          f32x16 o2;
#pragma unroll
          for (int r = 0; r < 16; ++r) o2[r] = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_) {
            const unsigned wo = (unsigned)((32 * j + frow) * 32 + (2 * s_ + kb) * 8) * 4u;
            const i32x4 wh = __builtin_amdgcn_raw_buffer_load_b128(rs2, wo, 0, 0);
            const i32x4 wl = __builtin_amdgcn_raw_buffer_load_b128(rs2, wo + 16u, 0, 0);
            o2 = ISI_MH(hl[s_], wh, o2);
            o2 = ISI_MH(hh[s_], wl, o2);
            o2 = ISI_MH(hh[s_], wh, o2);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * kb) * LDT + 32 * jj + frow] = o2[r] * f16s::kUnscale;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane -> (pixel, 8-channel group): GP groups per pixel in this pass, 64 / GP pixels per step
        const int GP = nj * 4, PPP = 64 / GP;
        const int g = lane % GP, psub = lane / GP;
        const int ch0 = 32 * j0 + g * 8;                      // first of this lane's 8 channels
        float b2v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b2v[e] = p.b2[ch0 + e];
        for (int it = 0; it < 32 / PPP; ++it) {
          const int px = it * PPP + psub;
          const int gx = x0 + 32 * i + px;
          const unsigned off = (gy < p.H && gx < p.W) ? (unsigned)(((b * p.H + gy) * p.W + gx) * C + ch0) * 4u : OOB_ST;
          const unsigned roff = ISI_RESPAIR_ABLBIT(p, 4) ? OOB_ST : off;   // measurement: no skip re-read
          const float4 v0 = *reinterpret_cast<const float4 *>(tb + px * LDT + g * 8);
          const float4 v1 = *reinterpret_cast<const float4 *>(tb + px * LDT + g * 8 + 4);
          // the skip connection: r's two pieces of this group
          const i32x4 rh = __builtin_amdgcn_raw_buffer_load_b128(rsi_b, roff, 0, 0);
          const i32x4 rl = __builtin_amdgcn_raw_buffer_load_b128(rsi_b, roff == OOB_ST ? OOB_ST : roff + 16u, 0, 0);
          float4 r0, r1;
          f16s::pair8_decode(__builtin_bit_cast(uint4, rh), __builtin_bit_cast(uint4, rl), r0, r1);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t_ = (v[e] + b2v[e]) + rr[e];
            if (p.relu) t_ = t_ < 0.f ? 0.f : t_;
            v[e] = t_;
          }
          uint4 w0, w1;
          if (p.out_pair) {
            f16s::pair8_encode(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), w0, w1);
          } else {
            w0 = make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                            __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
            w1 = make_uint4(__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]),
                            __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7]));
          }
          const unsigned soff = ISI_RESPAIR_ABLBIT(p, 8) ? OOB_ST : off;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w0), rso_b, soff, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, w1), rso_b, soff == OOB_ST ? OOB_ST : soff + 16u, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the wave's transpose rows are reused by its next pass
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
}
#undef ISI_MH

template <int TH>
int launch_res_pair(const ResPairK &a, hipStream_t stream) {
  constexpr int HPIX = (TH + 2) * HWD;
  constexpr int A_ROWS = (HPIX + 15) / 16 * 16;
  constexpr int NS = TH >= 8 ? 2 : 3;
  constexpr size_t ring = (size_t)NS * (A_ROWS * ROWB + 9 * 32 * ROWB + 1024);
  constexpr size_t tail = (size_t)TH * 64 * 128 + (size_t)TH * 32 * 68 * sizeof(float);   // h planes + transposes
  constexpr size_t smem = ring > tail ? ring : tail;
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = resblock_pair_kernel<TH>;
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
  ISI_PROF_LAUNCH(scope, kern, dim3(nitems < n_cu ? nitems : n_cu), dim3(TH * 64), smem, stream, a);
  return check_launch("resblock_pair_f16");
}

}  // namespace

// Does the DMA kernel beat resblock_f32.hip on this launch?  Measured (tools/bench_resblock.py) at B = 64: yes at the
// bottom resolution (32 x 128: 101 vs 114 us), no at the top one (16 x 64: 38 vs 32 us).  The two kernels accumulate in
// different orders, so the choice depends on the per-sample geometry ONLY: a sample's result must not depend on the
// batch it is in (tests: batch independence is bit-exact).
bool resblock_pair_preferred(int B, int H, int W, int C, int R) {
  (void)B;
  if (!resblock_pair_ok(C, R)) return false;
  if (knobs().respair_th) return true;
  return (long)H * W >= 2048;
}

bool resblock_pair_ok(int C, int R) {
  const bool off = knobs().no_resblock_pair_kernel != 0;
  return !off && R == 32 && C % 32 == 0 && C >= 32 && C <= 128;
}

// in / out: dense channels-last pair-format [B,H,W,C] (out: fp32 unless out_pair); w1_16 / w2_16: the blocked pair
// copies behind the packed 3x3 [32][9C] and 1x1 [C][32] weights.
int resblock_pair_f16(const float *in, const float *w1_16, const float *b1, const float *w2_16, const float *b2, float *out,
                      int B, int H, int W, int C, int relu, int out_pair, hipStream_t stream) {
  ResPairK a;
  memset(&a, 0, sizeof a);
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
    return launch_res_pair<8>(a, stream);
  }
  a.tiles_y = (H + 3) / 4;
  return launch_res_pair<4>(a, stream);
}

}  // namespace isi
