// Relative-position attention forward for gfx950, third generation (round 6): operands pre-split into 16-bit planes,
// staged by LDS-DMA, two wave groups half a step apart.
//
//   logit[i,j] = ( q_i.k_j + q_i.e[h, i - j + Ek - 1] ) * scale + mask(i,j)          (one channel per event: Cq = Ck = 1)
//   out_i      = sum_j softmax_j(logit[i,:]) v_j
//
// The operator behind the reference's VQCPCB.transformer.transformer_custom layers
// (priors/transformer.py:370-417,756-777); specification: oracle/prior_oracle.py (parity unpinned).
//
// What changed against rel_attn_fwd2_kernel (rel_attention_fwd2.hip), by that kernel's own stamps and counters (matrix pipe
// 11-22 % busy; a step = prefetch into registers, convert, ds_write, two block barriers; 64 KB ring of e rows; the band
// product E Q^T computed twice per (query, key) tile):
//   * K, V and the table e arrive as 16-bit planes (hi, and lo for three-term products) made ONCE per call by
//     attn_pack_kernel -- the in-projection's output converted element by element, row-major [pair][key][head_dim].  A
//     K / V tile is a pure copy: `buffer_load_dwordx4 ... lds` (no staging registers, no conversion, no ds_write), 16-byte
//     pieces XOR-swizzled on the SOURCE side (a DMA's destination is lane-linear) so that ds_read_b128 fragments (K) and
//     ds_read_b64_tr_b16 transposing fragments (V^T out of the row-major V tile) are bank-conflict free;
//   * the band rows of e never pass through LDS: a lane's A-operand fragment of the band product is 16 contiguous bytes of
//     a table row, loaded straight from global memory (L2-resident: 2 S rows of 128 B per head) into registers;
//   * a wave walks CONSECUTIVE 32-key sub-blocks (the two key groups split a block's key range in two contiguous halves
//     instead of interleaving), so the upper 32 rows of a sub-block's 64-row band are the lower 32 of the previous one:
//     kept in registers, ONE band tile per sub-block instead of two (12 instead of 16 matrix instructions per term);
//   * the two key groups run HALF A STEP APART: while group 0 is in its matrix segment (P V of sub-block n - 1, then
//     K Q^T and the band tile of sub-block n) group 1 is in its vector segment (skew, softmax, conversion of P) and
//     requests its next tiles, and vice versa -- the two waves of a SIMD (w, w + 4) alternate between the matrix pipe and
//     the VALU / LDS instead of meeting in both (MI355X guide, "Two waves per SIMD"); one block barrier per segment is the
//     hand-off of the tiles AND the phase lock;
//   * a group's tiles are requested at the start of its own vector segment and read in its next matrix segment: K and V
//     are single-buffered per group (32 KB in all for three-term products), the skew buffers take 68 KB.
// Layout per workgroup (8 waves): wave = (query subtile wq of 32 rows, key group grp); group 0 takes the first half of the
// block's 32-key sub-blocks, group 1 the second; each keeps its own (m, l, O), merged per block.
#include <algorithm>

#include "isi_common.h"
#include "isi_internal.h"
#include "prof.h"
#include "rel_attention.h"
#include "split_bf16.h"

namespace isi {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t *lds_s16x4_t;

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float NEG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int QB = 128;    // queries per block
constexpr int LD = 68;     // floats per query row of the skew buffer (64 band rows + 4)

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ s16x8_t buf_load_frag(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return __builtin_bit_cast(s16x8_t, v);
}
// a fragment load the COMPILER does not wait for (it would wait with a count that knows nothing of the DMAs in flight and
// drain them): the destination is valid only behind the kernel's own counted s_waitcnt (wait_tiles).  k-block by immediate
// offset, plane by the scalar offset: ONE address register per lane, which the caller keeps to itself for the whole loop.
template <int IMM>
__device__ __forceinline__ s16x8_t buf_load_frag_async(const i32x4 rsrc, const unsigned byte_off, const unsigned soff) {
  i32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(byte_off), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
  return __builtin_bit_cast(s16x8_t, v);
}
// one LDS-DMA: lane l's 16 bytes at rsrc.base + voff + soff + IMM land at (M0 + IMM) + 16 l (conv_pair_f16.hip has the
// form that rewrites M0 per piece).  Here M0 is written ONCE per wave, at kernel entry, and a wave's pieces are told apart by
// the instruction's immediate offset -- which the hardware adds to the memory address as well, hence `voff` carries - IMM
// (the descriptors start DMA_MARGIN bytes early so that it stays non-negative).  Found the hard way: with `s_mov_b32 m0`
// 1-5 wait states in front of the `buffer_load ... lds` (the documented requirement is 1) the DMA now and then still used
// the PREVIOUS value of M0 -- a piece landed in its neighbour's slot -- whenever the vector-memory port happened to be
// free at that moment: 8-18 of 32 runs wrong at S 1025, none with 8 wait states, all of them with the 8 wait states spent
// anywhere else (profiles/r06_attention_m0_hazard.txt).  A value of M0 that never changes cannot be stale.
template <int IMM>
__device__ __forceinline__ void dma16(const unsigned voff, const i32x4 rsrc, const unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" :: "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
}
constexpr unsigned DMA_MARGIN = 4096;
__device__ __forceinline__ i32x4 make_rsrc(const void *ptr, const unsigned bytes) {
  const unsigned long long b = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// the 16-bit operand type of a mode: conversions (round to nearest even) and the matrix instruction
template <bool F16> struct Prec;
template <> struct Prec<false> {
  static __device__ __forceinline__ unsigned pack2(const float a, const float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  }
  static __device__ __forceinline__ void split2(const float a, const float b, unsigned &hi, unsigned &lo) { isi::split2(a, b, hi, lo); }
  static __device__ __forceinline__ f32x16 mfma(const s16x8_t a, const s16x8_t b, const f32x16 c) { return ISI_MFB(a, b, c); }
};
template <> struct Prec<true> {
  static __device__ __forceinline__ unsigned pack2(const float a, const float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ void split2(const float a, const float b, unsigned &hi, unsigned &lo) {
    const f32x2_t v = {a, b};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f16x2_t l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2_t), f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
  }
  static __device__ __forceinline__ f32x16 mfma(const s16x8_t a, const s16x8_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// ---- operand planes: fp32 -> 16-bit hi (+ lo) pieces, row-major [row][HD]; K / V rows = (pair, key) with the key count
// padded to a multiple of 32 (zero rows), e rows = (head, table row).  One thread = 8 consecutive dims of a row.
struct PackArgs {
  const float *k, *v, *e;
  unsigned short *k16, *v16, *e16;        // plane 0 of each; plane 1 follows kv_plane / e_plane elements later
  long long kv_plane, e_plane;            // elements per plane
  long long n_kv, n_pad, n_e;             // 8-element items: real rows of K (= of V), zero rows of K (= of V), e
  int Sk, Skp, B, H, R, HD;
  int k_ss, k_sb, k_sh, v_ss, v_sb, v_sh;
};
template <bool F16, int NPL>
__global__ __launch_bounds__(256) void attn_pack_kernel(const PackArgs p) {
  using PR = Prec<F16>;
  const int g8 = p.HD >> 3;
  const long long total = 2 * (p.n_kv + p.n_pad) + p.n_e;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float *src = nullptr;
    unsigned short *dst;
    long long plane;
    if (i < 2 * p.n_kv) {
      // real rows, in the SOURCE's order (key, batch, head, 8-dim group): a wave reads 2 KiB of one (key, batch) row of the
      // projection's output and writes 128-byte pieces of eight heads' planes (pair-major on the reading side it gathered
      // 256-byte pieces 48 KiB apart)
      const bool is_v = i >= p.n_kv;
      long long it = is_v ? i - p.n_kv : i;
      const int g = (int)(it % g8); it /= g8;
      const int h = (int)(it % p.H); it /= p.H;
      const int b = (int)(it % p.B);
      const int s = (int)(it / p.B);
      src = is_v ? p.v + ((long long)s * p.v_ss + (long long)b * p.v_sb + (long long)h * p.v_sh + 8 * g)
                 : p.k + ((long long)s * p.k_ss + (long long)b * p.k_sb + (long long)h * p.k_sh + 8 * g);
      dst = (is_v ? p.v16 : p.k16) + (((long long)(b * p.H + h) * p.Skp + s) * g8 + g) * 8;
      plane = p.kv_plane;
    } else if (i < 2 * (p.n_kv + p.n_pad)) {     // the zero rows Sk .. Skp - 1 of every pair
      long long it = i - 2 * p.n_kv;
      const bool is_v = it >= p.n_pad;
      if (is_v) it -= p.n_pad;
      const int g = (int)(it % g8); it /= g8;
      const int npad = p.Skp - p.Sk;
      const int s = p.Sk + (int)(it % npad);
      const long long pair = it / npad;
      dst = (is_v ? p.v16 : p.k16) + ((pair * p.Skp + s) * g8 + g) * 8;
      plane = p.kv_plane;
    } else {      // e: fragment-major, [head][16-byte piece c = 2 t + half of the row][row][8]: the 32 rows of a band tile's
                  // fragment are 512 contiguous bytes per half-wave whatever the tile's first row (row-major, a lane's 16
                  // bytes sat 128 bytes from its neighbour's: ~80 cycles of issue per load, tools/stamps_fwd3.py)
      const long long it = i - 2 * (p.n_kv + p.n_pad);
      const long long row = it / g8;                       // (head, table row)
      const int c = (int)(it - row * g8);
      const long long hh = row / p.R, r = row - hh * p.R;
      src = p.e + it * 8;
      dst = p.e16 + ((hh * g8 + c) * p.R + r) * 8;
      plane = p.e_plane;
    }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (src) { a = *reinterpret_cast<const float4 *>(src); c = *reinterpret_cast<const float4 *>(src + 4); }
    uint4 hi, lo;
    if constexpr (NPL == 1) {
      hi = make_uint4(PR::pack2(a.x, a.y), PR::pack2(a.z, a.w), PR::pack2(c.x, c.y), PR::pack2(c.z, c.w));
    } else {
      PR::split2(a.x, a.y, hi.x, lo.x); PR::split2(a.z, a.w, hi.y, lo.y);
      PR::split2(c.x, c.y, hi.z, lo.z); PR::split2(c.z, c.w, hi.w, lo.w);
      *reinterpret_cast<uint4 *>(dst + plane) = lo;
    }
    *reinterpret_cast<uint4 *>(dst) = hi;
  }
}

struct Fwd3Args {
  const unsigned short *k16, *v16, *e16;
  unsigned kv_bytes, e_bytes;             // all planes of K (= of V), all planes of e
  unsigned kv_plane_bytes, e_plane_bytes;
  int Skp;
};

template <int HD, int TERMS> struct Tile3 {
  static constexpr int NPL = TERMS == 1 ? 1 : 2;
  static constexpr int ROWB = HD * 2;                       // bytes of a tile row
  static constexpr int TILEB = 32 * ROWB;                   // one plane of a 32-key tile
  static constexpr int PPT = TILEB / 1024;                  // 1-KiB DMA pieces per plane of a tile
  static constexpr int PT = NPL * PPT;                      // ... per tile
  static constexpr int NPW = (PT + 3) / 4;                  // ... per wave (the four waves of a group share a tile)
  // LDS: every wave owns one contiguous slot of two STAGES [K pieces NPW | V pieces NPW]: piece pid of a group's K tile
  // of sub-block n lives in stage n & 1 of wave pid % 4's slot at (pid / 4) KiB; its V tile in stage (n + 1) & 1, NPW KiB
  // further -- so that the two tiles a vector segment requests (K of n + 2, V of n + 1) go to ONE stage, under one M0
  static constexpr int STAGEB = 2 * NPW * 1024;
  static constexpr int WAVEB = 2 * STAGEB;
  static constexpr size_t kv_bytes = (size_t)8 * WAVEB;
  static constexpr size_t smem = kv_bytes + (size_t)(8 * 32 * LD) * sizeof(float);
};
}  // namespace

#ifdef ISI_MEASURE
__device__ long long g_fwd3_stamps[512];
#define ISI_F3_STAMP(i_) do { if (blockIdx.x == 0 && it == 0 && (wave & 3) == 0 && lane == 0 && (i_) < 256) \
    g_fwd3_stamps[(wave >> 2) * 256 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_F3_STAMP(i_) do { } while (0)
#endif

template <int HD, int TERMS, bool F16>
__global__ __launch_bounds__(512) void rel_attn_fwd3_kernel(const AttnKArgs p, const Fwd3Args x, const int nW) {
  using PR = Prec<F16>;
  using TL = Tile3<HD, TERMS>;
  constexpr bool ONE = TERMS == 1;
  constexpr int NPL = TL::NPL, ROWB = TL::ROWB, PPT = TL::PPT, PT = TL::PT, NPW = TL::NPW, STAGEB = TL::STAGEB, WAVEB = TL::WAVEB;
  constexpr int NKB = HD / 16;           // 16-deep k-blocks of the head dim
  constexpr int NDB = HD / 32;           // 32-row blocks of O^T
  constexpr int PPR = HD / 8;            // 16-byte pieces per tile row
  constexpr int KPP = 64 / PPR;          // keys per 1-KiB DMA
  constexpr int NE = NKB * NPL;          // loads of one band tile's fragments
  static_assert(HD == 64 || HD == 32, "head dim");
  static_assert(STAGEB <= 4096 && (2 * NPW - 1) * 1024 <= (int)DMA_MARGIN, "immediate offsets of a stage's pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char *KV = reinterpret_cast<unsigned char *>(smem);        // [wave 8][stage 2][K pieces NPW | V pieces NPW][1 KiB]
  float *Sr = reinterpret_cast<float *>(KV + 8 * WAVEB);              // [8][32][LD]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  int w, pair;
  if (!xcd_tile(nW, p.H * p.B, false, w, pair)) return;
  const int h = pair % p.H, b = pair / p.H;
  const int nqb = p.nblk;
  const int rag = (p.mask_mode == 1 && nqb * QB >= p.Sq) ? p.Sq % QB : 0;   // (see rel_attention_fwd2.hip)

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x.e16), 0, x.e_bytes, 0x00020000);
  const i32x4 re_a = make_rsrc(x.e16, x.e_bytes);
  const i32x4 rk = make_rsrc(reinterpret_cast<const unsigned char *>(x.k16) - DMA_MARGIN, x.kv_bytes + DMA_MARGIN);
  const i32x4 rv = make_rsrc(reinterpret_cast<const unsigned char *>(x.v16) - DMA_MARGIN, x.kv_bytes + DMA_MARGIN);
  // M0 = the stage of this wave's slot the NEXT requests go to; written once per step, a segment ahead of its use
  const unsigned m0_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)KV + (unsigned)(wave * WAVEB);
  auto set_m0 = [&](const int stage) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 7" :: "s"(m0_base + (unsigned)(stage * STAGEB)) : "memory"); };

  const float qscale = p.scale * LOG2E;
  float *sr = Sr + wave * 32 * LD + ql * LD;
  const bool dma_wave = wq < PT;          // (head_dim 32, single-term: a tile is two pieces, waves 2 and 3 of a group request none)

  // ---- this lane's share of the group's DMAs: piece pid = wq + 4 j of a tile = plane pid / PPT, keys KPP (pid % PPT) ..
  unsigned dma_voff_k[NPW], dma_voff_v[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int pid = wq + 4 * j, pl = pid / PPT, i = pid % PPT;
    const int key = KPP * i + lane / PPR, pos = lane % PPR;
    const int swk = HD == 64 ? (key >> 1) & 7 : (key >> 2) & 3;
    const int swv = HD == 64 ? ((key >> 1) & 1) << 2 : 0;
    dma_voff_k[j] = DMA_MARGIN - (unsigned)(j * 1024) + (unsigned)(pl * x.kv_plane_bytes + key * ROWB + ((pos ^ swk) << 4));
    dma_voff_v[j] = DMA_MARGIN - (unsigned)((NPW + j) * 1024) + (unsigned)(pl * x.kv_plane_bytes + key * ROWB + ((pos ^ swv) << 4));
  }
  const unsigned pair_row0 = (unsigned)pair * (unsigned)x.Skp;     // first row of this pair in a plane
  auto issue_k = [&](const int k0) {     // into the stage M0 points at
    const unsigned soff = (pair_row0 + (unsigned)k0) * (unsigned)ROWB;
    if (dma_wave) {
      dma16<0>(dma_voff_k[0], rk, soff);
      if constexpr (NPW > 1) dma16<1024>(dma_voff_k[NPW - 1], rk, soff);
    }
  };
  auto issue_v = [&](const int k0) {
    const unsigned soff = (pair_row0 + (unsigned)k0) * (unsigned)ROWB;
    if (dma_wave) {
      dma16<NPW * 1024>(dma_voff_v[0], rv, soff);
      if constexpr (NPW > 1) dma16<(NPW + 1) * 1024>(dma_voff_v[NPW - 1], rv, soff);
    }
  };
  // Requests in flight per wave, oldest first, when a vector segment ends: [the DMAs of the previous segment] [this segment's
  // band fragments for the next sub-block] [this segment's DMAs] (+ stores of kept logits, which may overtake loads, never
  // the other way round): the counted wait lets the last group fly on -- never vmcnt(0) in the loop.  (With stores in
  // flight the same count merely waits for more.)  No vector-memory request is issued in a matrix segment: the four
  // fragment loads there queued behind the other group's DMA burst and held the wave's matrix instructions back with them.
  auto wait_tiles = [&]() {
    if (dma_wave) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- fragment addresses.  Piece (plane pl, index i) of this group's K tile sits in the slot of wave (pl PPT + i) % 4 at
  // (pl PPT + i) / 4 KiB, its V twin NPW KiB further: for both head dims the plane and the 8-key (16-key) index enter as constants
  constexpr int PLANE_OFF = HD == 64 ? 1024 : 2 * WAVEB;       // plane 1 - plane 0 of the same keys
  constexpr int V_WHICH = HD == 64 ? WAVEB : 8 * ROWB;         // keys + 8
  constexpr int V_T = HD == 64 ? 2 * WAVEB : WAVEB;            // keys + 16
  // K: lane (key ql, half), k-block t -> piece 2 t + half of row ql
  const int swk_l = HD == 64 ? (ql >> 1) & 7 : (ql >> 2) & 3;
  const unsigned char *Kb = KV + (grp * 4 + ql / KPP) * WAVEB + (ql % KPP) * ROWB;
  // V^T via transposing reads: 16-lane group (i = lane & 15, g16 = (lane >> 4) & 1): lane i points at key row
  // 16 t + 8 which + 4 half + (i >> 2), dims db 32 + 16 g16 + 4 (i & 3) .. + 3 and receives dims db 32 + 16 g16 + i of the four
  // keys 16 t + 8 which + 4 half + 0 .. 3 -- k-slots (half, 4 which + 0 .. 3) of block t, as P's accumulator layout has them
  const int vi = lane & 15, g16 = (lane >> 4) & 1;
  const int vflip = HD == 64 ? (vi >> 3) & 1 : 0;
  const unsigned char *Vb[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
    Vb[d] = KV + grp * 4 * WAVEB + NPW * 1024 + (4 * half + (vi >> 2)) * ROWB + (((d ^ vflip) * 4 + 2 * g16 + ((vi & 3) >> 1)) << 4) + 8 * (vi & 1);
  auto vfrag = [](const unsigned a) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(size_t)a);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(size_t)(a + V_WHICH));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  unsigned Va[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) Va[d] = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)Vb[d];
  // e: A-operand fragment of band row `row` (lane = row), k-block t, plane pl: piece 2 t + half of the table row, in the
  // fragment-major layout [head][piece][row][16 B] (attn_pack_kernel)
  const unsigned e_lane = ((unsigned)(h * PPR + half) * (unsigned)p.R) * 16u;
  const unsigned e_tstep = 2u * (unsigned)p.R * 16u;       // k-block t -> t + 1

  struct Item { int q0, q_end, k_begin, k_end; bool valid; };
  auto item_of = [&](int it) {
    Item c;
    const int ps = it * nW + ((it & 1) ? nW - 1 - w : w);
    c.valid = ps < nqb;
    const int qblk = p.mask_mode == 1 ? nqb - 1 - ps : ps;
    c.q0 = rag ? (qblk ? rag + (qblk - 1) * QB : 0) : qblk * QB;
    c.q_end = (rag && qblk == 0) ? rag : min(p.Sq, c.q0 + QB);
    c.k_begin = p.mask_mode == 2 ? (c.q0 / 32) * 32 : 0;
    c.k_end = p.mask_mode == 1 ? min(p.Sk, c.q_end) : p.Sk;
    return c;
  };

  if (grp == 1) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses issue arbitration otherwise (MI355X guide)
  // ---- what a block needs before its first step: Q, the K tiles of its sub-blocks 0 and 1, the V tile of sub-block 0, the
  // first band fragments
  struct Geom { int q0, q_end, k_end, qw0, qi, sb_end, g_first, g_count, nsteps; };
  auto geom_of = [&](const Item &c) {
    Geom g;
    g.q0 = c.q0; g.q_end = c.q_end; g.k_end = c.k_end;
    g.qw0 = c.q0 + 32 * wq; g.qi = g.qw0 + ql;
    const int sb_begin = c.k_begin / 32;
    g.sb_end = (c.k_end + 31) / 32;
    const int nsb = g.sb_end - sb_begin, n0 = (nsb + 1) / 2;
    g.g_first = grp ? sb_begin + n0 : sb_begin;      // this group's sub-blocks: the first half of the block's, or the second
    g.g_count = grp ? nsb - n0 : n0;
    g.nsteps = n0;
    return g;
  };
  float4 qa[NKB], qc[NKB];
  s16x8_t en[NKB][NPL];                        // band fragments of the coming sub-block's NEW tile
  unsigned e_voff;
  // The asynchronous fragment loads: one address register, k-block and plane in the scalar offset (out-of-range lanes stay
  // out of range: the check looks at the vector offset alone).  Valid behind wait_tiles / vmcnt(0) only.
  auto load_e_async = [&](const int wrow) {
    const int r = wrow + ql;
    e_voff = (r >= 0 && r < p.R) ? e_lane + (unsigned)r * 16u : 0x7FFFFF00u;
#pragma unroll
    for (int t = 0; t < NKB; ++t)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        en[t][pl] = buf_load_frag_async<0>(re_a, e_voff, e_tstep * t + pl * x.e_plane_bytes);
  };
  auto request_block = [&](const Geom &g) {
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const unsigned off = g.qi < g.q_end ? (unsigned)(g.qi * p.q_ss + b * p.q_sb + h * p.q_sh + 16 * t + 8 * half) * 4u : OOB;
      qa[t] = buf_load4(rq, off);
      qc[t] = buf_load4(rq, off == OOB ? OOB : off + 16u);
    }
    // K of sub-blocks 0 and 1, V of sub-block 0 (stage parities: see Tile3); beyond the group's range the block's last one
    set_m0(0);
    issue_k(32 * min(g.g_first, g.sb_end - 1));
    set_m0(1);
    issue_k(32 * min(g.g_first + 1, g.sb_end - 1));
    issue_v(32 * min(g.g_first, g.sb_end - 1));
    load_e_async(g.qw0 - (32 * g.g_first + 31) + p.Ek - 1);
  };

  // workgroup barrier for LDS traffic only (__syncthreads would also wait for the next block's requests in flight)
  auto lds_sync = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  Item cur = item_of(0);
  for (int it = 0; cur.valid; ++it) {
    ISI_F3_STAMP(0);
    const Geom G = geom_of(cur);
    request_block(G);
    const int q_end = G.q_end, k_end = G.k_end, qw0 = G.qw0, qi = G.qi, sb_end = G.sb_end;
    const int g_first = G.g_first, g_count = G.g_count, nsteps = G.nsteps;
    auto live_at = [&](int s) {
      if (s < 0 || s >= g_count || qw0 >= q_end) return false;
      const int k0 = 32 * (g_first + s);
      bool lv = k0 < k_end;
      if (p.mask_mode == 1) lv = lv && k0 <= min(qw0 + 31, q_end - 1);
      if (p.mask_mode == 2) lv = lv && k0 + 31 >= qw0;
      return lv;
    };
    // first key of this group's sub-block s; beyond the group's range the last sub-block of the block again (requests are
    // unconditional -- the counted waits need the same number of them on every step -- and must stay inside the planes)
    auto k0_of = [&](int s) { return 32 * min(g_first + s, sb_end - 1); };
    // lowest table row of the 32-row band tile a sub-block at k0 adds (its band is rows wrow .. wrow + 63)
    auto wrow_of = [&](int s) { return qw0 - (32 * (g_first + s) + 31) + p.Ek - 1; };
    auto load_e = [&](s16x8_t (&f)[NKB][NPL], const int wrow) {
      const int r = wrow + ql;
      const unsigned base = (r >= 0 && r < p.R) ? e_lane + (unsigned)r * 16u : OOB;
#pragma unroll
      for (int t = 0; t < NKB; ++t)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          f[t][pl] = buf_load_frag(re, base == OOB ? OOB : base + e_tstep * t + pl * x.e_plane_bytes);
    };
    // ---- Q fragment of this lane's query, scaled and split once: k-block t holds dims 16 t + 8 half + 0..7
    s16x8_t qh[NKB], qlo[NKB];
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const float4 a = qa[t], c = qc[t];
      unsigned hh[4], ll[4];
      if constexpr (ONE) {
        hh[0] = PR::pack2(a.x * qscale, a.y * qscale); hh[1] = PR::pack2(a.z * qscale, a.w * qscale);
        hh[2] = PR::pack2(c.x * qscale, c.y * qscale); hh[3] = PR::pack2(c.z * qscale, c.w * qscale);
        ll[0] = ll[1] = ll[2] = ll[3] = 0u;
      } else {
        PR::split2(a.x * qscale, a.y * qscale, hh[0], ll[0]); PR::split2(a.z * qscale, a.w * qscale, hh[1], ll[1]);
        PR::split2(c.x * qscale, c.y * qscale, hh[2], ll[2]); PR::split2(c.z * qscale, c.w * qscale, hh[3], ll[3]);
      }
      qh[t] = __builtin_bit_cast(s16x8_t, make_uint4(hh[0], hh[1], hh[2], hh[3]));
      qlo[t] = __builtin_bit_cast(s16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));   // dead when ONE
    }
    // the row of this lane's query in the kept logits (training: the backward reads them back), its first key column
    float *lbase = p.logits ? p.logits + (((size_t)b * p.H + h) * p.Sq + min(qi, p.Sq - 1)) * p.ldl + 4 * half : nullptr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ISI_F3_STAMP(1);

    float m_run = NEG, l_run = 0.f;     // l_run: this lane's 16 keys per sub-block only (the halves are added at the end)
    f32x16 O[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
    f32x16 sacc, rnew, rprev;           // logits of the sub-block in flight; its band tile; the tile above it (the previous one's)
    s16x8_t ph[2], pl_[2];              // P of the sub-block whose P V is pending
    float alpha = 1.f;
    bool have_prev = false;             // rprev holds the band tile [wrow + 32, wrow + 64) of the coming sub-block

    // ---- matrix segment of step s: P V of sub-block s - 1, K Q^T and the band tile of sub-block s.  All LDS fragments are
    // requested FIRST (hipcc waited for each pair of reads right in front of its matrix instruction: ~1000 of the
    // segment's 1950 cycles were exposed LDS latency, tools/stamps_fwd3.py), the rescaling of O runs under their flight
    auto matrix_segment = [&](const int s, const bool pv_live, const bool qk_live) {
      const int st = (s & 1) * STAGEB;
      [[maybe_unused]] const int sbase = 8 + 8 * s;
      s16x8_t vf[NDB][2][NPL], kf[NKB][NPL];
      unsigned va[NDB];
#pragma unroll
      for (int d = 0; d < NDB; ++d) va[d] = Va[d] + (unsigned)st;
      // order: every LDS fragment requested first; the band tile (operands in registers: nothing to wait for) runs under
      // their flight, and its fragment registers are re-requested for the next sub-block right behind it; then K Q^T; then
      // the rescaling of O and P V
#ifndef F3KA
#define F3KA (NKB / 2)
#endif
#ifndef F3VA
#define F3VA 0
#endif
      constexpr int KA = ONE ? NKB : F3KA;   // K fragments requested up front (three-term: the rest behind the band tile)
      if (qk_live) {
#pragma unroll
        for (int t = 0; t < KA; ++t)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            kf[t][pl] = *reinterpret_cast<const s16x8_t *>(Kb + st + pl * PLANE_OFF + (((2 * t + half) ^ swk_l) << 4));
      }
      constexpr int VA = ONE ? 2 : F3VA;     // V fragments (key blocks t) requested up front (three-term: behind K Q^T)
      if (pv_live) {
#pragma unroll
        for (int t = 0; t < VA; ++t)
#pragma unroll
          for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) vf[d][t][pl] = vfrag(va[d] + pl * PLANE_OFF + t * V_T);
      }
      // the skew buffer is WRITTEN here, under the matrix instructions, and only read back in the vector segment (its eight
      // 1-KiB stores per wave used to be the first thing behind the barrier, together with the other group's fragment reads)
      auto put_tile = [&](const f32x16 &tile, const int col0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(sr + col0 + 8 * g + 4 * half) = make_float4(tile[4 * g], tile[4 * g + 1], tile[4 * g + 2], tile[4 * g + 3]);
      };
      if constexpr (ONE) {
        // single-term products: one matrix instruction per k-block and chain.  The band tile is summed in TWO accumulators
        // (even / odd k-blocks, added at the end); K Q^T merged instruction by instruction with the two chains of P V was
        // measured and dropped (it needs every fragment live at once: 256 registers + scratch, 58 -> 74 us)
        if (qk_live) {
          f32x16 rodd;
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] = rnew[r] = rodd[r] = 0.f;
#pragma unroll
          for (int t = 0; t < NKB; t += 2) {
            rnew = PR::mfma(en[t][0], qh[t], rnew);
            rodd = PR::mfma(en[t + 1][0], qh[t + 1], rodd);
          }
          if (!have_prev) {    // wave-uniform, once per wave and block: the upper band tile through the same fragment registers
            load_e(en, wrow_of(s) + 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) rprev[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NKB; ++t) rprev = PR::mfma(en[t][0], qh[t], rprev);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) rnew[r] += rodd[r];
        }
        ISI_F3_STAMP(sbase + 4);
        if (qk_live) {
#pragma unroll
          for (int t = 0; t < NKB; ++t) sacc = PR::mfma(kf[t][0], qh[t], sacc);
        }
        if (pv_live) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < NDB; ++d) O[d] = PR::mfma(vf[d][t][0], ph[t], O[d]);
        }
        if (qk_live) {
          put_tile(rprev, 32);
          put_tile(rnew, 0);
          rprev = rnew;                  // the next sub-block's upper band tile
          have_prev = true;
        }
        ISI_F3_STAMP(sbase + 5);
      } else {
        // three-term products: the three instructions of a product follow each other on one accumulator (forwarded); the
        // register budget (242 of 256) leaves no second band accumulator and stages the fragment requests
        if (qk_live) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] = rnew[r] = 0.f;
#pragma unroll
          for (int t = 0; t < NKB; ++t) {
            rnew = PR::mfma(en[t][1], qh[t], rnew);
            rnew = PR::mfma(en[t][0], qlo[t], rnew);
            rnew = PR::mfma(en[t][0], qh[t], rnew);
          }
          if (!have_prev) {    // wave-uniform, once per wave and block: the upper band tile through the same fragment registers
            load_e(en, wrow_of(s) + 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) rprev[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NKB; ++t) {
              rprev = PR::mfma(en[t][1], qh[t], rprev);
              rprev = PR::mfma(en[t][0], qlo[t], rprev);
              rprev = PR::mfma(en[t][0], qh[t], rprev);
            }
          }
        }
        ISI_F3_STAMP(sbase + 4);
        if (qk_live) {
          if constexpr (KA < NKB) {
#pragma unroll
            for (int t = KA; t < NKB; ++t)
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl)
                kf[t][pl] = *reinterpret_cast<const s16x8_t *>(Kb + st + pl * PLANE_OFF + (((2 * t + half) ^ swk_l) << 4));
          }
#pragma unroll
          for (int t = 0; t < NKB; ++t) {
            sacc = PR::mfma(kf[t][NPL - 1], qh[t], sacc);
            sacc = PR::mfma(kf[t][0], qlo[t], sacc);
            sacc = PR::mfma(kf[t][0], qh[t], sacc);
          }
          put_tile(rprev, 32);
          put_tile(rnew, 0);
          rprev = rnew;                  // the next sub-block's upper band tile
          have_prev = true;
        }
        ISI_F3_STAMP(sbase + 5);
        if (pv_live) {
          if constexpr (VA < 2) {
#pragma unroll
            for (int t = VA; t < 2; ++t)
#pragma unroll
              for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) vf[d][t][pl] = vfrag(va[d] + pl * PLANE_OFF + t * V_T);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
              O[d] = PR::mfma(vf[d][t][NPL - 1], ph[t], O[d]);
              O[d] = PR::mfma(vf[d][t][0], pl_[t], O[d]);
              O[d] = PR::mfma(vf[d][t][0], ph[t], O[d]);
            }
        }
      }
    };
    // the skewed read-back of sub-block s: sixteen explicit ds_read_b32, issued at the head of the vector segment; the
    // segment's vector-memory requests go out under their flight, ONE full lgkmcnt drain follows (softmax).  What hipcc makes
    // of the plain C++ form -- ds_read2_b32 pairs whose destination pair starts at their own address register, counted
    // lgkmcnt waits, v_pk_add_f32 with op_sel straight behind them -- now and then delivered a STALE upper register to the
    // add for lanes 48-63 (the band term of one key missing for 16 queries: 3-29 of 32 runs wrong at B 8 x H 8 x S 1025,
    // worst right behind the block barrier; profiles/r06_attention_skew_race.txt); this form: 0 of 300.
    float rl[16];
    auto skew_reads = [&]() {
      const float *rd = sr + ql + 31 - 4 * half;     // band row ql + 31 - (key row in the sub-block)
      const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)(rd - 27);
#define F3_RD(r_) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rl[r_]) : "v"(ra), "n"(4 * (27 - (((r_) & 3) + 8 * ((r_) >> 2)))) : "memory")
      F3_RD(0); F3_RD(1); F3_RD(2); F3_RD(3); F3_RD(4); F3_RD(5); F3_RD(6); F3_RD(7);
      F3_RD(8); F3_RD(9); F3_RD(10); F3_RD(11); F3_RD(12); F3_RD(13); F3_RD(14); F3_RD(15);
#undef F3_RD
    };
    // ---- mask, online softmax of sub-block s (vector segment): sacc + skewed band -> P operands, alpha
    auto softmax = [&](const int s) {
      const int k0 = 32 * (g_first + s);
      __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0), as the builtin: hipcc's own count of LDS operations in flight restarts at zero here
      asm volatile("" ::: "memory");
      ISI_F3_STAMP(8 + 8 * s + 6);
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] += rl[r];
      float sv[16];
      float tmax;
      float *lrow = lbase ? lbase + k0 : nullptr;
      bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < q_end;
      if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
      if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
      float psum = 0.f;
      if (full) {
        if (lrow) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(lrow + 8 * g) = make_float4(sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]);
        }
        tmax = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[r]);
        tmax = fmaxf(tmax, xor32_f32(tmax));
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sv[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new);
          psum += sv[r];
        }
        m_run = m_new;
      } else {
        tmax = NEG;
        const int khi = min(p.Sk - 1, p.mask_mode == 1 ? qi : 0x7fffffff) - k0 - 4 * half;
        const int klo = (p.mask_mode == 2 ? qi : 0) - k0 - 4 * half;
        if (p.mask) {      // additive mask tensor (masks the wrapper does not recognise as causal / anti-causal)
          const float *mrow = p.mask + (size_t)min(qi, p.Sq - 1) * p.Sk + k0 + 4 * half;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2);
            const bool ok = jj <= khi && jj >= klo;
            const float sc = ok ? sacc[r] + mrow[ok ? jj : 0] * LOG2E : NEG;
            sv[r] = sc;
            tmax = fmaxf(tmax, sc);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2);
            const float sc = (jj <= khi && jj >= klo) ? sacc[r] : NEG;
            sv[r] = sc;
            tmax = fmaxf(tmax, sc);
          }
        }
        if (lrow && qi < q_end) {    // (masked pairs carry -1e30; keys beyond Sk fall into the row's padding)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(lrow + 8 * g) = make_float4(sv[4 * g], sv[4 * g + 1], sv[4 * g + 2], sv[4 * g + 3]);
        }
        tmax = fmaxf(tmax, xor32_f32(tmax));
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr_ = sv[r] <= -1e29f ? 0.f : __builtin_amdgcn_exp2f(sv[r] - m_new);
          sv[r] = pr_;
          psum += pr_;
        }
        m_run = m_new;
      }
      l_run = l_run * alpha + psum;
      // O is rescaled HERE, in the vector segment (the matrix segment is the longer one of the two)
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
      // P as MFMA B operand: key block t = registers 8 t .. 8 t + 7
      if constexpr (ONE) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          ph[t] = __builtin_bit_cast(s16x8_t, make_uint4(PR::pack2(sv[8 * t], sv[8 * t + 1]), PR::pack2(sv[8 * t + 2], sv[8 * t + 3]),
                                                         PR::pack2(sv[8 * t + 4], sv[8 * t + 5]), PR::pack2(sv[8 * t + 6], sv[8 * t + 7])));
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          unsigned hh[4], ll[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) PR::split2(sv[8 * t + 2 * e], sv[8 * t + 2 * e + 1], hh[e], ll[e]);
          ph[t] = __builtin_bit_cast(s16x8_t, make_uint4(hh[0], hh[1], hh[2], hh[3]));
          pl_[t] = __builtin_bit_cast(s16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));
        }
      }
    };

    // ---- the segments.  Group 1 runs one segment behind group 0 (one extra barrier here, one for group 0 at the end).
    if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    for (int s = 0; s < nsteps; ++s) {
      [[maybe_unused]] const int sbase = 8 + 8 * s;
      ISI_F3_STAMP(sbase);
      set_m0(s & 1);            // for the requests of this step's vector segment
      const bool lv = live_at(s);
      if (!lv) have_prev = false;
      matrix_segment(s, live_at(s - 1), lv);
      // gfx950 counts at most 15 LDS operations in flight per wave and hipcc keeps its own tally: with the tile stores of one
      // step still on it, it put `s_waitcnt lgkmcnt(3 / 7)` in front of EVERY fragment read of the next -- a dozen exposed
      // LDS round trips, ~1000 cycles per matrix segment (tools/stamps_fwd3.py).  Drained here, visibly to the compiler.
      __builtin_amdgcn_s_waitcnt(0xc07f);
      ISI_F3_STAMP(sbase + 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ISI_F3_STAMP(sbase + 2);
      // vector segment: V of sub-block s + 1 and K of sub-block s + 2 into the stage the matrix segment has just read
      if (lv) skew_reads();
      load_e_async(wrow_of(s + 1));      // (on every step: the counted wait below needs the same requests each time)
      issue_v(k0_of(s + 1));
      issue_k(k0_of(s + 2));
      if (lv) softmax(s);
      ISI_F3_STAMP(sbase + 7);
      wait_tiles();
      ISI_F3_STAMP(sbase + 3);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    matrix_segment(nsteps, live_at(nsteps - 1), false);
    if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

    ISI_F3_STAMP(240);
    // ---- merge the two groups' softmax states (group 1 -> LDS -> group 0); the skew buffers are free now
    float *mg = Sr;
    constexpr int MGW = (NDB * 16 + 2) * 64;
    static_assert((size_t)4 * MGW <= (size_t)8 * 32 * LD, "merge buffer must fit in the skew buffers");
    __syncthreads();
    // (requesting the NEXT block's Q and first tiles here, under the merge and the stores, was measured: the prologue left
    // the timeline, but Q and the fragments then live across the merge -- 256 registers + scratch in the three-term kernel
    // (91 -> 97 us), nothing gained in the single-term one (60.5 -> 60.8 us))
    const Item nxt = item_of(it + 1);
    if (grp == 1) {
      float *dst = mg + wq * MGW + lane;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = O[d][r];
      dst[NDB * 16 * 64] = m_run;
      dst[(NDB * 16 + 1) * 64] = l_run;
    }
    lds_sync();
    if (grp == 0) {
      const float *src = mg + wq * MGW + lane;
      const float m1 = src[NDB * 16 * 64], l1 = src[(NDB * 16 + 1) * 64];
      const float m = fmaxf(m_run, m1);
      const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
      l_run = l_run * a0 + l1 * a1;
      l_run += xor32_f32(l_run);
      m_run = m;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] = O[d][r] * a0 + src[(d * 16 + r) * 64] * a1;
    }
    lds_sync();                 // the merge buffer is free again
    ISI_F3_STAMP(241);
    if (grp == 0 && qi < q_end) {
      const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
      float *orow = p.out + (size_t)qi * p.o_ss + (size_t)b * p.o_sb + (size_t)h * p.o_sh;
      if (p.lse && half == 0)  // natural-log domain
        p.lse[((size_t)b * p.H + h) * p.Sq + qi] = l_run > 0.f ? m_run * LN2 + logf(l_run) : 1e30f;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * half;
          *reinterpret_cast<float4 *>(orow + dd) =
              make_float4(O[d][4 * g] * inv, O[d][4 * g + 1] * inv, O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv);
        }
    }
    ISI_F3_STAMP(242);
    cur = nxt;
  }
}

int rel_attention_fwd3_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd3_stamps), sizeof(long long) * (size_t)(n < 512 ? n : 512)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

namespace {
struct PlaneLayout { int Skp, npl; size_t kv_plane, e_plane, k_off, v_off, e_off, total; };   // element counts / byte offsets
PlaneLayout plane_layout(const AttnKArgs &a, int head_dim, int precision) {
  PlaneLayout L;
  L.Skp = (a.Sk + 31) & ~31;
  L.npl = precision == 1 ? 2 : 1;
  L.kv_plane = (size_t)a.H * a.B * L.Skp * head_dim;
  L.e_plane = a.e ? (size_t)a.H * a.R * head_dim : 0;
  L.k_off = 0;
  L.v_off = round_up(L.k_off + L.npl * L.kv_plane * 2, 256);
  L.e_off = round_up(L.v_off + L.npl * L.kv_plane * 2, 256);
  L.total = round_up(L.e_off + L.npl * L.e_plane * 2, 256);
  return L;
}
}  // namespace

bool rel_attention_fwd3_ok(const AttnKArgs &a, int head_dim, int precision) {
  if (!(a.Cq == 1 && a.Ck == 1) || !(head_dim == 64 || head_dim == 32) || precision < 1 || precision > 3 || !a.e) return false;
  const PlaneLayout L = plane_layout(a, head_dim, precision);
  // one buffer descriptor per operand over all its planes (32-bit offsets)
  return L.npl * L.kv_plane * 2 < ((size_t)1 << 31) && L.npl * L.e_plane * 2 < ((size_t)1 << 31);
}

size_t rel_attention_fwd3_workspace_bytes(const AttnKArgs &a, int head_dim, int precision) {
  return rel_attention_fwd3_ok(a, head_dim, precision) ? plane_layout(a, head_dim, precision).total : 0;
}

namespace {
template <int HD, int TERMS, bool F16>
int launch_fwd3_t(const AttnKArgs &a, const PlaneLayout &L, void *ws, hipStream_t stream) {
  constexpr int NPL = TERMS == 1 ? 1 : 2;
  unsigned char *base = static_cast<unsigned char *>(ws);
  PackArgs pk;
  pk.k = a.k; pk.v = a.v; pk.e = a.e;
  pk.k16 = reinterpret_cast<unsigned short *>(base + L.k_off);
  pk.v16 = reinterpret_cast<unsigned short *>(base + L.v_off);
  pk.e16 = reinterpret_cast<unsigned short *>(base + L.e_off);
  pk.kv_plane = (long long)L.kv_plane; pk.e_plane = (long long)L.e_plane;
  pk.n_kv = (long long)a.Sk * a.B * a.H * (HD / 8);
  pk.n_pad = (long long)(L.Skp - a.Sk) * a.B * a.H * (HD / 8);
  pk.n_e = (long long)(L.e_plane / 8);
  pk.Sk = a.Sk; pk.Skp = L.Skp; pk.B = a.B; pk.H = a.H; pk.R = a.R; pk.HD = HD;
  pk.k_ss = a.k_ss; pk.k_sb = a.k_sb; pk.k_sh = a.k_sh; pk.v_ss = a.v_ss; pk.v_sb = a.v_sb; pk.v_sh = a.v_sh;
  {
    const long long total = 2 * (pk.n_kv + pk.n_pad) + pk.n_e;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    prof::Scope scope(prof::K_REL_ATTENTION, 0.0, 6.0 * (2.0 * L.kv_plane + L.e_plane), stream);
    ISI_PROF_LAUNCH(scope, (attn_pack_kernel<F16, NPL>), dim3(grid), dim3(256), 0, stream, pk);
    if (int rc = check_launch("attn_pack")) return rc;
  }
  Fwd3Args x;
  x.k16 = pk.k16; x.v16 = pk.v16; x.e16 = pk.e16;
  x.kv_plane_bytes = (unsigned)(L.kv_plane * 2); x.e_plane_bytes = (unsigned)(L.e_plane * 2);
  x.kv_bytes = (unsigned)(NPL * L.kv_plane * 2); x.e_bytes = (unsigned)(NPL * L.e_plane * 2);
  x.Skp = L.Skp;
  auto kern = rel_attn_fwd3_kernel<HD, TERMS, F16>;
  constexpr size_t smem = Tile3<HD, TERMS>::smem;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(rel_attn_fwd3)");
    attr_set.mark();
  }
  const int pairs = a.H * a.B, cus = current_device_cu_count();
  int nW = a.nblk;
  if ((int64_t)pairs * a.nblk > cus)
    nW = std::max(1, std::min(a.mask_mode ? (a.nblk + 1) / 2 : a.nblk, cus / pairs));
  const double npairs = (double)a.Sq * a.Sk * (a.mask_mode ? 0.5 : 1.0) * pairs;
  prof::Scope scope(prof::K_REL_ATTENTION, 2.0 * npairs * HD * (a.e ? 3 : 2),
                    4.0 * pairs * HD * (2.0 * a.Sq + 2.0 * a.Sk), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3(xcd_grid(nW, pairs)), dim3(512), smem, stream, a, x, nW);
  return check_launch("rel_attn_fwd3");
}
template <int HD>
int launch_fwd3_hd(const AttnKArgs &a, const PlaneLayout &L, int precision, void *ws, hipStream_t stream) {
  switch (precision) {
    case 1: return launch_fwd3_t<HD, 3, false>(a, L, ws, stream);
    case 2: return launch_fwd3_t<HD, 1, false>(a, L, ws, stream);
    case 3: return launch_fwd3_t<HD, 1, true>(a, L, ws, stream);
    default: return unsupported("rel_attention: precision");
  }
}
}  // namespace

int rel_attention_fwd3(const AttnKArgs &a, int head_dim, int precision, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!rel_attention_fwd3_ok(a, head_dim, precision)) return unsupported("rel_attention_fwd3: shape");
  const PlaneLayout L = plane_layout(a, head_dim, precision);
  if (!workspace || workspace_bytes < L.total || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return invalid("rel_attention: workspace too small or not 256-byte aligned (isi_rel_attention_workspace_bytes)");
  return head_dim == 64 ? launch_fwd3_hd<64>(a, L, precision, workspace, stream) : launch_fwd3_hd<32>(a, L, precision, workspace, stream);
}

}  // namespace isi
