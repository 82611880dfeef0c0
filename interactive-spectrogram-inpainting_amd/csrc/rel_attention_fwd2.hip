// Relative-position attention forward for gfx950, 16-bit matrix pipe, second generation (round 4).
//
//   logit[i,j] = ( q_i.k_j + q_i.e[h, r(i,j)] ) * scale + mask(i,j),   r(i,j) = floor(i/Cq) - floor(j/Ck) + (Ek - 1)
//   out_i      = sum_j softmax_j(logit[i,:]) v_j
//
// The operator behind the reference's VQCPCB.transformer.transformer_custom layers
// (priors/transformer.py:370-417,756-777); specification: oracle/prior_oracle.py (parity unpinned).
//
// What changed against rel_attention_split_kernel (rel_attention_f32.hip), which this kernel replaces for the
// 16-bit modes (precision 1 = three-term split-bf16, 2 = single-term bf16, 3 = single-term f16):
//   * a PERSISTENT workgroup walks several 128-query blocks of one (batch, head) pair, dealt in "snake" order over
//     the blocks sorted by cost: under a causal mask every workgroup gets the same number of key steps (the
//     B8 H8 S1025 case: 256 workgroups x 11 steps instead of 576 workgroups of 1..9 steps on 256 CUs);
//   * a step covers 2 x 64 keys (2 x 32 for three-term products at head_dim 64: the LDS budget), i.e. one pair of
//     block barriers per 128 keys instead of per 64; 32-key sub-blocks above the diagonal are skipped;
//   * the relative logits' skew buffer is written with 16-byte stores and read with immediate offsets, both
//     conflict-free (row stride 68 floats: 4 ql mod 32 for the stores, 5 ql mod 32 for the loads), straight into the
//     accumulator of the Q K^T product (no add pass);
//   * q is pre-multiplied by scale * log2(e) (no scaling pass), row sums stay per half-wave until the end, full
//     sub-blocks take a predicate-free softmax;
//   * V^T's key order inside a 16-key group is [0-3, 8-11 | 4-7, 12-15]: one ds_read_b128 per MFMA operand.
// Layout per workgroup (8 waves): wave = (query subtile wq of 32 rows, key group grp); group g takes keys
// [kp + g KT, kp + (g + 1) KT) of a step and keeps its own (m, l, O); the two states are merged per block.
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

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float NEG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int QB = 128;    // queries per block
constexpr int LD = 68;     // floats per query row of the skew buffer (64 band rows + 4)

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ float elem(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

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

template <int HD, int TERMS> struct Tile {
  static constexpr int KT = (TERMS == 3 && HD == 64) ? 32 : 64;   // keys per group and step
  static constexpr int NSB = KT / 32, KS = 2 * KT;
  static constexpr int RING = KS == 128 ? 256 : 192;               // rows of the band ring (>= 127/Cq + (KS-1)/Ck + 1)
  static constexpr int NPL = TERMS == 1 ? 1 : 2;                   // planes: hi (+ lo)
  static constexpr int VR = ((HD + 31) / 32) * 32;
  // LDS rows are padded by 16 bytes instead of swizzled: consecutive rows start 4 banks (mod 64) apart for HD = 64 /
  // KT = 64 (stride 36 dwords), so the 16-lane groups of a ds_read_b128 (lane = row, same column) touch every bank once,
  // and the k-blocks of a row are reached by immediate offsets from one address register
  // V^T rows hold 64 keys (128 bytes; at KT = 32 both groups' tiles side by side) and are swizzled instead: the staging
  // threads of a write instruction own rows 4 apart, which padding cannot spread (tools/probes/lds_conflict_probe.hip: 8-way
  // on padded rows).  16-byte slot c of row d sits at c ^ f(d), f(d) = ((d >> 2) & 7) ^ ((d & 2) << 1): the 16-lane groups
  // of ds_read_b128 (rows {0-3, 12-15, 20-27}, ...) see 16 distinct (row parity, slot) pairs, the writes are 2-way
  static constexpr int KLD = HD + 8, VLD = 64, NVTL = KS / 64;     // row strides of the [.][HD] and V^T planes (ushorts); V^T tiles
  static constexpr size_t operand_bytes = (size_t)(2 * NPL * KT * KLD + NVTL * NPL * VR * VLD + NPL * RING * KLD) * sizeof(unsigned short);
  static constexpr size_t smem = operand_bytes + (size_t)(8 * 32 * LD) * sizeof(float);
};
}  // namespace

// phase timestamps (-DISI_MEASURE builds; tools/stamps_fwd2.py): workgroup 0, waves 0 and 4, the first block's timeline
#ifdef ISI_MEASURE
__device__ long long g_fwd2_stamps[512];
#define ISI_F2_STAMP(i_) do { if (blockIdx.x == 0 && it == 0 && (wave & 3) == 0 && lane == 0 && (i_) < 256) \
    g_fwd2_stamps[(wave >> 2) * 256 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_F2_STAMP(i_) do { } while (0)
#endif
template <int HD, int TERMS, bool F16, bool UNIT>
__global__ __launch_bounds__(512) void rel_attn_fwd2_kernel(const AttnKArgs p, const int nW) {
  using PR = Prec<F16>;
  using TL = Tile<HD, TERMS>;
  constexpr bool ONE = TERMS == 1;
  constexpr int KT = TL::KT, NSB = TL::NSB, KS = TL::KS, RING = TL::RING, NPL = TL::NPL, VR = TL::VR, KLD = TL::KLD, VLD = TL::VLD;
  constexpr int NKB = HD / 16;           // 16-deep k-blocks of the head dim
  constexpr int NDB = (HD + 31) / 32;    // 32-row blocks of O^T
  constexpr int NQD = HD / 4;            // dim quads per row
  constexpr int NK = KS * NQD / 512;     // K (and band) quads per thread and step
  constexpr int NR = (RING - KS) * NQD / 512;   // quads per thread of the rest of a block's first band
  constexpr int NVT = KS * NQD / 4;      // threads that stage a 4 keys x 4 dims block of V
  static_assert(NK >= 1 && NVT <= 512 && NR * 512 == (RING - KS) * NQD, "staging roles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Kp = reinterpret_cast<unsigned short *>(smem);   // [group 2][plane][KT][KLD]
  unsigned short *Vp = Kp + 2 * NPL * KT * KLD;                    // [64-key tile][plane][VR][64 keys], slots swizzled (Tile)
  unsigned short *Ep = Vp + TL::NVTL * NPL * VR * VLD;             // [plane][RING][KLD]
  auto vswz = [](int d) { return ((d >> 2) & 7) ^ ((d & 2) << 1); };
  float *Sr = reinterpret_cast<float *>(Ep + NPL * RING * KLD);    // [8][32][LD]; columns 0 .. 63 of a row are band rows
  // (last event of the key's sub-block) - event(key) for the KS keys of a step (general Ck only): in the padding column
  // 64 of the skew buffer's rows
  auto evk_at = [&](int i) -> int & { return *reinterpret_cast<int *>(Sr + i * LD + 64); };
  auto ring_slot = [](int r) {
    if constexpr (RING == 256) return r & 255;
    else { r %= RING; return r < 0 ? r + RING : r; }
  };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: block / tile tests stay scalar
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  int w, pair;
  if (!xcd_tile(nW, p.H * p.B, false, w, pair)) return;
  const int h = pair % p.H, b = pair / p.H;
  const int nqb = p.nblk;
  const bool has_e = p.e != nullptr;
  // causal masks: the ragged block (Sq % QB rows) is block 0, where the key range is shortest (see rel_attention_f32.hip)
  const int rag = (p.mask_mode == 1 && p.Cq == 1 && nqb * QB >= p.Sq) ? p.Sq % QB : 0;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  if (VR > HD) {   // rows of V^T beyond the head dim feed zero products
    for (int i = tid; i < TL::NVTL * NPL * VR * VLD / 2; i += 512) reinterpret_cast<unsigned *>(Vp)[i] = 0u;
    __syncthreads();
  }
  if (grp == 1) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses issue arbitration otherwise (MI355X guide)
  const float qscale = p.scale * LOG2E;
  float *sr = Sr + wave * 32 * LD + ql * LD;
  const unsigned short *Kb = Kp + (grp * NPL) * KT * KLD;
  int evoff[16];
  if constexpr (!UNIT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) evoff[r] = 31 / p.Ck - mfma_row(r, half) / p.Ck;
  }
  const bool ck_regular = UNIT || (32 % p.Ck) == 0;
  const bool two_tiles = UNIT || (31 / p.Cq + 31 / p.Ck) >= 32;   // 32-row tiles of the band a sub-block can reach

  // ---- the blocks of this workgroup: snake order over the blocks sorted by cost (position ps -> workgroup w on even
  // rounds, nW - 1 - w on odd ones), heaviest block first under a causal mask
  struct Item { int q0, q_end, k_begin, k_end, evq_b0; bool valid; };
  auto item_of = [&](int it) {
    Item c;
    const int ps = it * nW + ((it & 1) ? nW - 1 - w : w);
    c.valid = ps < nqb;
    const int qblk = p.mask_mode == 1 ? nqb - 1 - ps : ps;
    c.q0 = rag ? (qblk ? rag + (qblk - 1) * QB : 0) : qblk * QB;
    c.q_end = (rag && qblk == 0) ? rag : min(p.Sq, c.q0 + QB);   // first row beyond this block's valid ones
    c.k_begin = p.mask_mode == 2 ? (c.q0 / 32) * 32 : 0;
    c.k_end = p.mask_mode == 1 ? min(p.Sk, c.q_end) : p.Sk;
    c.evq_b0 = UNIT ? c.q0 : c.q0 / p.Cq;
    return c;
  };
  // lowest table row the keys [kp, kp + KS) of a step reach from a block whose first query's event is evq_b0
  auto rb_step = [&](int kp, int evq_b0) { return evq_b0 - (UNIT ? kp + KS - 1 : (kp + KS - 1) / p.Ck) + p.Ek - 1; };

  // ---- staging: global -> registers (prefetch) -> 16-bit planes in LDS (commit)
  float4 pk[NK], pe[NK], pv[4], pr[NR > 0 ? NR : 1], qa[NKB], qc[NKB];
  // (requested in three parts -- K, band, V -- between the phases of a step: one burst of all twelve loads from all eight
  // waves parked every wave at the issue for ~2500 cycles, the vector-memory path takes 64 bytes per clock)
  auto prefetch_k = [&](int kp) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 512 * i, row = idx / NQD, qd = idx % NQD;
      const int kj = kp + row;
      pk[i] = buf_load4(rk, kj < p.Sk ? (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + qd * 4) * 4u : OOB);
    }
  };
  auto prefetch_e = [&](int kp, int evq_b0) {
    if (!has_e) return;
    const int rb = rb_step(kp, evq_b0);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 512 * i, row = idx / NQD, qd = idx % NQD;
      const int r = rb + row;
      pe[i] = buf_load4(re, r >= 0 && r < p.R ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
    }
  };
  auto prefetch_v = [&](int kp) {
    if (tid < NVT) {
      const int kg = tid / NQD, qd = tid % NQD;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kj = kp + 4 * kg + j;
        pv[j] = buf_load4(rv, kj < p.Sk ? (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + qd * 4) * 4u : OOB);
      }
    }
  };
  auto prefetch = [&](int kp, int evq_b0) { prefetch_k(kp); prefetch_e(kp, evq_b0); prefetch_v(kp); };
  auto put_row = [&](unsigned short *plane0, const int plane_stride, const int row, const int qd, const float4 v) {
    unsigned short *dst = plane0 + row * KLD + qd * 4;
    if constexpr (ONE) {
      *reinterpret_cast<uint2 *>(dst) = make_uint2(PR::pack2(v.x, v.y), PR::pack2(v.z, v.w));
    } else {
      uint2 hi, lo;
      PR::split2(v.x, v.y, hi.x, lo.x);
      PR::split2(v.z, v.w, hi.y, lo.y);
      *reinterpret_cast<uint2 *>(dst) = hi;
      *reinterpret_cast<uint2 *>(dst + plane_stride) = lo;
    }
  };
  auto commit = [&](int kp, int evq_b0) {
    const int rb = rb_step(kp, evq_b0);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 512 * i, row = idx / NQD, qd = idx % NQD;
      put_row(Kp + (row / KT) * NPL * KT * KLD, KT * KLD, row % KT, qd, pk[i]);
      if (has_e) put_row(Ep, RING * KLD, ring_slot(rb + row), qd, pe[i]);
    }
    if (tid < NVT) {   // V transposed: per dim the 4 keys of the block as one 8-byte unit
      const int kg = tid / NQD, qd = tid % NQD;
      const int tile = kg / 16, u = kg % 16;            // 64-key tile of the step, 4-key unit inside it
      const int c = 2 * (u >> 2) + (u & 1), sub = (u >> 1) & 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 4 * qd + e;
        const float a0 = elem(pv[0], e), a1 = elem(pv[1], e), a2 = elem(pv[2], e), a3 = elem(pv[3], e);
        unsigned short *dst = Vp + ((tile * NPL) * VR + d) * VLD + ((c ^ vswz(d)) * 8) + sub * 4;
        if constexpr (ONE) {
          *reinterpret_cast<uint2 *>(dst) = make_uint2(PR::pack2(a0, a1), PR::pack2(a2, a3));
        } else {
          uint2 hi, lo;
          PR::split2(a0, a1, hi.x, lo.x);
          PR::split2(a2, a3, hi.y, lo.y);
          *reinterpret_cast<uint2 *>(dst) = hi;
          *reinterpret_cast<uint2 *>(dst + VR * VLD) = lo;
        }
      }
    }
    if (!UNIT && tid < KS) evk_at(tid) = (kp + (tid | 31)) / p.Ck - (kp + tid) / p.Ck;
  };
  // what a block needs besides its first step: the rest of the band ring (RING rows from the first step's lowest one)
  // and the queries; requested while the previous block is merged and stored
  auto load_block = [&](const Item &c) {
    if (has_e && NR > 0) {
      const int rb = rb_step(c.k_begin, c.evq_b0);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int idx = KS * NQD + tid + 512 * i, row = idx / NQD, qd = idx % NQD, r = rb + row;
        pr[i] = buf_load4(re, r >= 0 && r < p.R ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
    const int qi = c.q0 + 32 * wq + ql;
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const unsigned off = qi < c.q_end ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + 16 * t + 8 * half) * 4u : OOB;
      qa[t] = buf_load4(rq, off);
      qc[t] = buf_load4(rq, off == OOB ? OOB : off + 16u);
    }
  };
  auto commit_ring_rest = [&](const Item &c) {
    if (has_e && NR > 0) {
      const int rb = rb_step(c.k_begin, c.evq_b0);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int idx = KS * NQD + tid + 512 * i, row = idx / NQD, qd = idx % NQD;
        put_row(Ep, RING * KLD, ring_slot(rb + row), qd, pr[i]);
      }
    }
  };

  Item cur = item_of(0);
  if (!cur.valid) return;
  prefetch(cur.k_begin, cur.evq_b0);

  for (int it = 0; cur.valid; ++it) {
    ISI_F2_STAMP(0);
    const Item nxt = item_of(it + 1);
    const int q0 = cur.q0, q_end = cur.q_end, k_begin = cur.k_begin, k_end = cur.k_end;
    const int qw0 = q0 + 32 * wq, qi = qw0 + ql;
    load_block(cur);
    ISI_F2_STAMP(1);
    commit(k_begin, cur.evq_b0);
    commit_ring_rest(cur);
    ISI_F2_STAMP(2);
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
    __syncthreads();
    ISI_F2_STAMP(3);
    const int evq = UNIT ? qi : qi / p.Cq;
    const int evq_w0 = UNIT ? qw0 : qw0 / p.Cq;

    float m_run = NEG, l_run = 0.f;     // l_run: this lane's 16 keys per sub-block only (the halves are added at the end)
    f32x16 O[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = 0.f;

    bool next_requested = false;
    for (int kp = k_begin; kp < k_end; kp += KS) {
      const bool more = kp + KS < k_end;
      [[maybe_unused]] const int sbase = 4 + 12 * ((kp - k_begin) / KS);
      ISI_F2_STAMP(sbase);
      // what the end of this step commits: the next step of this block, or the first step of the next block
      const bool req = more || nxt.valid;
      const int kp_req = more ? kp + KS : nxt.k_begin, ev_req = more ? cur.evq_b0 : nxt.evq_b0;
      if (req) { prefetch_k(kp_req); if (!more) next_requested = true; }
      int req_stage = 0;   // parts requested so far besides K (wave-uniform: a skipped sub-block requests nothing)

      ISI_F2_STAMP(sbase + 1);
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        const int k0 = kp + grp * KT + 32 * sb;
        bool live = qw0 < q_end && k0 < k_end;
        if (p.mask_mode == 1) live = live && k0 <= min(qw0 + 31, q_end - 1);
        if (p.mask_mode == 2) live = live && k0 + 31 >= qw0;
        if (!live) continue;   // wave-uniform

        // ---- S^T = K Q^T and the band tiles E Q^T (band row t of this wave = table row wrow0 + t): independent
        // accumulator chains, issued round-robin
        f32x16 sacc, racc0, racc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = racc0[r] = racc1[r] = 0.f;
        const int wrow0 = evq_w0 - (UNIT ? k0 + 31 : (k0 + 31) / p.Ck) + p.Ek - 1;
        // operand fragments: 16 bytes per lane and k-block at (row, 16 t + 8 half); two fragment sets in flight
        const unsigned short *ka = Kb + (32 * sb + ql) * KLD + 8 * half;
        const unsigned short *ea0 = Ep + ring_slot(wrow0 + ql) * KLD + 8 * half;
        const unsigned short *ea1 = Ep + ring_slot(wrow0 + 32 + ql) * KLD + 8 * half;
        s16x8_t fk[2][NPL], f0[2][NPL], f1[2][NPL];
        auto rd = [&](int t, int s_) {
#pragma unroll
          for (int pl_ = 0; pl_ < NPL; ++pl_) {
            fk[s_][pl_] = *reinterpret_cast<const s16x8_t *>(ka + 16 * t + pl_ * KT * KLD);
            if (has_e) {
              f0[s_][pl_] = *reinterpret_cast<const s16x8_t *>(ea0 + 16 * t + pl_ * RING * KLD);
              if (two_tiles) f1[s_][pl_] = *reinterpret_cast<const s16x8_t *>(ea1 + 16 * t + pl_ * RING * KLD);
            }
          }
        };
        rd(0, 0);
#pragma unroll
        for (int t = 0; t < NKB; ++t) {
          const int c_ = t & 1;
          if (t + 1 < NKB) rd(t + 1, c_ ^ 1);
          if constexpr (ONE) {
            if (has_e) {
              racc0 = PR::mfma(f0[c_][0], qh[t], racc0);
              if (two_tiles) racc1 = PR::mfma(f1[c_][0], qh[t], racc1);
            }
            sacc = PR::mfma(fk[c_][0], qh[t], sacc);
          } else {
            if (has_e) {
              racc0 = PR::mfma(f0[c_][NPL - 1], qh[t], racc0);
              if (two_tiles) racc1 = PR::mfma(f1[c_][NPL - 1], qh[t], racc1);
            }
            sacc = PR::mfma(fk[c_][NPL - 1], qh[t], sacc);
            if (has_e) {
              racc0 = PR::mfma(f0[c_][0], qlo[t], racc0);
              if (two_tiles) racc1 = PR::mfma(f1[c_][0], qlo[t], racc1);
            }
            sacc = PR::mfma(fk[c_][0], qlo[t], sacc);
            if (has_e) {
              racc0 = PR::mfma(f0[c_][0], qh[t], racc0);
              if (two_tiles) racc1 = PR::mfma(f1[c_][0], qh[t], racc1);
            }
            sacc = PR::mfma(fk[c_][0], qh[t], sacc);
          }
        }
        ISI_F2_STAMP(sbase + 2 + 4 * sb);
        if (req && req_stage == 0) { prefetch_e(kp_req, ev_req); req_stage = 1; }
        // ---- the skew: band tiles -> per-wave LDS buffer [query][band row] -> the entry of (query, key)
        if (has_e) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(sr + 8 * g + 4 * half) = make_float4(racc0[4 * g], racc0[4 * g + 1], racc0[4 * g + 2], racc0[4 * g + 3]);
          if (two_tiles) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4 *>(sr + 32 + 8 * g + 4 * half) = make_float4(racc1[4 * g], racc1[4 * g + 1], racc1[4 * g + 2], racc1[4 * g + 3]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          float rl[16];
          if constexpr (UNIT) {
            const float *rd = sr + ql + 31 - 4 * half;     // band row ql + 31 - (key row in the sub-block)
#pragma unroll
            for (int r = 0; r < 16; ++r) rl[r] = rd[-((r & 3) + 8 * (r >> 2))];
          } else {
            const int dq = evq - evq_w0;
            if (ck_regular) {
#pragma unroll
              for (int r = 0; r < 16; ++r) rl[r] = sr[dq + evoff[r]];
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) rl[r] = sr[dq + evk_at(grp * KT + 32 * sb + mfma_row(r, half))];
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] += rl[r];
        }

        ISI_F2_STAMP(sbase + 3 + 4 * sb);
        // ---- mask, online softmax (base 2; q carries scale * log2 e)
        float sv[16];
        float tmax;
        // the row of this lane's query in the kept logits (training: the backward reads them back), its 16 keys of this sub-block
        float *lrow = p.logits ? p.logits + (((size_t)b * p.H + h) * p.Sq + min(qi, p.Sq - 1)) * p.ldl + k0 + 4 * half : nullptr;
        bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < q_end;
        if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
        if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
        float psum = 0.f;
        float alpha;
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
          // visible keys of this lane's query: [klo, khi] (predicates instead of a mask tensor)
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

        ISI_F2_STAMP(sbase + 4 + 4 * sb);
        if (req && req_stage == 1) { prefetch_v(kp_req); req_stage = 2; }
        // ---- P as MFMA B operand: key block t = registers 8 t .. 8 t + 7
        s16x8_t ph[2], pl[2];
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
            pl[t] = __builtin_bit_cast(s16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));
          }
        }
        // ---- O^T = alpha * O^T + V^T P^T
        const bool rescale = __any(alpha != 1.f);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          if (rescale) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
          }
          // this wave's 16 keys of k-step t inside the step: 64-key tile and 16-key group of it
          const int ko = grp * KT + 32 * sb;
          const unsigned short *vr = Vp + (((ko / 64) * NPL) * VR + d * 32 + ql) * VLD;
          const int vx = vswz(ql);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int c = 2 * (((ko % 64) >> 4) + t) + half;
            const s16x8_t vh = *reinterpret_cast<const s16x8_t *>(vr + ((c ^ vx) * 8));
            if constexpr (!ONE) {
              const s16x8_t vl = *reinterpret_cast<const s16x8_t *>(vr + VR * VLD + ((c ^ vx) * 8));
              O[d] = PR::mfma(vl, ph[t], O[d]);
              O[d] = PR::mfma(vh, pl[t], O[d]);
            }
            O[d] = PR::mfma(vh, ph[t], O[d]);
          }
        }
        ISI_F2_STAMP(sbase + 5 + 4 * sb);
      }
      if (req && req_stage < 1) prefetch_e(kp_req, ev_req);
      if (req && req_stage < 2) prefetch_v(kp_req);
      __syncthreads();           // every wave is done with this step's tiles
      ISI_F2_STAMP(sbase + 10);
      if (more) {
        commit(kp + KS, cur.evq_b0);
        __syncthreads();
      }
      ISI_F2_STAMP(sbase + 11);
    }
    if (nxt.valid) {             // the next block's operands travel while this one is merged and stored
      if (!next_requested) prefetch(nxt.k_begin, nxt.evq_b0);
    }

    ISI_F2_STAMP(240);
    // ---- merge the two groups' softmax states (group 1 -> LDS -> group 0)
    float *mg = smem;
    constexpr int MGW = (NDB * 16 + 2) * 64;
    static_assert((size_t)4 * MGW * sizeof(float) <= TL::operand_bytes,
                  "merge buffer must fit in the operand area");
    if (grp == 1) {
      float *dst = mg + wq * MGW + lane;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = O[d][r];
      dst[NDB * 16 * 64] = m_run;
      dst[(NDB * 16 + 1) * 64] = l_run;
    }
    __syncthreads();
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
    __syncthreads();            // the merge buffer is free again: the next block's commit may overwrite it
    ISI_F2_STAMP(241);
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
          if (dd < HD)
            *reinterpret_cast<float4 *>(orow + dd) =
                make_float4(O[d][4 * g] * inv, O[d][4 * g + 1] * inv, O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv);
        }
    }
    ISI_F2_STAMP(242);
    cur = nxt;
  }
}

int rel_attention_fwd2_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd2_stamps), sizeof(long long) * (size_t)(n < 512 ? n : 512)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

bool rel_attention_fwd2_ok(const AttnKArgs &a, int head_dim) {
  (void)head_dim;
  // the band ring holds 127/Cq + (KS-1)/Ck + 1 rows for any Cq, Ck >= 1
  return a.Cq >= 1 && a.Ck >= 1;
}

namespace {
template <int HD, int TERMS, bool F16, bool UNIT>
int launch_fwd2_t(const AttnKArgs &a, hipStream_t stream) {
  auto kern = rel_attn_fwd2_kernel<HD, TERMS, F16, UNIT>;
  constexpr size_t smem = Tile<HD, TERMS>::smem;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(rel_attn_fwd2)");
    attr_set.mark();
  }
  // workgroups per (batch, head) pair: one block each while the chip is not full; otherwise as many as give every CU one
  // persistent workgroup, at most one per two blocks under a mask (a heavy and a light block each)
  const int pairs = a.H * a.B, cus = current_device_cu_count();
  int nW = a.nblk;
  if ((int64_t)pairs * a.nblk > cus)
    nW = std::max(1, std::min(a.mask_mode ? (a.nblk + 1) / 2 : a.nblk, cus / pairs));
  const double npairs = (double)a.Sq * a.Sk * (a.mask_mode ? 0.5 : 1.0) * pairs;
  prof::Scope scope(prof::K_REL_ATTENTION, 2.0 * npairs * HD * (a.e ? 3 : 2),
                    4.0 * pairs * HD * (2.0 * a.Sq + 2.0 * a.Sk), stream);
  ISI_PROF_LAUNCH(scope, kern, dim3(xcd_grid(nW, pairs)), dim3(512), smem, stream, a, nW);
  return check_launch("rel_attn_fwd2");
}
template <int HD>
int launch_fwd2_hd(const AttnKArgs &a, int precision, hipStream_t stream) {
  const bool unit = a.Cq == 1 && a.Ck == 1;
  switch (precision) {
    case 1: return unit ? launch_fwd2_t<HD, 3, false, true>(a, stream) : launch_fwd2_t<HD, 3, false, false>(a, stream);
    case 2: return unit ? launch_fwd2_t<HD, 1, false, true>(a, stream) : launch_fwd2_t<HD, 1, false, false>(a, stream);
    case 3: return unit ? launch_fwd2_t<HD, 1, true, true>(a, stream) : launch_fwd2_t<HD, 1, true, false>(a, stream);
    default: return unsupported("rel_attention: precision");
  }
}
}  // namespace

int rel_attention_fwd2(const AttnKArgs &a, int head_dim, int precision, hipStream_t stream) {
  switch (head_dim) {
    case 16: return launch_fwd2_hd<16>(a, precision, stream);
    case 32: return launch_fwd2_hd<32>(a, precision, stream);
    case 64: return launch_fwd2_hd<64>(a, precision, stream);
    default: return unsupported("rel_attention: head_dim must be 16, 32 or 64");
  }
}

}  // namespace isi
