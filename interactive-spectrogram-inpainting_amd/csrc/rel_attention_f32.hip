// Multi-head attention with learned relative-position logits for gfx950
// (exact-fp32 matrix pipe), flash style: the Sq x Sk score matrix and the
// causal / anti-causal masks are never materialised.
//
//   logit[i,j] = ( q_i.k_j + q_i.e[h, r(i,j)] ) * scale + mask(i,j)
//   r(i,j)     = floor(i/Cq) - floor(j/Ck) + (Ek - 1)
//   out_i      = sum_j softmax_j(logit[i,:]) v_j
//
// This is the operator the reference reaches through the absent package
// VQCPCB.transformer.transformer_custom (priors/transformer.py:370-417,756-777);
// its specification for this repository is oracle/prior_oracle.py (parity
// unpinned).
//
// Workgroup = 4 waves = 128 consecutive queries of one (batch, head); each wave
// owns 32 queries.  Per tile of 32 keys (K, V and the needed band of e staged in
// LDS, zero-filled past the ends by out-of-range buffer offsets):
//   S^T   = K Q^T            (MFMA 32x32x2 f32, operands swapped: a lane's 16
//                             accumulators are 16 keys of ONE query, so the row
//                             max / sum of the online softmax are lane-local
//                             plus one exchange with lane^32)
//   Srel  = E_band Q^T       (same shape; the entry a (query,key) pair needs is
//                             read back through a per-wave LDS buffer: the skew)
//   O^T  += V^T P^T          (P^T is already the B fragment: accumulator r of a
//                             lane is the k-slot of MFMA step r)
#include <algorithm>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_bf16.h"
#include "rel_attention.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));


namespace {
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float NEG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int QB = 128;        // queries per workgroup
constexpr int BAND = 160;      // rows of e a key tile can touch (>= 127/Cq + 31/Ck + 1)
constexpr int BAND2 = 192;     // rows a PAIR of key tiles can touch (>= 127/Cq + 63/Ck + 1)
constexpr int RING = 192;      // rows of the band ring (>= BAND2: new rows replace rows no tile needs any more)
__device__ __forceinline__ int ring_slot(int r) {
  r %= RING;
  return r < 0 ? r + RING : r;
}
constexpr int SRLD = 65;       // per-query row of the skew buffer (64 + 1: conflict-free)

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ float elem(const float4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
}  // namespace

// Workgroup = 8 waves = 128 consecutive queries of one (batch, head), in two groups of four waves.
// Both groups own the same 4 x 32 queries; group g walks the key tiles of parity g with its own
// online-softmax state (m, l, O) and the two states are merged at the end.  Two waves per SIMD: the
// softmax / skew (VALU, LDS) phase of one overlaps the MFMAs of the other.
//
// Staging is software pipelined: while the waves work on a pair of key tiles, the global loads of
// the next pair are already in flight into registers; they are written to LDS between two barriers.
// The band of e lives in a ring of RING rows addressed by (table row mod RING): consecutive tile
// pairs shift the band by at most 64 rows, so only 64 rows are (re)loaded per pair.
template <int HD>
__global__ __launch_bounds__(512) void rel_attention_f32_kernel(const AttnKArgs p) {
  constexpr int LDH = HD + 4;  // padded LDS row
  constexpr int NQ = HD / 8;   // float4 fragments per lane along the head dim
  constexpr int NDB = (HD + 31) / 32;  // 32-wide blocks of the head dim in O
  constexpr int NKQ = (HD / 4 + 7) / 8;  // staging quads per thread and row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ks = smem;                    // [2][32][LDH]   tile of group 0 / group 1
  float *Vs = Ks + 2 * 32 * LDH;       // [2][32][LDH]
  float *Eb = Vs + 2 * 32 * LDH;       // [RING][LDH]
  float *Sr = Eb + RING * LDH;         // [8][32][SRLD]
  int *evk = reinterpret_cast<int *>(Sr + 8 * 32 * SRLD);  // [2][32] last event of the tile - event(key)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  // causal rows: the last query blocks see the most keys -> launch them first
  const int qblk = p.mask_mode == 1 ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int q0 = qblk * QB, qw0 = q0 + 32 * wq, qi = qw0 + ql;
  const bool has_e = p.e != nullptr;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  // ---- Q fragment of this lane's query: quads (2s + half)
  float4 qf[NQ];
#pragma unroll
  for (int s = 0; s < NQ; ++s) {
    const unsigned off = qi < p.Sq ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + (2 * s + half) * 4) * 4u : OOB;
    qf[s] = buf_load4(rq, off);
  }
  const int evq = qi / p.Cq;
  const int evq_w0 = qw0 / p.Cq, evq_b0 = q0 / p.Cq;

  float m_run = NEG, l_run = 0.f;
  f32x16 O[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;

  // key range this workgroup has to visit
  int k_begin = 0, k_end = p.Sk;
  if (p.mask_mode == 1) k_end = min(p.Sk, q0 + QB);
  if (p.mask_mode == 2) k_begin = (q0 / 32) * 32;

  // staging role: tile st of the pair, row srow, quads squad + 8 i
  const int st = tid >> 8, srow = (tid >> 3) & 31, squad = tid & 7;
  float4 pk[NKQ], pv[NKQ], pe[NKQ];
  // first table row of the band of the key tile that starts at k
  auto band0 = [&](int k) { return evq_b0 - (k + 31) / p.Ck + p.Ek - 1; };
  auto prefetch = [&](int k0) {  // K, V rows of the pair at k0 and the 64 lowest rows of its band
    const int kj = k0 + 32 * st + srow;
    const bool ok = kj < p.Sk;
    const int r = band0(k0 + 32) + 32 * st + srow;
    const bool rok = has_e && r >= 0 && r < p.R;
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      const bool in = qd < HD / 4;
      pk[i] = buf_load4(rk, ok && in ? (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + qd * 4) * 4u : OOB);
      pv[i] = buf_load4(rv, ok && in ? (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + qd * 4) * 4u : OOB);
      pe[i] = buf_load4(re, rok && in ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
    }
  };
  auto commit = [&](int k0) {
    const int slot = ring_slot(band0(k0 + 32) + 32 * st + srow);
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      if (qd < HD / 4) {
        *reinterpret_cast<float4 *>(Ks + (st * 32 + srow) * LDH + qd * 4) = pk[i];
        *reinterpret_cast<float4 *>(Vs + (st * 32 + srow) * LDH + qd * 4) = pv[i];
        if (has_e) *reinterpret_cast<float4 *>(Eb + slot * LDH + qd * 4) = pe[i];
      }
    }
    if (tid < 64) {
      const int kt = k0 + (tid & 32);
      evk[tid] = (kt + 31) / p.Ck - (kt + (tid & 31)) / p.Ck;
    }
  };

  // ---- prologue: first pair and the rest of its band (BAND2 rows from band0 of the pair's second tile)
  if (k_begin < k_end) {
    prefetch(k_begin);
    commit(k_begin);
    if (has_e) {
      const int rb = band0(k_begin + 32);
      for (int row = 64 + (tid >> 3); row < BAND2; row += 64) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        for (int qd = squad; qd < HD / 4; qd += 8)
          *reinterpret_cast<float4 *>(Eb + ring_slot(r) * LDH + qd * 4) =
              buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the prologue are in; in-loop waits then only cover the prefetch
  __syncthreads();
  const float scale2 = p.scale * LOG2E;
  // (last event of a key tile) - event(key): the same for every tile when Ck divides 32
  const bool ck_regular = (32 % p.Ck) == 0;
  int evoff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) evoff[r] = 31 / p.Ck - mfma_row(r, half) / p.Ck;
  const float *Kb = Ks + grp * 32 * LDH, *Vb = Vs + grp * 32 * LDH;
  const int *evkb = evk + grp * 32;

  for (int kp = k_begin; kp < k_end; kp += 64) {
    const bool more = kp + 64 < k_end;
    if (more) prefetch(kp + 64);
    const int k0 = kp + 32 * grp;  // this group's tile
    const int rb = band0(k0);

    // does this wave's query tile see any key of this tile?
    bool live = qw0 < p.Sq && k0 < k_end;
    if (p.mask_mode == 1) live = live && k0 <= qw0 + 31;
    if (p.mask_mode == 2) live = live && k0 + 31 >= qw0;
    if (live) {  // wave-uniform
      // ---- S^T = K Q^T
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      {
        const float *kr = Kb + ql * LDH + half * 4;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
          const float4 kf = *reinterpret_cast<const float4 *>(kr + s * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(kf, e), elem(qf[s], e), sacc, 0, 0, 0);
        }
      }
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];

      // ---- relative logits through the skew buffer
      if (has_e) {
        float *sr = Sr + wave * 32 * SRLD + ql * SRLD;
        const int wrow0 = rb + evq_w0 - evq_b0;  // table row of this wave's first band row
        const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;  // 32-row tiles of the band actually reachable
        for (int t = 0; t < nt; ++t) {
          f32x16 racc;
#pragma unroll
          for (int r = 0; r < 16; ++r) racc[r] = 0.f;
          const float *er = Eb + ring_slot(wrow0 + 32 * t + ql) * LDH + half * 4;
#pragma unroll
          for (int s = 0; s < NQ; ++s) {
            const float4 ef = *reinterpret_cast<const float4 *>(er + s * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              racc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(ef, e), elem(qf[s], e), racc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sr[32 * t + mfma_row(r, half)] = racc[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int dq = evq - evq_w0;
        if (ck_regular) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sv[r] += sr[dq + evoff[r]];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) sv[r] += sr[dq + evkb[mfma_row(r, half)]];
        }
        __builtin_amdgcn_wave_barrier();
      }

      // ---- scale, mask, online softmax (this lane: query qi, 16 of the tile's keys)
      // (logits are kept in base 2: exp2 is one v_exp_f32; m_run, tmax are base-2 maxima)
      float tmax = NEG;
      // tiles entirely inside the visible region need no per-element predicate
      bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < p.Sq;
      if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
      if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sv[r] *= scale2;
          tmax = fmaxf(tmax, sv[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = k0 + mfma_row(r, half);
          bool ok = kj < p.Sk;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float s = sv[r] * scale2;
          if (p.mask && ok && qi < p.Sq) s += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;
          s = ok ? s : NEG;
          sv[r] = s;
          tmax = fmaxf(tmax, s);
        }
      }
      tmax = fmaxf(tmax, xor32_f32(tmax));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = sv[r] <= -1e29f ? 0.f : __builtin_amdgcn_exp2f(sv[r] - m_new);
        sv[r] = pr;
        psum += pr;
      }
      psum += xor32_f32(psum);
      l_run = l_run * alpha + psum;
      m_run = m_new;

      // ---- O^T = alpha * O^T + V^T P^T   (the running maximum settles after a few tiles:
      //      skip the rescale when no lane of the wave needs it)
      const bool rescale = __any(alpha != 1.f);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        if (rescale) {
#pragma unroll
          for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
        }
        const int dcol = d * 32 + ql;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float vv = dcol < HD ? Vb[mfma_row(t, half) * LDH + dcol] : 0.f;
          O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sv[t], O[d], 0, 0, 0);
        }
      }
    }
    __syncthreads();           // every wave is done with this pair's tiles
    if (more) commit(kp + 64);
    __syncthreads();
  }

  // ---- merge the two groups' softmax states (group 1 -> LDS -> group 0)
  float *mg = smem;            // [4][NDB*16 + 2][64]
  constexpr int MGW = (NDB * 16 + 2) * 64;
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
  if (grp == 1) return;
  {
    const float *src = mg + wq * MGW + lane;
    const float m1 = src[NDB * 16 * 64], l1 = src[(NDB * 16 + 1) * 64];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * a0 + l1 * a1;
    m_run = m;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = O[d][r] * a0 + src[(d * 16 + r) * 64] * a1;
  }

  // ---- normalise and store: accumulator regs 4g..4g+3 are 4 consecutive head dims
  if (qi < p.Sq) {
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
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant (args->precision = 1, the Python default): the three contractions run on the bf16
// matrix pipe as THREE-term split products (x = hi + lo in bf16; hi.hi + hi.lo + lo.hi, fp32
// accumulation: relative error of a product ~2^-16), everything else -- logits, skew, online softmax,
// merge -- stays fp32.  The exact-fp32 kernel above is matrix-bound (the band GEMM doubles the QK^T
// work); here the matrix time drops 5x and the kernel becomes bound by the softmax / skew arithmetic.
//   S^T  = K Q^T     A = K rows from LDS bf16 planes [key][HD], B = the lane's Q fragment (registers)
//   band = E Q^T     A = ring rows of e
//   O^T += V^T P^T   A = V^T from LDS planes [d][32 keys] (V is transposed while staged: a thread stages a
//                    4 keys x 4 dims block), B = P packed from the accumulator layout; MFMA k-slot (half h, e)
//                    of key block t is key 16 t + 8 (e >> 2) + 4 h + (e & 3), i.e. exactly the lane's registers
// LDS rows are unpadded and XOR-swizzled: 16-B slot s of row r of a [.][HD] plane sits at s ^ ((r / (128/HD))
// mod (HD/8)); 8-B unit u of row d of a V^T plane at u ^ ((d >> 2) & 7)  (conflict-free ds_read_b128 / b64).
// ONE = true (args->precision = 2): SINGLE-term bf16 products (north_star's "MFMA bf16" mode): the lo planes are
// neither computed, stored nor multiplied -- a third of the matrix work, half of the staging conversions and LDS
// traffic; operands are rounded to bf16 (8 significand bits), accumulation / logits / softmax stay fp32.
// phase timestamps (-DISI_MEASURE builds; tools/stamps_attention.py): the heaviest full workgroup (blockIdx.x == 1 of
// (h, b) = (0, 0)), waves 0 and 4, key-pair iteration 4
#ifdef ISI_MEASURE
__device__ long long g_attn_stamps[64];
#define ISI_ATT_STAMP(i_) do { if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && kp == k_begin + 4 * 64) \
    g_attn_stamps[(wave >> 2) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_ATT_STAMP(i_) do { } while (0)
#endif
template <int HD, bool ONE = false>
__global__ __launch_bounds__(512) void rel_attention_split_kernel(const AttnKArgs p) {
  constexpr int NKB = HD / 16;           // 16-deep k-blocks of the head dim
  constexpr int NSL = HD / 8;            // 16-B slots per [.][HD] row
  constexpr int RPB = 128 / HD;          // rows per 256-B bank row
  constexpr int NDB = (HD + 31) / 32;    // 32-row blocks of O^T
  constexpr int VR = NDB * 32;           // rows of a V^T plane (rows >= HD stay zero)
  constexpr int NQD = HD / 4;            // dim quads per key row
  constexpr int NKQ = (HD / 4 + 7) / 8;  // band quads per thread and row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Kp = reinterpret_cast<unsigned short *>(smem);   // [tile 2][plane 2][32][HD]
  unsigned short *Vp = Kp + 2 * 2 * 32 * HD;                       // [tile 2][plane 2][VR][32]
  unsigned short *Ep = Vp + 2 * 2 * VR * 32;                       // [plane 2][RING][HD]
  float *Sr = reinterpret_cast<float *>(Ep + 2 * RING * HD);       // [8][32][SRLD]
  int *evk = reinterpret_cast<int *>(Sr + 8 * 32 * SRLD);          // [2][32]
  auto swz = [](int row, int slot) { return (slot ^ ((row / RPB) % NSL)) * 8; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  // (tile, head, batch) from the 1-D launch: the query blocks of one (batch, head) share an XCD's L2 (xcd_tile)
  const int nqb = p.nblk;
  int qt, pair;
  if (!xcd_tile(nqb, p.H * p.B, p.mask_mode != 0, qt, pair)) return;
  const int h = pair % p.H, b = pair / p.H;
  const int qblk = p.mask_mode == 1 ? nqb - 1 - qt : qt;      // heavy blocks first
  // causal masks: the ragged block (Sq % QB rows: the ONE extra row of a 1025-row sequence) is block 0, where the key
  // range is shortest, instead of the last block, where it cost as much as a full one (9 of 45 block-steps at
  // Sq = 1025).  Blocks then start at rag + 128 (qblk - 1); the band logic takes any origin when Cq = 1.
  const int rag = (p.mask_mode == 1 && p.Cq == 1 && nqb * QB >= p.Sq) ? p.Sq % QB : 0;
  const int q0 = rag ? (qblk ? rag + (qblk - 1) * QB : 0) : qblk * QB;
  const int q_end = (rag && qblk == 0) ? rag : p.Sq;           // first row beyond this block's valid ones
  const int qw0 = q0 + 32 * wq, qi = qw0 + ql;
  const bool has_e = p.e != nullptr;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  // ---- Q fragment of this lane's query, split once: k-block t holds dims 16 t + 8 half + 0..7
  s16x8_t qh[NKB], qlo[NKB];
#pragma unroll
  for (int t = 0; t < NKB; ++t) {
    const unsigned off = qi < q_end ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + 16 * t + 8 * half) * 4u : OOB;
    uint2 h0, l0, h1, l1;
    split_f4(buf_load4(rq, off), h0, l0);
    split_f4(buf_load4(rq, off == OOB ? OOB : off + 16u), h1, l1);
    qh[t] = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
    qlo[t] = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));   // dead when ONE
  }
  const int evq = qi / p.Cq;
  const int evq_w0 = qw0 / p.Cq, evq_b0 = q0 / p.Cq;

  float m_run = NEG, l_run = 0.f;
  f32x16 O[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;

  int k_begin = 0, k_end = p.Sk;
  if (p.mask_mode == 1) k_end = min(p.Sk, q0 + QB);
  if (p.mask_mode == 2) k_begin = (q0 / 32) * 32;

  // staging roles.  K / V: a 4 keys x 4 dims block per thread (threads [0, 16 NQD) stage K, [256, 256 + 16 NQD) V);
  // band: row 32 st + srow of the pair's 64 new rows, quads squad + 8 i
  const int kind = tid >> 8, bidx = tid & 255;
  const int bqd = bidx % NQD, bkg = (bidx / NQD) & 7, btile = bidx / (8 * NQD);
  const bool blk_on = btile < 2;
  const int st = tid >> 8, srow = (tid >> 3) & 31, squad = tid & 7;
  float4 pb[4], pe[NKQ];
  auto band0 = [&](int k) { return evq_b0 - (k + 31) / p.Ck + p.Ek - 1; };
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kj = k0 + 32 * btile + 4 * bkg + j;
      const bool ok = blk_on && kj < p.Sk;
      const unsigned ko = (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + bqd * 4) * 4u;
      const unsigned vo = (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + bqd * 4) * 4u;
      pb[j] = kind == 0 ? buf_load4(rk, ok ? ko : OOB) : buf_load4(rv, ok ? vo : OOB);
    }
    const int r = band0(k0 + 32) + 32 * st + srow;
    const bool rok = has_e && r >= 0 && r < p.R;
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      pe[i] = buf_load4(re, rok && qd < NQD ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
    }
  };
  auto commit = [&](int k0) {
    if (blk_on) {
      if (kind == 0) {   // K rows: 8 bytes (4 dims) per key and plane
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * bkg + j;
          uint2 hi, lo;
          split_f4(pb[j], hi, lo);
          const int o = ((btile * 2) * 32 + row) * HD + swz(row, bqd >> 1) + (bqd & 1) * 4;
          *reinterpret_cast<uint2 *>(Kp + o) = hi;
          if constexpr (!ONE) *reinterpret_cast<uint2 *>(Kp + o + 32 * HD) = lo;
        }
      } else {           // V transposed: per dim the 4 keys of the block as one 8-byte unit
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = 4 * bqd + e;
          const float a0 = e == 0 ? pb[0].x : e == 1 ? pb[0].y : e == 2 ? pb[0].z : pb[0].w;
          const float a1 = e == 0 ? pb[1].x : e == 1 ? pb[1].y : e == 2 ? pb[1].z : pb[1].w;
          const float a2 = e == 0 ? pb[2].x : e == 1 ? pb[2].y : e == 2 ? pb[2].z : pb[2].w;
          const float a3 = e == 0 ? pb[3].x : e == 1 ? pb[3].y : e == 2 ? pb[3].z : pb[3].w;
          uint2 hi, lo;
          split2(a0, a1, hi.x, lo.x);
          split2(a2, a3, hi.y, lo.y);
          const int o = ((btile * 2) * VR + d) * 32 + ((bkg ^ ((d >> 2) & 7)) * 4);
          *reinterpret_cast<uint2 *>(Vp + o) = hi;
          if constexpr (!ONE) *reinterpret_cast<uint2 *>(Vp + o + VR * 32) = lo;
        }
      }
    }
    if (has_e) {
      const int slot = ring_slot(band0(k0 + 32) + 32 * st + srow);
#pragma unroll
      for (int i = 0; i < NKQ; ++i) {
        const int qd = squad + 8 * i;
        if (qd < NQD) {
          uint2 hi, lo;
          split_f4(pe[i], hi, lo);
          const int o = slot * HD + swz(slot, qd >> 1) + (qd & 1) * 4;
          *reinterpret_cast<uint2 *>(Ep + o) = hi;
          if constexpr (!ONE) *reinterpret_cast<uint2 *>(Ep + o + RING * HD) = lo;
        }
      }
    }
    if (tid < 64) {
      const int kt = k0 + (tid & 32);
      evk[tid] = (kt + 31) / p.Ck - (kt + (tid & 31)) / p.Ck;
    }
  };

  // ---- prologue
  if (VR > HD) {   // rows of V^T beyond the head dim feed zero products
    for (int i = tid; i < 2 * 2 * VR * 32 / 2; i += 512) reinterpret_cast<unsigned *>(Vp)[i] = 0u;
    __syncthreads();
  }
  if (k_begin < k_end) {
    prefetch(k_begin);
    commit(k_begin);
    if (has_e) {
      const int rb = band0(k_begin + 32);
      for (int row = 64 + (tid >> 3); row < BAND2; row += 64) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        const int slot = ring_slot(r);
        for (int qd = squad; qd < NQD; qd += 8) {
          uint2 hi, lo;
          split_f4(buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB), hi, lo);
          const int o = slot * HD + swz(slot, qd >> 1) + (qd & 1) * 4;
          *reinterpret_cast<uint2 *>(Ep + o) = hi;
          if constexpr (!ONE) *reinterpret_cast<uint2 *>(Ep + o + RING * HD) = lo;
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const float scale2 = p.scale * LOG2E;
  const bool ck_regular = (32 % p.Ck) == 0;
  int evoff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) evoff[r] = 31 / p.Ck - mfma_row(r, half) / p.Ck;
  const unsigned short *Kb = Kp + (grp * 2) * 32 * HD, *Vb = Vp + (grp * 2) * VR * 32;
  const int *evkb = evk + grp * 32;

  for (int kp = k_begin; kp < k_end; kp += 64) {
    const bool more = kp + 64 < k_end;
    ISI_ATT_STAMP(0);
    if (more) prefetch(kp + 64);
    ISI_ATT_STAMP(1);
    const int k0 = kp + 32 * grp;
    const int rb = band0(k0);

    bool live = qw0 < q_end && k0 < k_end;
    if (p.mask_mode == 1) live = live && k0 <= qw0 + 31;
    if (p.mask_mode == 2) live = live && k0 + 31 >= qw0;
    if (live) {
      // ---- S^T = K Q^T
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const int o = ql * HD + swz(ql, 2 * t + half);
        const s16x8_t kh = *reinterpret_cast<const s16x8_t *>(Kb + o);
        if constexpr (!ONE) {
          const s16x8_t kl = *reinterpret_cast<const s16x8_t *>(Kb + o + 32 * HD);
          sacc = ISI_MFB(kl, qh[t], sacc);
          sacc = ISI_MFB(kh, qlo[t], sacc);
        }
        sacc = ISI_MFB(kh, qh[t], sacc);
      }
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];
      ISI_ATT_STAMP(2);

      // ---- relative logits through the skew buffer
      if (has_e) {
        float *sr = Sr + wave * 32 * SRLD + ql * SRLD;
        const int wrow0 = rb + evq_w0 - evq_b0;
        const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;
        for (int tb = 0; tb < nt; ++tb) {
          f32x16 racc;
#pragma unroll
          for (int r = 0; r < 16; ++r) racc[r] = 0.f;
          const int slot = ring_slot(wrow0 + 32 * tb + ql);
#pragma unroll
          for (int t = 0; t < NKB; ++t) {
            const int o = slot * HD + swz(slot, 2 * t + half);
            const s16x8_t eh = *reinterpret_cast<const s16x8_t *>(Ep + o);
            if constexpr (!ONE) {
              const s16x8_t el = *reinterpret_cast<const s16x8_t *>(Ep + o + RING * HD);
              racc = ISI_MFB(el, qh[t], racc);
              racc = ISI_MFB(eh, qlo[t], racc);
            }
            racc = ISI_MFB(eh, qh[t], racc);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sr[32 * tb + mfma_row(r, half)] = racc[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int dq = evq - evq_w0;
        if (ck_regular) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sv[r] += sr[dq + evoff[r]];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) sv[r] += sr[dq + evkb[mfma_row(r, half)]];
        }
        __builtin_amdgcn_wave_barrier();
      }

      ISI_ATT_STAMP(3);
      // ---- scale, mask, online softmax (base 2)
      float tmax = NEG;
      bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < q_end;
      if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
      if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sv[r] *= scale2;
          tmax = fmaxf(tmax, sv[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = k0 + mfma_row(r, half);
          bool ok = kj < p.Sk;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float sc = sv[r] * scale2;
          if (p.mask && ok && qi < q_end) sc += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;
          sc = ok ? sc : NEG;
          sv[r] = sc;
          tmax = fmaxf(tmax, sc);
        }
      }
      tmax = fmaxf(tmax, xor32_f32(tmax));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = sv[r] <= -1e29f ? 0.f : __builtin_amdgcn_exp2f(sv[r] - m_new);
        sv[r] = pr;
        psum += pr;
      }
      psum += xor32_f32(psum);
      l_run = l_run * alpha + psum;
      m_run = m_new;

      ISI_ATT_STAMP(4);
      // ---- P split: key block t = registers 8 t .. 8 t + 7
      s16x8_t ph[2], pl[2];
      split_acc16(sv, ph, pl);
      // ---- O^T = alpha * O^T + V^T P^T
      const bool rescale = __any(alpha != 1.f);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        if (rescale) {
#pragma unroll
          for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
        }
        const int drow = d * 32 + ql;
        const int sx = (drow >> 2) & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned short *vr = Vb + drow * 32;
          const uint2 h0 = *reinterpret_cast<const uint2 *>(vr + (((4 * t + half) ^ sx) * 4));
          const uint2 h1 = *reinterpret_cast<const uint2 *>(vr + (((4 * t + 2 + half) ^ sx) * 4));
          const s16x8_t vh = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
          if constexpr (!ONE) {
            const uint2 l0 = *reinterpret_cast<const uint2 *>(vr + VR * 32 + (((4 * t + half) ^ sx) * 4));
            const uint2 l1 = *reinterpret_cast<const uint2 *>(vr + VR * 32 + (((4 * t + 2 + half) ^ sx) * 4));
            const s16x8_t vl = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
            O[d] = ISI_MFB(vl, ph[t], O[d]);
            O[d] = ISI_MFB(vh, pl[t], O[d]);
          }
          O[d] = ISI_MFB(vh, ph[t], O[d]);
        }
      }
    }
    ISI_ATT_STAMP(5);
    __syncthreads();
    ISI_ATT_STAMP(6);
    if (more) commit(kp + 64);
    ISI_ATT_STAMP(7);
    __syncthreads();
    ISI_ATT_STAMP(8);
  }

  // ---- merge the two groups' softmax states (group 1 -> LDS -> group 0)
  float *mg = smem;
  constexpr int MGW = (NDB * 16 + 2) * 64;
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
  if (grp == 1) return;
  {
    const float *src = mg + wq * MGW + lane;
    const float m1 = src[NDB * 16 * 64], l1 = src[(NDB * 16 + 1) * 64];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * a0 + l1 * a1;
    m_run = m;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = O[d][r] * a0 + src[(d * 16 + r) * 64] * a1;
  }
  if (qi < q_end) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float *orow = p.out + (size_t)qi * p.o_ss + (size_t)b * p.o_sb + (size_t)h * p.o_sh;
    if (p.lse && half == 0)
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
}

// ---- one query row per workgroup, unmasked, exact fp32 (rel_attention_f32 below gives the one or two rows beyond the last
// full query block to this kernel instead of a block of their own).  G = HD / 4 lanes own a key row (one coalesced
// 16-byte load per lane for k, v and the relative row), 512 / G x 8 keys are in flight per step; every lane group keeps
// its own running softmax state, merged at the end (wave shuffles, then the 8 waves through LDS).
template <int HD>
__global__ __launch_bounds__(512) void attn_fwd_tail_row_kernel(const AttnKArgs p, int row0) {
  constexpr int G = HD / 4, RPP = 512 / G, U = 8, NW = 8;
  __shared__ __attribute__((aligned(16))) float part[NW * HD];
  __shared__ float red[2 * NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G, gl = tid % G;
  const int h = blockIdx.x, b = blockIdx.y, i = row0 + blockIdx.z;
  const float4 qq = *reinterpret_cast<const float4 *>(p.q + (size_t)i * p.q_ss + (size_t)b * p.q_sb + (size_t)h * p.q_sh + gl * 4);
  const int evq = i / p.Cq;
  const float *kb = p.k + (size_t)b * p.k_sb + (size_t)h * p.k_sh + gl * 4;
  const float *vb = p.v + (size_t)b * p.v_sb + (size_t)h * p.v_sh + gl * 4;
  const float *eb = p.e ? p.e + (size_t)h * p.R * HD + gl * 4 : nullptr;
  float m = NEG, l = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = grp; j0 < p.Sk; j0 += U * RPP) {
    float4 kk[U], vv[U], ee[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * RPP, jc = j < p.Sk ? j : j0;
      kk[u] = *reinterpret_cast<const float4 *>(kb + (size_t)jc * p.k_ss);
      vv[u] = *reinterpret_cast<const float4 *>(vb + (size_t)jc * p.v_ss);
      ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int rho = evq - jc / p.Ck + p.Ek - 1;
      if (eb && rho >= 0 && rho < p.R) ee[u] = *reinterpret_cast<const float4 *>(eb + (size_t)rho * HD);
    }
    float sv[U], bm = NEG;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 kq = make_float4(kk[u].x + ee[u].x, kk[u].y + ee[u].y, kk[u].z + ee[u].z, kk[u].w + ee[u].w);
      float acc = (qq.x * kq.x + qq.y * kq.y) + (qq.z * kq.z + qq.w * kq.w);
      acc = G == 16 ? row16_sum(acc) : G == 8 ? group8_sum(acc) : group4_sum(acc);
      sv[u] = j0 + u * RPP < p.Sk ? acc * p.scale : NEG;
      if (p.logits && gl == 0 && j0 + u * RPP < p.Sk)
        p.logits[(((size_t)b * p.H + h) * p.Sq + i) * p.ldl + j0 + u * RPP] = sv[u] * LOG2E;
      bm = fmaxf(bm, sv[u]);
    }
    const float mn = fmaxf(m, bm), corr = __expf(m - mn);
    l *= corr; o.x *= corr; o.y *= corr; o.z *= corr; o.w *= corr;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pj = j0 + u * RPP < p.Sk ? __expf(sv[u] - mn) : 0.f;
      l += pj;
      o.x += pj * vv[u].x; o.y += pj * vv[u].y; o.z += pj * vv[u].z; o.w += pj * vv[u].w;
    }
    m = mn;
  }
  // merge the wave's lane groups, then the waves
  float mw = m;
#pragma unroll
  for (int d = G; d < 64; d <<= 1) mw = fmaxf(mw, __shfl_xor(mw, d));
  const float f = __expf(m - mw);
  l *= f; o.x *= f; o.y *= f; o.z *= f; o.w *= f;
#pragma unroll
  for (int d = G; d < 64; d <<= 1) {
    l += __shfl_xor(l, d);
    o.x += __shfl_xor(o.x, d); o.y += __shfl_xor(o.y, d); o.z += __shfl_xor(o.z, d); o.w += __shfl_xor(o.w, d);
  }
  if (lane < G) *reinterpret_cast<float4 *>(part + wave * HD + lane * 4) = o;
  if (lane == 0) { red[wave] = mw; red[NW + wave] = l; }
  __syncthreads();
  if (tid < HD) {
    float M = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float fw = __expf(red[w] - M);
      num += fw * part[w * HD + tid];
      den += fw * red[NW + w];
    }
    p.out[(size_t)i * p.o_ss + (size_t)b * p.o_sb + (size_t)h * p.o_sh + tid] = num / den;
    if (p.lse && tid == 0) p.lse[((size_t)b * p.H + h) * p.Sq + i] = M + logf(den);
  }
}

int rel_attention_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(long long) * (size_t)(n < 64 ? n : 64)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

template <int HD>
static int launch_attn(const AttnKArgs &a, int B, hipStream_t stream) {
  auto kern = a.split == 2 ? rel_attention_split_kernel<HD, true> : a.split ? rel_attention_split_kernel<HD, false> : rel_attention_f32_kernel<HD>;
  constexpr int VR = ((HD + 31) / 32) * 32;
  constexpr size_t smem_f = (size_t)((128 + RING) * (HD + 4) + 8 * 32 * SRLD) * sizeof(float) + 64 * sizeof(int);
  constexpr size_t smem_s = (size_t)(2 * 2 * 32 * HD + 2 * 2 * VR * 32 + 2 * RING * HD) * sizeof(unsigned short) +
                            (size_t)(8 * 32 * SRLD) * sizeof(float) + 64 * sizeof(int);
  const size_t smem = a.split ? smem_s : smem_f;
  static DeviceOnce attr_set[3];
  if (!attr_set[a.split].done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(rel_attention)");
    attr_set[a.split].mark();
  }
  const double pairs = (double)a.Sq * a.Sk * (a.mask_mode ? 0.5 : 1.0) * a.H * B;
  prof::Scope scope(prof::K_REL_ATTENTION, 2.0 * pairs * HD * (a.e ? 3 : 2),
                    4.0 * B * a.H * HD * (2.0 * a.Sq + 2.0 * a.Sk), stream);
  if (a.split) ISI_PROF_LAUNCH(scope, kern, dim3(xcd_grid(a.nblk, a.H * B)), dim3(512), smem, stream, a);   // (xcd_tile)
  else ISI_PROF_LAUNCH(scope, kern, dim3((a.Sq + QB - 1) / QB, a.H, B), dim3(512), smem, stream, a);
  return check_launch("rel_attention_f32");
}

static int64_t span(int64_t S, int64_t ss, int64_t B, int64_t sb, int64_t H, int64_t sh, int64_t hd) {
  return (S - 1) * ss + (B - 1) * sb + (H - 1) * sh + hd;
}

int rel_attention_f32(const isi_attn_args *g, hipStream_t stream) {
  if (!g || !g->q || !g->k || !g->v || !g->out) return invalid("rel_attention: null pointer");
  if (g->Sq <= 0 || g->Sk <= 0 || g->B <= 0 || g->H <= 0) return invalid("rel_attention: bad shape");
  if (g->Cq <= 0 || g->Ck <= 0 || g->Ek <= 0) return invalid("rel_attention: bad event layout");
  if (g->mask_mode < 0 || g->mask_mode > 2) return invalid("rel_attention: bad mask mode");
  if (127 / g->Cq + 31 / g->Ck + 1 > BAND) return unsupported("rel_attention: band too wide");
  const int64_t lim = (int64_t)1 << 30;
  const int64_t eq = span(g->Sq, g->q_ss, g->B, g->q_sb, g->H, g->q_sh, g->head_dim);
  const int64_t ek = span(g->Sk, g->k_ss, g->B, g->k_sb, g->H, g->k_sh, g->head_dim);
  const int64_t ev = span(g->Sk, g->v_ss, g->B, g->v_sb, g->H, g->v_sh, g->head_dim);
  const int64_t eo = span(g->Sq, g->o_ss, g->B, g->o_sb, g->H, g->o_sh, g->head_dim);
  if (eq > lim || ek > lim || ev > lim || eo > lim) return unsupported("rel_attention: tensor spans 4 GiB or more");
  const int64_t all = g->q_ss | g->q_sb | g->q_sh | g->k_ss | g->k_sb | g->k_sh | g->v_ss | g->v_sb | g->v_sh |
                      g->o_ss | g->o_sb | g->o_sh;
  if ((all & 3) || ((reinterpret_cast<uintptr_t>(g->q) | reinterpret_cast<uintptr_t>(g->k) |
                     reinterpret_cast<uintptr_t>(g->v) | reinterpret_cast<uintptr_t>(g->out) |
                     reinterpret_cast<uintptr_t>(g->rel_embeddings)) & 15))
    return invalid("rel_attention: strides must be multiples of 4 floats and pointers 16-byte aligned");
  AttnKArgs a;
  a.q = g->q; a.k = g->k; a.v = g->v; a.e = g->rel_embeddings; a.mask = g->dense_mask; a.out = g->out; a.lse = g->lse;
  a.q_bytes = (unsigned)(eq * 4); a.k_bytes = (unsigned)(ek * 4); a.v_bytes = (unsigned)(ev * 4);
  a.R = g->rel_rows;
  a.e_bytes = (unsigned)((size_t)g->H * g->rel_rows * g->head_dim * 4);
  a.Sq = g->Sq; a.Sk = g->Sk; a.H = g->H; a.B = g->B;
  a.q_ss = (int)g->q_ss; a.q_sb = (int)g->q_sb; a.q_sh = (int)g->q_sh;
  a.k_ss = (int)g->k_ss; a.k_sb = (int)g->k_sb; a.k_sh = (int)g->k_sh;
  a.v_ss = (int)g->v_ss; a.v_sb = (int)g->v_sb; a.v_sh = (int)g->v_sh;
  a.o_ss = (int)g->o_ss; a.o_sb = (int)g->o_sb; a.o_sh = (int)g->o_sh;
  a.Cq = g->Cq; a.Ck = g->Ck; a.Ek = g->Ek;
  a.mask_mode = g->mask_mode; a.scale = g->scale;
  if (g->precision < 0 || g->precision > 3) return invalid("rel_attention: precision must be 0 .. 3");
  // 0 fp32 pipe | 1 three-term split-bf16 | 2 single-term bf16 | 3 single-term f16 (rel_attention_fwd2.hip only)
  a.split = g->precision == 1 ? 1 : g->precision >= 2 ? 2 : 0;
  a.logits = g->logits; a.ldl = (int)g->logits_ld;
  if (a.logits) {
    if (g->logits_ld < ((g->Sk + 31) & ~31) || (g->logits_ld & 3) || (reinterpret_cast<uintptr_t>(a.logits) & 15))
      return invalid("rel_attention: logits_ld must be a multiple of 4, at least Sk rounded up to 32; logits 16-byte aligned");
    if ((int64_t)g->B * g->H * g->Sq * g->logits_ld > ((int64_t)1 << 40)) return unsupported("rel_attention: logits buffer too large");
    if (!a.split || (g->precision != 3 && knobs().attn_old_fwd) || !rel_attention_fwd2_ok(a, g->head_dim))
      return unsupported("rel_attention: the logits are stored by the 64-key-tile kernels only (precision >= 1)");
  }
  if (a.e && a.R <= 0) return invalid("rel_attention: rel_rows must be positive");
  // One or two rows beyond the last full query block (the prior's sequences are 1024 codes + a start row) would be a
  // block of their own that runs as long as a full one: 576 instead of 512 workgroups on 256 CUs -- a third round
  // for one row per (batch, head) (244 vs 165 us, dense, B 8 x H 8 x 1025 x 1025).  They go through a one-row kernel
  // instead (attn_fwd_tail_row_kernel: exact fp32, one workgroup per (batch, head) and row).
  const int tail = attention_tail_rows(g->Sq, g->mask_mode, g->dense_mask != nullptr);
  const bool split_tail = a.split && tail > 0;
  a.nblk = split_tail ? g->Sq / QB : (g->Sq + QB - 1) / QB;
  int rc;
  if (a.split && g->workspace && !knobs().attn_no_fwd3 && !knobs().attn_old_fwd && rel_attention_fwd3_ok(a, g->head_dim, g->precision)) {
    // operands as 16-bit planes in the caller's workspace, staged by LDS-DMA (rel_attention_fwd3.hip)
    rc = rel_attention_fwd3(a, g->head_dim, g->precision, g->workspace, g->workspace_bytes, stream);
  } else
  if (a.split && (g->precision == 3 || !knobs().attn_old_fwd) && rel_attention_fwd2_ok(a, g->head_dim)) {
    rc = rel_attention_fwd2(a, g->head_dim, g->precision, stream);
  } else
  switch (g->head_dim) {
    case 16: rc = launch_attn<16>(a, g->B, stream); break;
    case 32: rc = launch_attn<32>(a, g->B, stream); break;
    case 64: rc = launch_attn<64>(a, g->B, stream); break;
    default: return unsupported("rel_attention: head_dim must be 16, 32 or 64");
  }
  if (rc || !split_tail) return rc;
  const dim3 tgrid(g->H, g->B, tail);
  switch (g->head_dim) {
    case 16: hipLaunchKernelGGL(attn_fwd_tail_row_kernel<16>, tgrid, dim3(512), 0, stream, a, g->Sq - tail); break;
    case 32: hipLaunchKernelGGL(attn_fwd_tail_row_kernel<32>, tgrid, dim3(512), 0, stream, a, g->Sq - tail); break;
    default: hipLaunchKernelGGL(attn_fwd_tail_row_kernel<64>, tgrid, dim3(512), 0, stream, a, g->Sq - tail); break;
  }
  return check_launch("attn_fwd_tail_row");
}

// bytes of isi_attn_args.workspace that let the forward run its plane-staged kernel (0: that kernel does not take the call)
size_t rel_attention_workspace_bytes(const isi_attn_args *g) {
  if (!g || g->Sq <= 0 || g->Sk <= 0 || g->B <= 0 || g->H <= 0 || g->precision < 1 || g->precision > 3) return 0;
  AttnKArgs a{};
  a.e = g->rel_embeddings; a.R = g->rel_rows; a.Sq = g->Sq; a.Sk = g->Sk; a.H = g->H; a.B = g->B; a.Cq = g->Cq; a.Ck = g->Ck;
  // the plane-staged kernel pays a pack launch (10-13 us at B 8 x H 8 x S 1025): measured, it wins with three-term products
  // (causal 91 vs 96 us, unmasked 127 vs 141) and loses 3-4 us with single-term ones -- those ask for no workspace unless
  // ISI_ATTN_FWD3_ALL is set (a caller that hands one over anyway gets the plane-staged kernel)
  if (g->precision != 1 && !knobs().attn_fwd3_all) return 0;
  if (knobs().attn_no_fwd3 || knobs().attn_old_fwd) return 0;
  return rel_attention_fwd3_workspace_bytes(a, g->head_dim, g->precision);
}

// Rows (keys, in the backward's key-stationary kernel) beyond the last full 128-row block that are NOT given a block of
// their own: 1 or 2 of them behind at least two full blocks, unmasked attention only (under a causal mask the ragged
// block is made the cheapest one instead, see the kernels: measured better than the extra launch).
int attention_tail_rows(int S, int mask_mode, bool dense_mask) {
  const int tail = S % QB;
  return (tail >= 1 && tail <= 2 && S >= 2 * QB && !dense_mask && mask_mode == 0) ? tail : 0;
}

}  // namespace isi
