// Backward of the relative-position attention of rel_attention_f32.hip, gfx950
// exact-fp32 matrix pipe.  Nothing of size Sq x Sk is stored by the forward: both
// kernels recompute the probabilities P = exp(logit - LSE) tile by tile from q, k,
// e and the per-query log-sum-exp the forward kernel wrote.
//
//   dP = dO V^T                      D_i = sum_d dO[i,d] O[i,d]
//   dS = P o (dP - D) * scale
//   dV = P^T dO      dK = dS^T Q      dQ = dS K + G E      dE = G^T Q
//   G[i, r] = sum over keys j with r(i,j) = r of dS[i,j]      (the "un-skew" of dS)
//
//   rel_attention_bwd_kv_kernel   key-stationary: a wave owns 32 keys (K, V rows in
//       registers, dK^T / dV^T in accumulators), query tiles stream through LDS.  No
//       atomics: every dK / dV element is produced by exactly one wave.
//   rel_attention_bwd_q_kernel    query-stationary (the forward's loop): dQ^T in
//       accumulators; dS is transposed through LDS so that one query's 32 keys land
//       on consecutive columns of G (row-coalesced float atomics; a G row is only
//       ever touched by the one wave that owns the query, in program order).
//   then two plain GEMMs on the existing kernels:  dQ += G E  (1x1 "convolution",
//       residual = dQ) and dE = G^T Q (the pixel-reduction GEMM of conv_wgrad_f32).
//
// Replaces autograd through the attention of the absent package
// VQCPCB.transformer.transformer_custom behind `loss.backward()`
// (reference train_autoregressive_model.py:257); specification:
// oracle/prior_oracle.py::attention differentiated by torch autograd.
#include <algorithm>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_bf16.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct AttnBwdKArgs {
  const float *q, *k, *v, *e, *mask, *dout, *out, *lse;
  float *dsum;            // [B,H,Sq]
  float *dq, *dk, *dv;
  float *g;               // [H][Sq][B][Rp]: rows query-major, so that the rows of a GEMM tile share their band of columns
  unsigned q_bytes, k_bytes, v_bytes, e_bytes, o_bytes;
  int Sq, Sk, H, B;
  int nqb, nkb;           // stationary query / key blocks the split kernels run (all, or only the full ones)
  int q_ss, q_sb, q_sh, k_ss, k_sb, k_sh, v_ss, v_sb, v_sh, o_ss, o_sb, o_sh;  // element strides
  int Cq, Ck, Ek, R, Rp;
  int rho_lo;             // G column c holds table row rho_lo + c (causal modes only touch half of the table)
  int g_band_only;        // G is zeroed only around its band: EVERY (query, key) pair the mask allows is stored (zeros too)
  int mask_mode;
  float scale;
  const float *logits;    // optional [B,H,Sq,ldl]: the forward's base-2 logits of the allowed pairs (isi_attn_args.logits):
  int ldl;                // the split kernels read them instead of forming Q K^T and the skewed band product again
  int g_from_kv;          // (kept logits, Cq = Ck = 1, band-only G) the KEY-stationary kernel stores dS into G -- lanes run
                          // along keys: 128-byte runs of a query's row -- and the query-stationary kernel only reads it back
                          // for dQ += dS K: no second exp / dO V^T / dS pass, no V tiles, no scatter
};

namespace {
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int QB = 128;        // queries (keys) per workgroup
constexpr int BAND = 160;      // rows of e a tile can touch
constexpr int RING = 256;      // rows of the band ring (power of two, >= BAND + 32)
constexpr float LOG2E = 1.4426950408889634f;
constexpr int SRLD = 65;       // row of the per-wave skew buffer
constexpr int TLD = 33;        // row of the per-wave transpose buffer (aliases the skew buffer)

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ float elem(const float4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

// D[b,h,i] = sum_d dO[i,b,h,d] * O[i,b,h,d].  HD / 4 lanes per (b, h, i) row: a row's 16-byte pieces are consecutive lanes
// (one thread per row read 256-byte rows with a stride of whole rows: 14.5 us for 2 x 16.8 MB at B8 H8 S1025 hd64)
__global__ __launch_bounds__(256) void attn_dsum_kernel(const AttnBwdKArgs p, int HD) {
  const int G = HD >> 2;                                   // 4, 8 or 16 lanes per row
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = t / G, c = (t % G) * 4;
  const int n = p.B * p.H * p.Sq;
  const int ic = min(idx, n - 1);                          // (whole lane groups stay active for the DPP sums)
  const int i = ic % p.Sq, bh = ic / p.Sq, h = bh % p.H, b = bh / p.H;
  const size_t off = (size_t)i * p.o_ss + (size_t)b * p.o_sb + (size_t)h * p.o_sh + c;
  const float4 a = *reinterpret_cast<const float4 *>(p.out + off), g = *reinterpret_cast<const float4 *>(p.dout + off);
  float s = (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
  s = G == 16 ? row16_sum(s) : G == 8 ? group8_sum(s) : group4_sum(s);
  if (idx < n && c == 0) p.dsum[idx] = s;
}

// ------------------------------------------------------------------ dQ and G
template <int HD>
__global__ __launch_bounds__(256) void rel_attention_bwd_q_kernel(const AttnBwdKArgs p) {
  constexpr int LDH = HD + 4;
  constexpr int NQ = HD / 8;
  constexpr int NDB = (HD + 31) / 32;
  constexpr int NKQ = (HD / 4 + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ks = smem;                    // [2][32][LDH]
  float *Vs = Ks + 2 * 32 * LDH;       // [2][32][LDH]
  float *Eb = Vs + 2 * 32 * LDH;       // [RING][LDH]
  float *Sr = Eb + RING * LDH;         // [4][32][SRLD]
  int *evk = reinterpret_cast<int *>(Sr + 4 * 32 * SRLD);  // [2][32] evk_max - event(key)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblk = p.mask_mode == 1 ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;  // heavy blocks first
  const int q0 = qblk * QB, qw0 = q0 + 32 * wave, qi = qw0 + ql;
  const bool has_e = p.e != nullptr;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dout), 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  // Q and dO fragments of this lane's query: quads (2s + half)
  float4 qf[NQ], dof[NQ];
#pragma unroll
  for (int s = 0; s < NQ; ++s) {
    const bool ok = qi < p.Sq;
    qf[s] = buf_load4(rq, ok ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + (2 * s + half) * 4) * 4u : OOB);
    dof[s] = buf_load4(rdo, ok ? (unsigned)(qi * p.o_ss + b * p.o_sb + h * p.o_sh + (2 * s + half) * 4) * 4u : OOB);
  }
  const int stat = (b * p.H + h) * p.Sq + qi;
  const float lse2 = (qi < p.Sq ? p.lse[stat] : 0.f) * LOG2E;
  const float dsum_i = qi < p.Sq ? p.dsum[stat] : 0.f;
  const int evq = qi / p.Cq;
  const int evq_w0 = qw0 / p.Cq, evq_b0 = q0 / p.Cq;

  f32x16 dQ[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQ[d][r] = 0.f;

  int k_begin = 0, k_end = p.Sk;
  if (p.mask_mode == 1) k_end = min(p.Sk, q0 + QB);
  if (p.mask_mode == 2) k_begin = (q0 / 32) * 32;

  const int srow = tid >> 3, squad = tid & 7;
  float *gbase = p.g + ((size_t)h * p.Sq * p.B + b) * p.Rp;   // row of query q: + q * gstride (rows ordered query-major)
  const size_t gstride = (size_t)p.B * p.Rp;
  float4 pk[NKQ], pv[NKQ], pe[NKQ];
  auto band0 = [&](int k0) { return evq_b0 - (k0 + 31) / p.Ck + p.Ek - 1; };
  auto prefetch = [&](int k0) {
    const int kj = k0 + srow;
    const bool ok = kj < p.Sk;
    const int r = band0(k0) + srow;
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
  auto commit = [&](int k0, int buf) {
    const int slot = (band0(k0) + srow) & (RING - 1);
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      if (qd < HD / 4) {
        *reinterpret_cast<float4 *>(Ks + (buf * 32 + srow) * LDH + qd * 4) = pk[i];
        *reinterpret_cast<float4 *>(Vs + (buf * 32 + srow) * LDH + qd * 4) = pv[i];
        if (has_e) *reinterpret_cast<float4 *>(Eb + slot * LDH + qd * 4) = pe[i];
      }
    }
    if (tid < 32) evk[buf * 32 + tid] = (k0 + 31) / p.Ck - (k0 + tid) / p.Ck;
  };

  if (k_begin < k_end) {
    prefetch(k_begin);
    commit(k_begin, 0);
    if (has_e) {
      const int rb = band0(k_begin);
      for (int row = 32 + srow; row < BAND; row += 32) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        for (int qd = squad; qd < HD / 4; qd += 8)
          *reinterpret_cast<float4 *>(Eb + (r & (RING - 1)) * LDH + qd * 4) =
              buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): fragments and prologue are in
  __syncthreads();
  const float scale2 = p.scale * LOG2E;

  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < k_end;
    if (more) prefetch(k0 + 32);
    const int rb = band0(k0);
    const float *Kb = Ks + buf * 32 * LDH, *Vb = Vs + buf * 32 * LDH;
    const int *evkb = evk + buf * 32;

    bool live = qw0 < p.Sq;
    if (p.mask_mode == 1) live = live && k0 <= qw0 + 31;
    if (p.mask_mode == 2) live = live && k0 + 31 >= qw0;
    if (live) {  // wave-uniform
      // ---- S^T = K Q^T and dP^T = V dO^T  (rows = keys, this lane's column = its query)
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
      {
        const float *kr = Kb + ql * LDH + half * 4;
        const float *vr = Vb + ql * LDH + half * 4;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
          const float4 kf = *reinterpret_cast<const float4 *>(kr + s * 8);
          const float4 vf = *reinterpret_cast<const float4 *>(vr + s * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(kf, e), elem(qf[s], e), sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(vf, e), elem(dof[s], e), dpacc, 0, 0, 0);
          }
        }
      }
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];

      float *sr = Sr + wave * 32 * SRLD + ql * SRLD;
      if (has_e) {
        const int wrow0 = rb + evq_w0 - evq_b0;
        const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;
        for (int t = 0; t < nt; ++t) {
          f32x16 racc;
#pragma unroll
          for (int r = 0; r < 16; ++r) racc[r] = 0.f;
          const float *er = Eb + ((wrow0 + 32 * t + ql) & (RING - 1)) * LDH + half * 4;
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
        wave_lds_sync();
        const int dq = evq - evq_w0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] += sr[dq + evkb[mfma_row(r, half)]];
        wave_lds_sync();
      }

      // ---- P, dS (scaled)
      bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < p.Sq;
      if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
      if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(sv[r] * scale2 - lse2);
          sv[r] = pr * (dpacc[r] - dsum_i) * p.scale;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = k0 + mfma_row(r, half);
          bool ok = kj < p.Sk && qi < p.Sq;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float s = sv[r] * scale2;
          if (p.mask && ok) s += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;
          const float pr = ok ? __builtin_amdgcn_exp2f(s - lse2) : 0.f;
          sv[r] = pr * (dpacc[r] - dsum_i) * p.scale;
        }
      }

      // ---- dQ^T += K^T dS^T
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int dcol = d * 32 + ql;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float kk = dcol < HD ? Kb[mfma_row(t, half) * LDH + dcol] : 0.f;
          dQ[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, sv[t], dQ[d], 0, 0, 0);
        }
      }

      // ---- G[i, r(i,j)] += dS[i,j]: transpose through LDS so that lanes run along the keys of one query
      if (has_e) {
        float *tb = Sr + wave * 32 * SRLD;  // [key][query], row TLD
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[mfma_row(r, half) * TLD + ql] = sv[r];
        wave_lds_sync();
        const int kev = (k0 + ql) / p.Ck;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int qq = 2 * it + half;
          const float val = tb[ql * TLD + qq];
          const int q = qw0 + qq;
          const int rho = __shfl(evq, qq) - kev + p.Ek - 1;   // lane qq holds event(query qw0 + qq): no division here
          const int col = rho - p.rho_lo;
          if (val != 0.f && rho >= 0 && rho < p.R && col >= 0 && col < p.Rp)
            unsafeAtomicAdd(gbase + (size_t)q * gstride + col, val);
        }
        wave_lds_sync();
      }
    }
    if (more) commit(k0 + 32, buf ^ 1);
    __syncthreads();
  }

  if (qi < p.Sq) {
    float *orow = p.dq + (size_t)qi * p.q_ss + (size_t)b * p.q_sb + (size_t)h * p.q_sh;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        if (dd < HD)
          *reinterpret_cast<float4 *>(orow + dd) =
              make_float4(dQ[d][4 * g], dQ[d][4 * g + 1], dQ[d][4 * g + 2], dQ[d][4 * g + 3]);
      }
  }
}

// ------------------------------------------------------------------ dK and dV
template <int HD>
__global__ __launch_bounds__(256) void rel_attention_bwd_kv_kernel(const AttnBwdKArgs p) {
  constexpr int LDH = HD + 4;
  constexpr int NQ = HD / 8;
  constexpr int NDB = (HD + 31) / 32;
  constexpr int NKQ = (HD / 4 + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Qs = smem;                    // [2][32][LDH]
  float *Gs = Qs + 2 * 32 * LDH;       // [2][32][LDH]   dO tiles
  float *Eb = Gs + 2 * 32 * LDH;       // [RING][LDH]
  float *Sr = Eb + RING * LDH;         // [4][32][SRLD]
  float *lse_s = Sr + 4 * 32 * SRLD;   // [2][32]  (base 2)
  float *dsum_s = lse_s + 64;          // [2][32]
  int *evq_s = reinterpret_cast<int *>(dsum_s + 64);  // [2][32] event(query) - first event of the tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  // anti-causal rows (j >= i): the last key blocks see the most queries -> launch them first
  const int kblk = p.mask_mode == 2 ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int k0b = kblk * QB, kw0 = k0b + 32 * wave, kj = kw0 + ql;
  const bool has_e = p.e != nullptr;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dout), 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  float4 kf[NQ], vf[NQ];
#pragma unroll
  for (int s = 0; s < NQ; ++s) {
    const bool ok = kj < p.Sk;
    kf[s] = buf_load4(rk, ok ? (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + (2 * s + half) * 4) * 4u : OOB);
    vf[s] = buf_load4(rv, ok ? (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + (2 * s + half) * 4) * 4u : OOB);
  }
  const int evk_max_b = (k0b + QB - 1) / p.Ck, evk_max_w = (kw0 + 31) / p.Ck;
  const int dkv = evk_max_w - kj / p.Ck;     // >= 0
  const int wrow0 = evk_max_b - evk_max_w;   // this wave's first band row (relative to the tile's band)

  f32x16 dK[NDB], dV[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dK[d][r] = 0.f; dV[d][r] = 0.f; }

  int q_begin = 0, q_end = p.Sq;
  if (p.mask_mode == 1) q_begin = (k0b / 32) * 32;          // j <= i
  if (p.mask_mode == 2) q_end = min(p.Sq, k0b + QB);        // j >= i

  const int srow = tid >> 3, squad = tid & 7;
  const int statb = (b * p.H + h) * p.Sq;
  float4 pq[NKQ], pg[NKQ], pe[NKQ];
  float plse = 0.f, pdsum = 0.f;
  // first table row of the band of the query tile that starts at q0
  auto band0 = [&](int q0) { return q0 / p.Cq - evk_max_b + p.Ek - 1; };
  auto prefetch = [&](int q0) {  // Q, dO rows of tile q0 and the 32 highest band rows of that tile
    const int qi = q0 + srow;
    const bool ok = qi < p.Sq;
    const int r = band0(q0) + (BAND - 32) + srow;
    const bool rok = has_e && r >= 0 && r < p.R;
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      const bool in = qd < HD / 4;
      pq[i] = buf_load4(rq, ok && in ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + qd * 4) * 4u : OOB);
      pg[i] = buf_load4(rdo, ok && in ? (unsigned)(qi * p.o_ss + b * p.o_sb + h * p.o_sh + qd * 4) * 4u : OOB);
      pe[i] = buf_load4(re, rok && in ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
    }
    if (tid < 32) {
      const int q = q0 + tid;
      plse = q < p.Sq ? p.lse[statb + q] : 0.f;
      pdsum = q < p.Sq ? p.dsum[statb + q] : 0.f;
    }
  };
  auto commit = [&](int q0, int buf) {
    const int slot = (band0(q0) + (BAND - 32) + srow) & (RING - 1);
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      const int qd = squad + 8 * i;
      if (qd < HD / 4) {
        *reinterpret_cast<float4 *>(Qs + (buf * 32 + srow) * LDH + qd * 4) = pq[i];
        *reinterpret_cast<float4 *>(Gs + (buf * 32 + srow) * LDH + qd * 4) = pg[i];
        if (has_e) *reinterpret_cast<float4 *>(Eb + slot * LDH + qd * 4) = pe[i];
      }
    }
    if (tid < 32) {
      lse_s[buf * 32 + tid] = plse * LOG2E;
      dsum_s[buf * 32 + tid] = pdsum;
      evq_s[buf * 32 + tid] = (q0 + tid) / p.Cq - q0 / p.Cq;
    }
  };

  if (q_begin < q_end) {
    prefetch(q_begin);
    commit(q_begin, 0);
    if (has_e) {
      const int rb = band0(q_begin);
      for (int row = srow; row < BAND - 32; row += 32) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        for (int qd = squad; qd < HD / 4; qd += 8)
          *reinterpret_cast<float4 *>(Eb + (r & (RING - 1)) * LDH + qd * 4) =
              buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): fragments and prologue are in
  __syncthreads();
  const float scale2 = p.scale * LOG2E;

  int buf = 0;
  for (int q0 = q_begin; q0 < q_end; q0 += 32, buf ^= 1) {
    const bool more = q0 + 32 < q_end;
    if (more) prefetch(q0 + 32);
    const int rb = band0(q0);
    const float *Qb = Qs + buf * 32 * LDH, *Gb = Gs + buf * 32 * LDH;
    const float *lseb = lse_s + buf * 32, *dsumb = dsum_s + buf * 32;
    const int *evqb = evq_s + buf * 32;

    bool live = kw0 < p.Sk;
    if (p.mask_mode == 1) live = live && q0 + 31 >= kw0;
    if (p.mask_mode == 2) live = live && q0 <= kw0 + 31;
    if (live) {  // wave-uniform
      // ---- S = Q K^T and dP = dO V^T  (rows = queries, this lane's column = its key)
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
      const float *qr = Qb + ql * LDH + half * 4;
      {
        const float *gr = Gb + ql * LDH + half * 4;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
          const float4 qfr = *reinterpret_cast<const float4 *>(qr + s * 8);
          const float4 gf = *reinterpret_cast<const float4 *>(gr + s * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(qfr, e), elem(kf[s], e), sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(gf, e), elem(vf[s], e), dpacc, 0, 0, 0);
          }
        }
      }
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];

      if (has_e) {
        // U = Q E_band^T : rows = queries, columns = band rows; entry (i, j) sits at column
        // (event(i) - first event of the tile) + (last event of the wave's keys - event(j))
        float *sw = Sr + wave * 32 * SRLD;
        const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;
        for (int t = 0; t < nt; ++t) {
          f32x16 racc;
#pragma unroll
          for (int r = 0; r < 16; ++r) racc[r] = 0.f;
          const float *er = Eb + ((rb + wrow0 + 32 * t + ql) & (RING - 1)) * LDH + half * 4;
#pragma unroll
          for (int s = 0; s < NQ; ++s) {
            const float4 ef = *reinterpret_cast<const float4 *>(er + s * 8);
            const float4 qfr = *reinterpret_cast<const float4 *>(qr + s * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              racc = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(qfr, e), elem(ef, e), racc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sw[mfma_row(r, half) * SRLD + 32 * t + ql] = racc[r];
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qrow = mfma_row(r, half);
          sv[r] += sw[qrow * SRLD + evqb[qrow] + dkv];
        }
        wave_lds_sync();
      }

      float pv[16];
      bool full = !p.mask && kw0 + 31 < p.Sk && q0 + 31 < p.Sq;
      if (p.mask_mode == 1) full = full && kw0 + 31 <= q0;
      if (p.mask_mode == 2) full = full && kw0 >= q0 + 31;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qrow = mfma_row(r, half);
          const float pr = __builtin_amdgcn_exp2f(sv[r] * scale2 - lseb[qrow]);
          pv[r] = pr;
          sv[r] = pr * (dpacc[r] - dsumb[qrow]) * p.scale;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qrow = mfma_row(r, half);
          const int qi = q0 + qrow;
          bool ok = kj < p.Sk && qi < p.Sq;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float s = sv[r] * scale2;
          if (p.mask && ok) s += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;
          const float pr = ok ? __builtin_amdgcn_exp2f(s - lseb[qrow]) : 0.f;
          pv[r] = pr;
          sv[r] = pr * (dpacc[r] - dsumb[qrow]) * p.scale;
        }
      }

      // ---- dV^T += dO^T P ,  dK^T += Q^T dS
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int dcol = d * 32 + ql;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = mfma_row(t, half);
          const float gg = dcol < HD ? Gb[row * LDH + dcol] : 0.f;
          const float qq = dcol < HD ? Qb[row * LDH + dcol] : 0.f;
          dV[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg, pv[t], dV[d], 0, 0, 0);
          dK[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qq, sv[t], dK[d], 0, 0, 0);
        }
      }
    }
    if (more) commit(q0 + 32, buf ^ 1);
    __syncthreads();
  }

  if (kj < p.Sk) {
    float *krow = p.dk + (size_t)kj * p.k_ss + (size_t)b * p.k_sb + (size_t)h * p.k_sh;
    float *vrow = p.dv + (size_t)kj * p.v_ss + (size_t)b * p.v_sb + (size_t)h * p.v_sh;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        if (dd < HD) {
          *reinterpret_cast<float4 *>(krow + dd) =
              make_float4(dK[d][4 * g], dK[d][4 * g + 1], dK[d][4 * g + 2], dK[d][4 * g + 3]);
          *reinterpret_cast<float4 *>(vrow + dd) =
              make_float4(dV[d][4 * g], dV[d][4 * g + 1], dV[d][4 * g + 2], dV[d][4 * g + 3]);
        }
      }
  }
}

// ================================================================== split-bf16 variants (fwd.precision = 1)
// Same mathematics with every contraction on the bf16 matrix pipe as a three-term split product
// (split_bf16.h) and fp32 everywhere else.  Layout of rel_attention_split_kernel (forward): workgroup =
// 8 waves in two groups, a group per tile of the streamed pair; operands pre-split once per workgroup into
// swizzled hi/lo bf16 planes in LDS -- row planes [row][HD] where the contraction runs over the head dim,
// transposed planes [d][32 rows] where it runs over the streamed rows; band of e in a 192-row ring.  The
// skew buffer holds 32 columns and the (up to) 64-column band is skewed in two passes, which is what lets
// all of this fit 160 KB.
namespace {
constexpr int RING_S = 192, BAND2_S = 192, SRL = 33;
__device__ __forceinline__ int ring_s(int r) {
  r %= RING_S;
  return r < 0 ? r + RING_S : r;
}
}  // namespace

// ------------------------------------------------------------------ dQ and G (split)
// ONE = true (precision 2): single-term bf16 products -- the lo.hi and hi.lo MFMAs of every product are left out
// (north_star's "MFMA bf16" mode; the lo planes are still staged: the kernels are not bound by them)
// phase timestamps of the query-stationary kernel (-DISI_MEASURE builds; tools/stamps_attention_bwd.py): workgroup 0, waves 0 and 4
#ifdef ISI_MEASURE
__device__ long long g_attn_q_stamps[512];
#define ISI_Q_STAMP(i_) do { if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && (i_) < 256) \
    g_attn_q_stamps[(wave >> 2) * 256 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_Q_STAMP(i_) do { } while (0)
#endif
template <int HD, bool ONE = false, bool SAVED = false>
__global__ __launch_bounds__(512) void rel_attention_bwd_q_split_kernel(const AttnBwdKArgs p) {
  constexpr int NKB = HD / 16, NSL = HD / 8, RPB = 128 / HD, NDB = (HD + 31) / 32, VR = NDB * 32;
  constexpr int NQD = HD / 4, NKQ = (HD / 4 + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Kp = reinterpret_cast<unsigned short *>(smem);   // [tile 2][plane 2][32][HD]   K rows
  unsigned short *Ktp = Kp + 2 * 2 * 32 * HD;                      // [tile 2][plane 2][VR][32]   K transposed
  unsigned short *Vp = Ktp + 2 * 2 * VR * 32;                      // [tile 2][plane 2][32][HD]   V rows
  unsigned short *Ep = Vp + 2 * 2 * 32 * HD;                       // [plane 2][RING_S][HD]
  float *Sr = reinterpret_cast<float *>(Ep + 2 * RING_S * HD);     // [8][32][SRL]
  int *evk = reinterpret_cast<int *>(Sr + 8 * 32 * SRL);           // [2][32]
  auto swz = [](int row, int slot) { return (slot ^ ((row / RPB) % NSL)) * 8; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  // (tile, head, batch) from the 1-D launch: one XCD's L2 per (batch, head) pair, heaviest tiles first (xcd_tile); causal
  // masks: the ragged query block is block 0, where the key range is shortest (as in the forward kernel)
  const int nqb = p.nqb;
  int qt, pair;
  if (!xcd_tile(nqb, p.H * p.B, p.mask_mode != 0, qt, pair)) return;
  const int h = pair % p.H, b = pair / p.H;
  const int qblk = p.mask_mode == 1 ? nqb - 1 - qt : qt;
  const int rag = (p.mask_mode == 1 && p.Cq == 1 && nqb * QB >= p.Sq) ? p.Sq % QB : 0;
  const int q0 = rag ? (qblk ? rag + (qblk - 1) * QB : 0) : qblk * QB;
  const int q_end = (rag && qblk == 0) ? rag : p.Sq;           // first row beyond this block's valid ones
  const int qw0 = q0 + 32 * wq, qi = qw0 + ql;
  const bool has_e = p.e != nullptr;
  constexpr bool saved = SAVED;                // logits kept by the forward (p.logits): no Q K^T, no band product, no skew here
  const bool band = has_e && !saved;           // the band of e is staged only when the relative logits are recomputed
  const bool from_g = saved && p.g_from_kv;    // dS is read back from G (written by the key-stationary kernel)

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dout), 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  // Q and dO fragments of this lane's query, split once: k-block t holds dims 16 t + 8 half + 0..7
  s16x8_t qh[NKB], qlo[NKB], doh[NKB], dol[NKB];
#pragma unroll
  for (int t = 0; t < NKB; ++t) {
    const bool ok = qi < q_end;
    const unsigned oq = ok ? (unsigned)(qi * p.q_ss + b * p.q_sb + h * p.q_sh + 16 * t + 8 * half) * 4u : OOB;
    const unsigned od = ok ? (unsigned)(qi * p.o_ss + b * p.o_sb + h * p.o_sh + 16 * t + 8 * half) * 4u : OOB;
    uint2 h0, l0, h1, l1;
    split_f4(buf_load4(rq, oq), h0, l0);
    split_f4(buf_load4(rq, ok ? oq + 16u : OOB), h1, l1);
    qh[t] = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
    qlo[t] = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
    split_f4(buf_load4(rdo, od), h0, l0);
    split_f4(buf_load4(rdo, ok ? od + 16u : OOB), h1, l1);
    doh[t] = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
    dol[t] = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
  }
  const int stat = (b * p.H + h) * p.Sq + qi;
  const float lse2 = (qi < q_end ? p.lse[stat] : 0.f) * LOG2E;
  const float dsum_i = qi < q_end ? p.dsum[stat] : 0.f;
  const int evq = qi / p.Cq;
  const int evq_w0 = qw0 / p.Cq, evq_b0 = q0 / p.Cq;

  f32x16 dQ[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQ[d][r] = 0.f;

  int k_begin = 0, k_end = p.Sk;
  if (p.mask_mode == 1) k_end = min(p.Sk, q0 + QB);
  if (p.mask_mode == 2) k_begin = (q0 / 32) * 32;

  // staging roles: a 4 keys x 4 dims block of K (threads [0, 256)) or V ([256, 512)); band row 32 st + srow
  // (the opaque-index form of the dK / dV kernel was measured here: 256 -> 251 VGPRs, same time)
  const int kind = tid >> 8, bidx = tid & 255;
  const int bqd = bidx % NQD, bkg = (bidx / NQD) & 7, btile = bidx / (8 * NQD);
  const bool blk_on = btile < 2;
  const int st = tid >> 8, srow = (tid >> 3) & 31, squad = tid & 7;
  float *gbase = p.g + ((size_t)h * p.Sq * p.B + b) * p.Rp;   // row of query q: + q * gstride (rows ordered query-major)
  const size_t gstride = (size_t)p.B * p.Rp;
  const bool unique_rho = p.Cq == 1 && p.Ck == 1;
  float4 pb[4], pe[NKQ];
  auto band0 = [&](int k) { return evq_b0 - (k + 31) / p.Ck + p.Ek - 1; };
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kj = k0 + 32 * btile + 4 * bkg + j;
      const bool ok = blk_on && kj < p.Sk;
      const unsigned ko = (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + bqd * 4) * 4u;
      const unsigned vo = (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + bqd * 4) * 4u;
      pb[j] = kind == 0 ? buf_load4(rk, ok ? ko : OOB) : buf_load4(rv, ok && !from_g ? vo : OOB);
    }
    const int r = band0(k0 + 32) + 32 * st + srow;
    const bool rok = band && r >= 0 && r < p.R;
    if constexpr (!saved) {
#pragma unroll
      for (int i = 0; i < NKQ; ++i) {
        const int qd = squad + 8 * i;
        pe[i] = buf_load4(re, rok && qd < NQD ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
  };
  auto put_e = [&](int slot, int qd, const float4 v) {
    uint2 hi, lo;
    split_f4(v, hi, lo);
    const int o = slot * HD + swz(slot, qd >> 1) + (qd & 1) * 4;
    *reinterpret_cast<uint2 *>(Ep + o) = hi;
    *reinterpret_cast<uint2 *>(Ep + o + RING_S * HD) = lo;
  };
  auto commit = [&](int k0) {
    if (blk_on) {
      unsigned short *rows = (kind == 0 ? Kp : Vp) + (btile * 2) * 32 * HD;
      if (!(saved && kind == 0))      // (K rows feed Q K^T only)
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // row planes: 8 bytes (4 dims) per key
        const int row = 4 * bkg + j;
        uint2 hi, lo;
        split_f4(pb[j], hi, lo);
        const int o = row * HD + swz(row, bqd >> 1) + (bqd & 1) * 4;
        *reinterpret_cast<uint2 *>(rows + o) = hi;
        *reinterpret_cast<uint2 *>(rows + o + 32 * HD) = lo;
      }
      if (kind == 0) {                // K transposed: per dim the block's 4 keys as one 8-byte unit
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = 4 * bqd + e;
          uint2 hi, lo;
          split2(elem(pb[0], e), elem(pb[1], e), hi.x, lo.x);
          split2(elem(pb[2], e), elem(pb[3], e), hi.y, lo.y);
          const int o = ((btile * 2) * VR + d) * 32 + ((bkg ^ ((d >> 2) & 7)) * 4);
          *reinterpret_cast<uint2 *>(Ktp + o) = hi;
          *reinterpret_cast<uint2 *>(Ktp + o + VR * 32) = lo;
        }
      }
    }
    if (band) {
      const int slot = ring_s(band0(k0 + 32) + 32 * st + srow);
#pragma unroll
      for (int i = 0; i < NKQ; ++i)
        if (squad + 8 * i < NQD) put_e(slot, squad + 8 * i, pe[i]);
    }
    if (tid < 64) {
      const int kt = k0 + (tid & 32);
      evk[tid] = (kt + 31) / p.Ck - (kt + (tid & 31)) / p.Ck;
    }
  };

  if (VR > HD) {
    for (int i = tid; i < 2 * 2 * VR * 32 / 2; i += 512) reinterpret_cast<unsigned *>(Ktp)[i] = 0u;
    __syncthreads();
  }
  if (k_begin < k_end) {
    prefetch(k_begin);
    commit(k_begin);
    if (band) {
      const int rb = band0(k_begin + 32);
      for (int row = 64 + (tid >> 3); row < BAND2_S; row += 64) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        const int slot = ring_s(r);
        for (int qd = squad; qd < NQD; qd += 8)
          put_e(slot, qd, buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const float scale2 = p.scale * LOG2E;
  const unsigned short *Kb = Kp + (grp * 2) * 32 * HD, *Ktb = Ktp + (grp * 2) * VR * 32, *Vb = Vp + (grp * 2) * 32 * HD;
  const int *evkb = evk + grp * 32;
  float *sr = Sr + wave * 32 * SRL + ql * SRL;
  float *tb = Sr + wave * 32 * SRL;
  const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;

  ISI_Q_STAMP(0);
  for (int kp = k_begin; kp < k_end; kp += 64) {
    const bool more = kp + 64 < k_end;
    const int sb_ = 4 + 8 * ((kp - k_begin) >> 6);
    ISI_Q_STAMP(sb_);
    if (more) prefetch(kp + 64);
    ISI_Q_STAMP(sb_ + 1);
    const int k0 = kp + 32 * grp;
    const int rb = band0(k0);

    bool live = qw0 < q_end && k0 < k_end;
    if (p.mask_mode == 1) live = live && k0 <= qw0 + 31;
    if (p.mask_mode == 2) live = live && k0 + 31 >= qw0;
    if (live) {  // wave-uniform
      f32x16 acc;
      float sv[16];
      typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
      bool full = !p.mask && k0 + 31 < p.Sk && qw0 + 31 < q_end;
      if (p.mask_mode == 1) full = full && k0 + 31 <= qw0;
      if (p.mask_mode == 2) full = full && k0 >= qw0 + 31;
      if (from_g) {
        // ---- dS of this lane's query from its row of G: keys 8 g + 4 half + 0..3 are 4 consecutive columns, descending
        const float *grow = gbase + (size_t)min(qi, p.Sq - 1) * gstride + (qi - k0 - 4 * half + p.Ek - 1 - p.rho_lo);
        if (full) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_u v = *reinterpret_cast<const f32x4_u *>(grow - 8 * g - 3);
            sv[4 * g + 3] = v.x; sv[4 * g + 2] = v.y; sv[4 * g + 1] = v.z; sv[4 * g] = v.w;
          }
        } else {      // (pairs the mask forbids have no element of their own in a band-only G)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2), kj = k0 + 4 * half + jj;
            bool ok = kj < p.Sk && qi < q_end;
            if (p.mask_mode == 1) ok = ok && kj <= qi;
            if (p.mask_mode == 2) ok = ok && kj >= qi;
            sv[r] = ok ? grow[-jj] : 0.f;
          }
        }
      } else {
      if (saved) {
        // ---- the forward's logits of this lane's query: keys 8 g + 4 half + 0..3 of the tile (already in units of exp2)
        const float *lrow = p.logits + (((size_t)b * p.H + h) * p.Sq + min(qi, p.Sq - 1)) * p.ldl + k0 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4 *>(lrow + 8 * g);
          sv[4 * g] = v.x; sv[4 * g + 1] = v.y; sv[4 * g + 2] = v.z; sv[4 * g + 3] = v.w;
        }
      } else {
      // ---- S^T = K Q^T
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const int o = ql * HD + swz(ql, 2 * t + half);
        const s16x8_t kh = *reinterpret_cast<const s16x8_t *>(Kb + o);
        const s16x8_t kl = *reinterpret_cast<const s16x8_t *>(Kb + o + 32 * HD);
        if constexpr (!ONE) acc = ISI_MFB(kl, qh[t], acc);
        if constexpr (!ONE) acc = ISI_MFB(kh, qlo[t], acc);
        acc = ISI_MFB(kh, qh[t], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = acc[r];

      // ---- relative logits: band GEMM, skewed through LDS 32 columns at a time
      if (has_e) {
        const int wrow0 = rb + evq_w0 - evq_b0;
        const int dq = evq - evq_w0;
        for (int tbi = 0; tbi < nt; ++tbi) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const int slot = ring_s(wrow0 + 32 * tbi + ql);
#pragma unroll
          for (int t = 0; t < NKB; ++t) {
            const int o = slot * HD + swz(slot, 2 * t + half);
            const s16x8_t eh = *reinterpret_cast<const s16x8_t *>(Ep + o);
            const s16x8_t el = *reinterpret_cast<const s16x8_t *>(Ep + o + RING_S * HD);
            if constexpr (!ONE) acc = ISI_MFB(el, qh[t], acc);
            if constexpr (!ONE) acc = ISI_MFB(eh, qlo[t], acc);
            acc = ISI_MFB(eh, qh[t], acc);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sr[mfma_row(r, half)] = acc[r];
          wave_lds_sync();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = dq + evkb[mfma_row(r, half)] - 32 * tbi;
            const float v = sr[idx & 31];
            sv[r] += (unsigned)idx < 32u ? v : 0.f;
          }
          wave_lds_sync();
        }
      }

      }
      // ---- P
      const float sc2 = saved ? 1.f : scale2;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_exp2f(sv[r] * sc2 - lse2);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = k0 + mfma_row(r, half);
          bool ok = kj < p.Sk && qi < q_end;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float s = sv[r] * sc2;
          if (p.mask && ok && !saved) s += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;     // (kept logits include the mask)
          sv[r] = ok ? __builtin_amdgcn_exp2f(s - lse2) : 0.f;
        }
      }
      // ---- dP^T = V dO^T ;  dS = P (dP - D) scale
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const int o = ql * HD + swz(ql, 2 * t + half);
        const s16x8_t vh = *reinterpret_cast<const s16x8_t *>(Vb + o);
        const s16x8_t vl = *reinterpret_cast<const s16x8_t *>(Vb + o + 32 * HD);
        if constexpr (!ONE) acc = ISI_MFB(vl, doh[t], acc);
        if constexpr (!ONE) acc = ISI_MFB(vh, dol[t], acc);
        acc = ISI_MFB(vh, doh[t], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sv[r] * (acc[r] - dsum_i) * p.scale;

      }
      ISI_Q_STAMP(sb_ + 2);      // dS in registers
      // ---- dQ^T += K^T dS^T
      s16x8_t sh[2], sl[2];
      split_acc16(sv, sh, sl);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int drow = d * 32 + ql;
        const int sx = (drow >> 2) & 7;
        const unsigned short *kr = Ktb + drow * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint2 h0 = *reinterpret_cast<const uint2 *>(kr + (((4 * t + half) ^ sx) * 4));
          const uint2 h1 = *reinterpret_cast<const uint2 *>(kr + (((4 * t + 2 + half) ^ sx) * 4));
          const uint2 l0 = *reinterpret_cast<const uint2 *>(kr + VR * 32 + (((4 * t + half) ^ sx) * 4));
          const uint2 l1 = *reinterpret_cast<const uint2 *>(kr + VR * 32 + (((4 * t + 2 + half) ^ sx) * 4));
          const s16x8_t kth = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
          const s16x8_t ktl = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
          if constexpr (!ONE) dQ[d] = ISI_MFB(ktl, sh[t], dQ[d]);
          if constexpr (!ONE) dQ[d] = ISI_MFB(kth, sl[t], dQ[d]);
          dQ[d] = ISI_MFB(kth, sh[t], dQ[d]);
        }
      }

      ISI_Q_STAMP(sb_ + 3);      // dQ issued
      // ---- G[i, r(i,j)] += dS[i,j]
      if (from_g) {
        // (G was written by the key-stationary kernel)
      } else if (has_e && unique_rho && full) {
        // one channel per event on both sides and every pair of the tile allowed: the lane's 16 keys of its query are 4 runs
        // of 4 consecutive table rows (descending) of the query's own row of G -- four 16-byte stores (4-byte aligned),
        // no transpose through LDS, no per-element address arithmetic
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        float *grow = gbase + (size_t)qi * gstride + (qi - k0 - 4 * half + p.Ek - 1 - p.rho_lo);   // column of key k0 + 4 half
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_u v = {sv[4 * g + 3], sv[4 * g + 2], sv[4 * g + 1], sv[4 * g]};
          *reinterpret_cast<f32x4_u *>(grow - 8 * g - 3) = v;
        }
      } else if (has_e) {   // transpose through LDS so that lanes run along the keys of one query
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[mfma_row(r, half) * SRL + ql] = sv[r];
        wave_lds_sync();
        const int kev = (k0 + ql) / p.Ck;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int qq = 2 * it + half;
          const float val = tb[ql * SRL + qq];
          const int q = qw0 + qq;
          const int rho = __shfl(evq, qq) - kev + p.Ek - 1;   // lane qq holds event(query qw0 + qq): no division here
          const int col = rho - p.rho_lo;
          bool wr = val != 0.f;
          if (p.g_band_only) {     // (unique_rho) the band itself is not zeroed: store every pair the mask allows
            const int key = k0 + ql;
            wr = q < q_end && key < p.Sk && (p.mask_mode == 1 ? key <= q : p.mask_mode == 2 ? key >= q : true);
          }
          if (wr && rho >= 0 && rho < p.R && col >= 0 && col < p.Rp) {
            // one channel per event on both sides: (query, key) -> table row is one-to-one, the element of the zeroed G is
            // written exactly once -- a plain store instead of a read-modify-write in the L2
            if (unique_rho) gbase[(size_t)q * gstride + col] = val;
            else unsafeAtomicAdd(gbase + (size_t)q * gstride + col, val);
          }
        }
        wave_lds_sync();
      }
    }
    ISI_Q_STAMP(sb_ + 4);
    __syncthreads();
    ISI_Q_STAMP(sb_ + 5);
    if (more) commit(kp + 64);
    ISI_Q_STAMP(sb_ + 6);
    __syncthreads();
    ISI_Q_STAMP(sb_ + 7);
  }
  ISI_Q_STAMP(1);

  // ---- add the two groups' partial dQ (group 1 -> LDS -> group 0) and store
  float *mg = smem;
  constexpr int MGW = NDB * 16 * 64;
  if (grp == 1) {
    float *dst = mg + wq * MGW + lane;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = dQ[d][r];
  }
  __syncthreads();
  if (grp == 1) return;
  if (qi < q_end) {
    const float *src = mg + wq * MGW + lane;
    float *orow = p.dq + (size_t)qi * p.q_ss + (size_t)b * p.q_sb + (size_t)h * p.q_sh;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        if (dd < HD)
          *reinterpret_cast<float4 *>(orow + dd) =
              make_float4(dQ[d][4 * g] + src[(d * 16 + 4 * g) * 64], dQ[d][4 * g + 1] + src[(d * 16 + 4 * g + 1) * 64],
                          dQ[d][4 * g + 2] + src[(d * 16 + 4 * g + 2) * 64], dQ[d][4 * g + 3] + src[(d * 16 + 4 * g + 3) * 64]);
      }
  }
}

// phase timestamps of the key-stationary kernel (-DISI_MEASURE builds; tools/stamps_attention_bwd.py): the first workgroup
// of XCD 0 (under a causal mask: the heaviest block of its pair), waves 0 and 4, 12 stamps per step
#ifdef ISI_MEASURE
__device__ long long g_attn_kv_stamps[512];
#define ISI_KV_STAMP(i_) do { if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && (i_) < 256) \
    g_attn_kv_stamps[(wave >> 2) * 256 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_KV_STAMP(i_) do { } while (0)
#endif
// ------------------------------------------------------------------ dK and dV (split)
// Key-stationary: wave wq of either group owns keys k0b + 32 wq .. + 31 (K, V fragments split once, in
// registers as B operands; dK^T / dV^T in accumulators); group g walks the query tiles of parity g and the
// two partial results are added at the end.  Q and dO tiles are staged as row planes (S = Q K^T,
// dP = dO V^T, U = Q E^T contract over the head dim) AND transposed planes (dV^T += dO^T P,
// dK^T += Q^T dS contract over the queries).
template <int HD, bool ONE = false, bool SAVED = false>
__global__ __launch_bounds__(512) void rel_attention_bwd_kv_split_kernel(const AttnBwdKArgs p) {
  constexpr int NKB = HD / 16, NSL = HD / 8, RPB = 128 / HD, NDB = (HD + 31) / 32, VR = NDB * 32;
  constexpr int NQD = HD / 4, NKQ = (HD / 4 + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short *Qp = reinterpret_cast<unsigned short *>(smem);   // [tile 2][plane 2][32][HD]   Q rows
  unsigned short *Qtp = Qp + 2 * 2 * 32 * HD;                      // [tile 2][plane 2][VR][32]   Q transposed
  unsigned short *Gp = Qtp + 2 * 2 * VR * 32;                      // dO rows
  unsigned short *Gtp = Gp + 2 * 2 * 32 * HD;                      // dO transposed
  unsigned short *Ep = Gtp + 2 * 2 * VR * 32;                      // [plane 2][RING_S][HD]
  float *Sr = reinterpret_cast<float *>(Ep + 2 * RING_S * HD);     // [8][32][SRL]
  float *lse_s = Sr + 8 * 32 * SRL;                                // [2][32]  (base 2)
  float *dsum_s = lse_s + 64;                                      // [2][32]
  int *evq_s = reinterpret_cast<int *>(dsum_s + 64);               // [2][32] event(query) - first event of the tile
  auto swz = [](int row, int slot) { return (slot ^ ((row / RPB) % NSL)) * 8; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int ql = lane & 31, half = lane >> 5;
  // (tile, head, batch) from the 1-D launch: one XCD's L2 per (batch, head) pair, heaviest tiles first (xcd_tile)
  const int nkb = p.nkb;
  int kt, pair;
  if (!xcd_tile(nkb, p.H * p.B, p.mask_mode != 0, kt, pair)) return;
  const int h = pair % p.H, b = pair / p.H;
  const int kblk = p.mask_mode == 2 ? nkb - 1 - kt : kt;
  // anti-causal masks: the ragged key block (Sk % QB keys) would see EVERY query -- this kernel's heaviest block for one
  // key of a 1025-key sequence; it is block 0 instead, where the query range is shortest (the mirror image of the
  // query-stationary kernels under a causal mask; the band logic takes any origin when Ck = 1)
  const int rag = (p.mask_mode == 2 && p.Ck == 1 && nkb * QB >= p.Sk) ? p.Sk % QB : 0;
  const int k0b = rag ? (kblk ? rag + (kblk - 1) * QB : 0) : kblk * QB;
  const int k_lim = (rag && kblk == 0) ? rag : p.Sk;            // first key beyond this block's valid ones
  const int kw0 = k0b + 32 * wq, kj = kw0 + ql;
  const bool has_e = p.e != nullptr;
  constexpr bool saved = SAVED;                // logits kept by the forward (p.logits): no Q K^T, no band product, no skew here
  const bool band = has_e && !saved;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.q), 0, p.q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.k), 0, p.k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.v), 0, p.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dout), 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_e ? p.e : p.q), 0, has_e ? p.e_bytes : 4u, 0x00020000);

  s16x8_t kh[NKB], kl[NKB], vh[NKB], vl[NKB];
#pragma unroll
  for (int t = 0; t < NKB; ++t) {
    const bool ok = kj < k_lim;
    const unsigned ok_ = ok ? (unsigned)(kj * p.k_ss + b * p.k_sb + h * p.k_sh + 16 * t + 8 * half) * 4u : OOB;
    const unsigned ov = ok ? (unsigned)(kj * p.v_ss + b * p.v_sb + h * p.v_sh + 16 * t + 8 * half) * 4u : OOB;
    uint2 h0, l0, h1, l1;
    split_f4(buf_load4(rk, ok_), h0, l0);
    split_f4(buf_load4(rk, ok ? ok_ + 16u : OOB), h1, l1);
    kh[t] = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
    kl[t] = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
    split_f4(buf_load4(rv, ov), h0, l0);
    split_f4(buf_load4(rv, ok ? ov + 16u : OOB), h1, l1);
    vh[t] = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
    vl[t] = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
  }
  const int evk_max_b = (k0b + QB - 1) / p.Ck, evk_max_w = (kw0 + 31) / p.Ck;
  const int dkv = evk_max_w - kj / p.Ck;     // >= 0
  const int wrow0 = evk_max_b - evk_max_w;   // this wave's first band row (relative to the tile's band)

  f32x16 dK[NDB], dV[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dK[d][r] = 0.f; dV[d][r] = 0.f; }

  int q_begin = 0, q_end = p.Sq;
  if (p.mask_mode == 1) q_begin = (k0b / 32) * 32;          // j <= i
  if (p.mask_mode == 2) q_end = min(p.Sq, k0b + QB);        // j >= i

  // staging roles: a 4 queries x 4 dims block of Q (threads [0, 256)) or dO ([256, 512)); band row 32 st + srow.
  // They are derived from an OPAQUE copy of the thread index inside prefetch / commit (a few VALU per call): as loop
  // invariants they and the address terms built on them were parked in registers across the whole tile step.
  const int statb = (b * p.H + h) * p.Sq;
  float4 pb[4], pe[NKQ];
  float plse = 0.f, pdsum = 0.f;
  // first table row of the band of the query tile that starts at q
  auto band0 = [&](int q) { return q / p.Cq - evk_max_b + p.Ek - 1; };
#define ISI_KV_ROLES                                                                                   \
  int tid_o = tid;                                                                                     \
  asm volatile("" : "+v"(tid_o));                                                                      \
  const int kind = __builtin_amdgcn_readfirstlane(tid >> 8), bidx = tid_o & 255;    /* (wave-uniform: as a vector value the buffer descriptor chosen by it was one too -- a waterfall loop per load and a wait for ALL of them right behind) */ \
  const int bqd = bidx % NQD, bkg = (bidx / NQD) & 7, btile = bidx / (8 * NQD);                        \
  const bool blk_on = btile < 2;                                                                       \
  const int st = kind, srow = (tid_o >> 3) & 31, squad = tid_o & 7;                                    \
  (void)st; (void)srow; (void)squad; (void)kind; (void)bqd; (void)bkg; (void)blk_on
  auto prefetch = [&](int q0) __attribute__((always_inline)) {  // Q / dO blocks of the pair at q0 and the 64 highest rows of its band
    ISI_KV_ROLES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qj = q0 + 32 * btile + 4 * bkg + j;
      const bool ok = blk_on && qj < p.Sq;
      const unsigned qo = (unsigned)(qj * p.q_ss + b * p.q_sb + h * p.q_sh + bqd * 4) * 4u;
      const unsigned go = (unsigned)(qj * p.o_ss + b * p.o_sb + h * p.o_sh + bqd * 4) * 4u;
      pb[j] = kind == 0 ? buf_load4(rq, ok ? qo : OOB) : buf_load4(rdo, ok ? go : OOB);
    }
    const int r = band0(q0) + (BAND2_S - 64) + 32 * st + srow;
    const bool rok = band && r >= 0 && r < p.R;
    if constexpr (!saved) {
#pragma unroll
      for (int i = 0; i < NKQ; ++i) {
        const int qd = squad + 8 * i;
        pe[i] = buf_load4(re, rok && qd < NQD ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB);
      }
    }
    if (wave == 0) {     // (no zero-fill of the two registers in front of a conditional load: the compiler then waits for
                         // every older memory operation -- the previous step's stores into G included -- before that write)
      const int q = min(q0 + lane, p.Sq - 1);
      plse = p.lse[statb + q];
      pdsum = p.dsum[statb + q];
    }
  };
  auto put_e = [&](int slot, int qd, const float4 v) __attribute__((always_inline)) {
    uint2 hi, lo;
    split_f4(v, hi, lo);
    const int o = slot * HD + swz(slot, qd >> 1) + (qd & 1) * 4;
    *reinterpret_cast<uint2 *>(Ep + o) = hi;
    *reinterpret_cast<uint2 *>(Ep + o + RING_S * HD) = lo;
  };
  // With kept logits neither the Q rows nor the ring of e are staged: their LDS holds a SECOND stage of Q^T / dO / dO^T (and
  // of the per-query statistics), so that the next step's tiles are written while this step's are read -- one barrier per
  // step instead of barrier, commit, barrier (2.7 k of a 9.5 k-cycle step, tools/stamps_attention_bwd.py)
  unsigned short *Qtp1 = Qp, *Gp1 = Ep, *Gtp1 = Ep + 2 * 2 * 32 * HD;
  float *stat1 = reinterpret_cast<float *>(Gtp1 + 2 * 2 * VR * 32);          // lse [64], dsum [64], evq [64]
  constexpr bool two = SAVED && (size_t)2 * RING_S * HD >= (size_t)2 * 2 * 32 * HD + 2 * 2 * VR * 32 + 3 * 64 * 2;   // (hd 16: it does not fit)
  auto commit = [&](int q0, int stage = 0) __attribute__((always_inline)) {
    ISI_KV_ROLES;
    if (blk_on) {
      // (stage 1 as element offsets from the first stage's arrays: a select between LDS pointers made them generic ones and
      // sent the kernel's argument block to scratch)
      const int o_qt = stage ? (int)(Qtp1 - Qtp) : 0, o_g = stage ? (int)(Gp1 - Gp) : 0, o_gt = stage ? (int)(Gtp1 - Gtp) : 0;
      unsigned short *rows = (kind == 0 ? Qp : Gp + o_g) + (btile * 2) * 32 * HD;
      unsigned short *cols = (kind == 0 ? Qtp + o_qt : Gtp + o_gt) + (btile * 2) * VR * 32;
      if (!(saved && kind == 0))      // (Q rows feed Q K^T and the band product only)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * bkg + j;
        uint2 hi, lo;
        split_f4(pb[j], hi, lo);
        const int o = row * HD + swz(row, bqd >> 1) + (bqd & 1) * 4;
        *reinterpret_cast<uint2 *>(rows + o) = hi;
        *reinterpret_cast<uint2 *>(rows + o + 32 * HD) = lo;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 4 * bqd + e;
        uint2 hi, lo;
        split2(elem(pb[0], e), elem(pb[1], e), hi.x, lo.x);
        split2(elem(pb[2], e), elem(pb[3], e), hi.y, lo.y);
        const int o = d * 32 + ((bkg ^ ((d >> 2) & 7)) * 4);
        *reinterpret_cast<uint2 *>(cols + o) = hi;
        *reinterpret_cast<uint2 *>(cols + o + VR * 32) = lo;
      }
    }
    if (band) {
      const int slot = ring_s(band0(q0) + (BAND2_S - 64) + 32 * st + srow);
#pragma unroll
      for (int i = 0; i < NKQ; ++i)
        if (squad + 8 * i < NQD) put_e(slot, squad + 8 * i, pe[i]);
    }
    if (tid < 64) {
      const int qt = q0 + (tid & 32);
      const int o_st = stage ? (int)(stat1 - lse_s) : 0;           // stage 1: lse at stat1, dsum at stat1 + 64
      lse_s[o_st + tid] = q0 + tid < p.Sq ? plse * LOG2E : 0.f;
      lse_s[o_st + 64 + tid] = q0 + tid < p.Sq ? pdsum : 0.f;
      if (!stage) evq_s[tid] = (q0 + tid) / p.Cq - qt / p.Cq;      // (the band product's; not used with kept logits)
    }
  };

#undef ISI_KV_ROLES
  if (VR > HD) {
    for (int i = tid; i < 2 * 2 * VR * 32 / 2; i += 512) {
      reinterpret_cast<unsigned *>(Qtp)[i] = 0u;
      reinterpret_cast<unsigned *>(Gtp)[i] = 0u;
    }
    __syncthreads();
  }
  if (q_begin < q_end) {
    prefetch(q_begin);
    commit(q_begin);
    if (band) {
      const int rb = band0(q_begin);
      const int squad = tid & 7;
      for (int row = tid >> 3; row < BAND2_S - 64; row += 64) {
        const int r = rb + row;
        const bool ok = r >= 0 && r < p.R;
        const int slot = ring_s(r);
        for (int qd = squad; qd < NQD; qd += 8)
          put_e(slot, qd, buf_load4(re, ok ? (unsigned)((h * p.R + r) * HD + qd * 4) * 4u : OOB));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const float scale2 = p.scale * LOG2E;
  const unsigned short *Qb = Qp + (grp * 2) * 32 * HD, *Qtb = Qtp + (grp * 2) * VR * 32;
  const unsigned short *Gb = Gp + (grp * 2) * 32 * HD, *Gtb = Gtp + (grp * 2) * VR * 32;
  const float *lseb = lse_s + grp * 32, *dsumb = dsum_s + grp * 32;
  const int *evqb = evq_s + grp * 32;      // (used by the band product only: not with kept logits)
  if (two && q_begin + 64 < q_end) prefetch(q_begin + 64);      // (two stages: the second step's tiles are on their way)
  float *sw = Sr + wave * 32 * SRL;
  const int nt = (31 / p.Cq + 31 / p.Ck) < 32 ? 1 : 2;

  // The kept logits of a tile (units of exp2).  The accumulator layout wants a lane's key and 16 query rows: 16 dword loads
  // per lane -- and the vector-memory path takes a wave's instruction at 4 lanes per clock whatever its width, so that the
  // 32 dword loads / stores of a tile step (logits in, dS out) were ~4 k of its 12 k cycles (tools/stamps_attention_bwd.py).
  // The tile is requested as four 16-byte pieces of query rows per lane, ONE STEP AHEAD (asked for at the head of their own
  // step every wave of the workgroup -- they run in step between barriers -- sat out the round trip together), and turned in
  // the wave's LDS buffer when its step comes.
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
  f32x4_u lvn[4];
  auto load_logits = [&](int q0_) __attribute__((always_inline)) {
    const float *lt = p.logits + ((size_t)b * p.H + h) * p.Sq * p.ldl + min(kw0 + 4 * (lane & 7), p.ldl - 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) lvn[i] = *reinterpret_cast<const f32x4_u *>(lt + (size_t)min(q0_ + (lane >> 3) + 8 * i, p.Sq - 1) * p.ldl);
  };
  if (saved && q_begin < q_end) load_logits(q_begin + 32 * grp);
  ISI_KV_STAMP(0);
  for (int qp = q_begin; qp < q_end; qp += 64) {
    const bool more = qp + 64 < q_end;
    const int sb_ = 4 + 12 * ((qp - q_begin) >> 6);
    ISI_KV_STAMP(sb_);
    if (two) {
      const int stg = ((qp - q_begin) >> 6) & 1;
      if (more) commit(qp + 64, stg ^ 1);           // requested a step ago; that stage was last read a step ago (barrier below)
      if (qp + 128 < q_end) prefetch(qp + 128);
      Qtb = Qtp + (stg ? (int)(Qtp1 - Qtp) : 0) + (grp * 2) * VR * 32;
      Gb = Gp + (stg ? (int)(Gp1 - Gp) : 0) + (grp * 2) * 32 * HD;
      Gtb = Gtp + (stg ? (int)(Gtp1 - Gtp) : 0) + (grp * 2) * VR * 32;
      lseb = lse_s + (stg ? (int)(stat1 - lse_s) : 0) + grp * 32;
      dsumb = lseb + 64;
    } else if (more) prefetch(qp + 64);
    ISI_KV_STAMP(sb_ + 1);
    const int q0 = qp + 32 * grp;
    const int rb = band0(q0);

    bool live = kw0 < k_lim && q0 < q_end;
    if (p.mask_mode == 1) live = live && q0 + 31 >= kw0;
    if (p.mask_mode == 2) live = live && q0 <= kw0 + 31;
    if (saved) {     // this step's logits go to the wave's LDS buffer, the next step's are asked for (live tile or not)
      float *wt = sw + (lane >> 3) * SRL + 4 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wt[8 * i * SRL] = lvn[i].x; wt[8 * i * SRL + 1] = lvn[i].y; wt[8 * i * SRL + 2] = lvn[i].z; wt[8 * i * SRL + 3] = lvn[i].w;
      }
      load_logits(min(qp + 64, p.Sq - 1) + 32 * grp);      // (clamped rows: the last step re-reads valid memory)
      wave_lds_sync();
    }
    if (live) {  // wave-uniform
      // The lane index is made opaque per iteration: the dozens of LDS offsets derived from it are loop-invariant and
      // would otherwise be hoisted into registers the accumulators and fragments need (the three-term kernel spilled
      // 49 VGPRs to scratch; scratch reloads queue behind the kernel's stores -- DESIGN.md section 0).
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int ql = lane_o & 31, half = lane_o >> 5;
      f32x16 acc;
      float sv[16];
      if (saved) {
        // ---- the forward's logits: this lane's key, the 16 query rows of its accumulator registers
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sw[mfma_row(r, half) * SRL + ql];
        wave_lds_sync();
      } else {
      // ---- S = Q K^T  (rows = queries, this lane's column = its key)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const int o = ql * HD + swz(ql, 2 * t + half);
        const s16x8_t qfh = *reinterpret_cast<const s16x8_t *>(Qb + o);
        const s16x8_t qfl = *reinterpret_cast<const s16x8_t *>(Qb + o + 32 * HD);
        if constexpr (!ONE) acc = ISI_MFB(qfl, kh[t], acc);
        if constexpr (!ONE) acc = ISI_MFB(qfh, kl[t], acc);
        acc = ISI_MFB(qfh, kh[t], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = acc[r];

      if (has_e) {
        // U = Q E_band^T : rows = queries, columns = band rows; entry (i, j) sits at column
        // (event(i) - first event of the tile) + (last event of the wave's keys - event(j))
        for (int tbi = 0; tbi < nt; ++tbi) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const int slot = ring_s(rb + wrow0 + 32 * tbi + ql);
#pragma unroll
          for (int t = 0; t < NKB; ++t) {
            const int o = slot * HD + swz(slot, 2 * t + half);
            const int oq = ql * HD + swz(ql, 2 * t + half);
            const s16x8_t eh = *reinterpret_cast<const s16x8_t *>(Ep + o);
            const s16x8_t el = *reinterpret_cast<const s16x8_t *>(Ep + o + RING_S * HD);
            const s16x8_t qfh = *reinterpret_cast<const s16x8_t *>(Qb + oq);
            const s16x8_t qfl = *reinterpret_cast<const s16x8_t *>(Qb + oq + 32 * HD);
            if constexpr (!ONE) acc = ISI_MFB(qfl, eh, acc);
            if constexpr (!ONE) acc = ISI_MFB(qfh, el, acc);
            acc = ISI_MFB(qfh, eh, acc);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sw[mfma_row(r, half) * SRL + ql] = acc[r];
          wave_lds_sync();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = evqb[mfma_row(r, half)] + dkv - 32 * tbi;
            const float v = sw[mfma_row(r, half) * SRL + (idx & 31)];
            sv[r] += (unsigned)idx < 32u ? v : 0.f;
          }
          wave_lds_sync();
        }
      }

      }
      // ---- P
      bool full = !p.mask && kw0 + 31 < k_lim && q0 + 31 < p.Sq;
      if (p.mask_mode == 1) full = full && kw0 + 31 <= q0;
      if (p.mask_mode == 2) full = full && kw0 >= q0 + 31;
      const float sc2 = saved ? 1.f : scale2;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_exp2f(sv[r] * sc2 - lseb[mfma_row(r, half)]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qrow = mfma_row(r, half);
          const int qi = q0 + qrow;
          bool ok = kj < k_lim && qi < p.Sq;
          if (p.mask_mode == 1) ok = ok && kj <= qi;
          if (p.mask_mode == 2) ok = ok && kj >= qi;
          float s = sv[r] * sc2;
          if (p.mask && ok && !saved) s += p.mask[(size_t)qi * p.Sk + kj] * LOG2E;
          sv[r] = ok ? __builtin_amdgcn_exp2f(s - lseb[qrow]) : 0.f;
        }
      }
      ISI_KV_STAMP(sb_ + 2);      // P formed (logits arrived)
      // ---- dV^T += dO^T P   (A = dO transposed, k-slots = queries in accumulator order)
      s16x8_t sh[2], sl[2];
      split_acc16(sv, sh, sl);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int drow = d * 32 + ql;
        const int sx = (drow >> 2) & 7;
        const unsigned short *gr = Gtb + drow * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint2 h0 = *reinterpret_cast<const uint2 *>(gr + (((4 * t + half) ^ sx) * 4));
          const uint2 h1 = *reinterpret_cast<const uint2 *>(gr + (((4 * t + 2 + half) ^ sx) * 4));
          const uint2 l0 = *reinterpret_cast<const uint2 *>(gr + VR * 32 + (((4 * t + half) ^ sx) * 4));
          const uint2 l1 = *reinterpret_cast<const uint2 *>(gr + VR * 32 + (((4 * t + 2 + half) ^ sx) * 4));
          const s16x8_t gh = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
          const s16x8_t gl = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
          if constexpr (!ONE) dV[d] = ISI_MFB(gl, sh[t], dV[d]);
          if constexpr (!ONE) dV[d] = ISI_MFB(gh, sl[t], dV[d]);
          dV[d] = ISI_MFB(gh, sh[t], dV[d]);
        }
      }
      ISI_KV_STAMP(sb_ + 3);      // dV issued
      // ---- dP = dO V^T ;  dS = P (dP - D) scale
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const int o = ql * HD + swz(ql, 2 * t + half);
        const s16x8_t gh = *reinterpret_cast<const s16x8_t *>(Gb + o);
        const s16x8_t gl = *reinterpret_cast<const s16x8_t *>(Gb + o + 32 * HD);
        if constexpr (!ONE) acc = ISI_MFB(gl, vh[t], acc);
        if constexpr (!ONE) acc = ISI_MFB(gh, vl[t], acc);
        acc = ISI_MFB(gh, vh[t], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sv[r] * (acc[r] - dsumb[mfma_row(r, half)]) * p.scale;
      ISI_KV_STAMP(sb_ + 4);      // dS formed
      if (saved && p.g_from_kv) {
        // ---- G[i, i - j + Ek - 1] = dS[i, j]: for one query (register) the lanes' keys are consecutive columns, descending
        // -- 128-byte runs; element (i, j) sits at g0[i (row stride + 1)]
        float *g0 = p.g + ((size_t)h * p.Sq * p.B + b) * p.Rp + (p.Ek - 1 - p.rho_lo - kj);
        const size_t gs1 = (size_t)p.B * p.Rp + 1;
        if (full) {
          // whole tile: turned in the wave's LDS buffer so that a lane holds 4 consecutive keys of a query = 4 consecutive
          // (descending) columns of its row of G: four 16-byte stores per lane instead of 16 dword stores
#pragma unroll
          for (int r = 0; r < 16; ++r) sw[mfma_row(r, half) * SRL + ql] = sv[r];
          wave_lds_sync();
          const int r8 = lane_o >> 3, c4 = lane_o & 7;
          float *gq = p.g + ((size_t)h * p.Sq * p.B + b) * p.Rp + (p.Ek - 1 - p.rho_lo - kw0) - 4 * c4 - 3;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float *s4 = sw + (r8 + 8 * i) * SRL + 4 * c4;
            const f32x4_u v = {s4[3], s4[2], s4[1], s4[0]};
            *reinterpret_cast<f32x4_u *>(gq + (size_t)(q0 + r8 + 8 * i) * gs1) = v;
          }
          wave_lds_sync();
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qi = q0 + mfma_row(r, half);
            bool ok = kj < k_lim && qi < p.Sq;
            if (p.mask_mode == 1) ok = ok && kj <= qi;
            if (p.mask_mode == 2) ok = ok && kj >= qi;
            if (ok) g0[(size_t)qi * gs1] = sv[r];
          }
        }
      }
      ISI_KV_STAMP(sb_ + 5);      // G stores issued
      // ---- dK^T += Q^T dS
      split_acc16(sv, sh, sl);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int drow = d * 32 + ql;
        const int sx = (drow >> 2) & 7;
        const unsigned short *qr = Qtb + drow * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint2 h0 = *reinterpret_cast<const uint2 *>(qr + (((4 * t + half) ^ sx) * 4));
          const uint2 h1 = *reinterpret_cast<const uint2 *>(qr + (((4 * t + 2 + half) ^ sx) * 4));
          const uint2 l0 = *reinterpret_cast<const uint2 *>(qr + VR * 32 + (((4 * t + half) ^ sx) * 4));
          const uint2 l1 = *reinterpret_cast<const uint2 *>(qr + VR * 32 + (((4 * t + 2 + half) ^ sx) * 4));
          const s16x8_t qth = __builtin_bit_cast(s16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
          const s16x8_t qtl = __builtin_bit_cast(s16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
          if constexpr (!ONE) dK[d] = ISI_MFB(qtl, sh[t], dK[d]);
          if constexpr (!ONE) dK[d] = ISI_MFB(qth, sl[t], dK[d]);
          dK[d] = ISI_MFB(qth, sh[t], dK[d]);
        }
      }
    }
    ISI_KV_STAMP(sb_ + 6);        // dK issued
    __syncthreads();
    ISI_KV_STAMP(sb_ + 7);
    if (!two) {
      if (more) commit(qp + 64);
      ISI_KV_STAMP(sb_ + 8);
      __syncthreads();
    } else ISI_KV_STAMP(sb_ + 8);
    ISI_KV_STAMP(sb_ + 9);
  }
  ISI_KV_STAMP(1);

  // ---- add the two groups' partial results (group 1 -> LDS -> group 0) and store
  float *mg = smem;
  constexpr int MGW = 2 * NDB * 16 * 64;
  if (grp == 1) {
    float *dst = mg + wq * MGW + lane;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dst[(d * 16 + r) * 64] = dK[d][r];
        dst[((NDB + d) * 16 + r) * 64] = dV[d][r];
      }
  }
  __syncthreads();
  if (grp == 1) return;
  if (kj < k_lim) {
    const float *src = mg + wq * MGW + lane;
    float *krow = p.dk + (size_t)kj * p.k_ss + (size_t)b * p.k_sb + (size_t)h * p.k_sh;
    float *vrow = p.dv + (size_t)kj * p.v_ss + (size_t)b * p.v_sb + (size_t)h * p.v_sh;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        if (dd < HD) {
          const float *sk = src + (d * 16 + 4 * g) * 64, *sv_ = src + ((NDB + d) * 16 + 4 * g) * 64;
          *reinterpret_cast<float4 *>(krow + dd) = make_float4(dK[d][4 * g] + sk[0], dK[d][4 * g + 1] + sk[64],
                                                               dK[d][4 * g + 2] + sk[128], dK[d][4 * g + 3] + sk[192]);
          *reinterpret_cast<float4 *>(vrow + dd) = make_float4(dV[d][4 * g] + sv_[0], dV[d][4 * g + 1] + sv_[64],
                                                               dV[d][4 * g + 2] + sv_[128], dV[d][4 * g + 3] + sv_[192]);
        }
      }
  }
}

// packed GEMM operand of one head's table rows [lo, lo + n), transposed: w[d][c] = e[h][lo + c][d], zero padded to Kpad
__global__ void pack_rel_T_kernel(const float *__restrict__ e, float *__restrict__ out, int H, int R, int HD,
                                  int Kpad, int lo, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * HD * Kpad) return;
  const int c = (int)(i % Kpad);
  const int d = (int)((i / Kpad) % HD);
  const int h = (int)(i / ((int64_t)Kpad * HD));
  out[i] = c < n ? e[((int64_t)h * R + lo + c) * HD + d] : 0.f;
}

// d_rel[h][r][d] = dw[h][r - lo][d] for table rows in [lo, lo + n), 0 elsewhere (dw rows padded to Rp, columns to Kp)
__global__ void unpack_drel_kernel(const float *__restrict__ dw, float *__restrict__ out, int H, int R, int HD,
                                   int Rp, int Kp, int lo, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * R * HD) return;
  const int d = (int)(i % HD);
  const int r = (int)((i / HD) % R);
  const int h = (int)(i / ((int64_t)HD * R));
  const int c = r - lo;
  out[i] = c >= 0 && c < n ? dw[((int64_t)h * Rp + c) * Kp + d] : 0.f;
}

namespace {
struct BwdLayout {
  int Rp, Kp;                                // padded table rows (x4) / GEMM K of the table (x32) ; Kp: head dim x32
  size_t dsum, g, wT, dw, wg, total;         // float offsets into the workspace
  size_t wg_floats;
};
// Table rows a (query, key) pair can select: causal / anti-causal self-attention (Cq == Ck) only ever
// reaches the upper / lower half of the table, which halves G and the two GEMMs over it.
void rho_range(const isi_attn_args *g, int *lo, int *n) {
  const int R = g->rel_rows;
  *lo = 0; *n = R;
  if (g->Cq == g->Ck && !g->dense_mask) {
    if (g->mask_mode == 1) { *lo = std::min(std::max(g->Ek - 1, 0), R - 1); *n = R - *lo; }   // j <= i: rho >= Ek - 1
    if (g->mask_mode == 2) { *n = std::max(1, std::min(R, g->Ek)); }                           // j >= i: rho <= Ek - 1
  }
}
BwdLayout bwd_layout(int B, int H, int Sq, int R, int HD) {   // R = number of table rows G covers
  BwdLayout L;
  L.Rp = (int)round_up((size_t)std::max(R, 1), kBK);      // (a whole number of K chunks: G is a GEMM operand as it stands)
  L.Kp = (int)round_up((size_t)HD, kBK);
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += round_up(n, 64); return o; };
  L.dsum = take((size_t)B * H * Sq);
  if (R > 0) {
    L.g = take((size_t)H * B * Sq * L.Rp);
    L.wT = take((size_t)H * HD * round_up((size_t)L.Rp, kBK));
    L.dw = take((size_t)H * L.Rp * L.Kp);
    L.wg_floats = conv_wgrad_batched_workspace_floats(L.Rp, HD, B * Sq, 1, H);
    L.wg = take(L.wg_floats);
  } else {
    L.g = L.wT = L.dw = L.wg = 0; L.wg_floats = 0;
  }
  L.total = off;
  return L;
}
// ---- the one or two rows / keys beyond the last full 128-row block (attention_tail_rows: the prior's sequences are
// 1024 codes + a start row).  As a stationary block of the kernels above such a row costs as much as 128 of them -- a
// third round of workgroups on the 256 CUs for one row per (batch, head): 1252 vs 1007 us for the dense 1025 x 1025
// backward.  These two kernels do the same arithmetic for ONE query row (dQ, its row of G) resp. ONE key (dK, dV) per
// workgroup in exact fp32: G = HD / 4 lanes per streamed row, one coalesced 16-byte load per lane and operand.
constexpr int TAIL_THREADS = 512, TAIL_U = 8;     // 512 / (HD / 4) x 8 streamed rows in flight per workgroup (256 at HD 64)

template <int HD>
__device__ __forceinline__ void attn_bwd_tail_row(const AttnBwdKArgs &p, float *part, int i) {
  constexpr int G = HD / 4, RPP = TAIL_THREADS / G, U = TAIL_U, NW = TAIL_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G, gl = tid % G;
  const int h = blockIdx.x, b = blockIdx.y;
  const float4 qq = *reinterpret_cast<const float4 *>(p.q + (size_t)i * p.q_ss + (size_t)b * p.q_sb + (size_t)h * p.q_sh + gl * 4);
  const float4 dd = *reinterpret_cast<const float4 *>(p.dout + (size_t)i * p.o_ss + (size_t)b * p.o_sb + (size_t)h * p.o_sh + gl * 4);
  const int stat = (b * p.H + h) * p.Sq + i;
  const float lse = p.lse[stat], dsum = p.dsum[stat];
  const int Sk = p.mask_mode == 1 ? min(p.Sk, i + 1) : p.Sk;          // (no anti-causal / additive masks on this path)
  const int evq = i / p.Cq;
  const float *kb = p.k + (size_t)b * p.k_sb + (size_t)h * p.k_sh + gl * 4;
  const float *vb = p.v + (size_t)b * p.v_sb + (size_t)h * p.v_sh + gl * 4;
  const float *eb = p.e ? p.e + (size_t)h * p.R * HD + gl * 4 : nullptr;
  float *grow = p.g ? p.g + (((size_t)h * p.Sq + i) * p.B + b) * p.Rp : nullptr;
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = grp; j0 < Sk; j0 += U * RPP) {
    float4 kk[U], vv[U], ee[U];
    int rho[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * RPP, jc = j < Sk ? j : j0;
      kk[u] = *reinterpret_cast<const float4 *>(kb + (size_t)jc * p.k_ss);
      vv[u] = *reinterpret_cast<const float4 *>(vb + (size_t)jc * p.v_ss);
      ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      rho[u] = evq - jc / p.Ck + p.Ek - 1;
      if (eb && rho[u] >= 0 && rho[u] < p.R) ee[u] = *reinterpret_cast<const float4 *>(eb + (size_t)rho[u] * HD);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * RPP;
      const float4 kq = make_float4(kk[u].x + ee[u].x, kk[u].y + ee[u].y, kk[u].z + ee[u].z, kk[u].w + ee[u].w);
      float sdot = (qq.x * kq.x + qq.y * kq.y) + (qq.z * kq.z + qq.w * kq.w);
      float pdot = (dd.x * vv[u].x + dd.y * vv[u].y) + (dd.z * vv[u].z + dd.w * vv[u].w);
      sdot = G == 16 ? row16_sum(sdot) : G == 8 ? group8_sum(sdot) : group4_sum(sdot);
      pdot = G == 16 ? row16_sum(pdot) : G == 8 ? group8_sum(pdot) : group4_sum(pdot);
      const float pj = j < Sk ? __expf(sdot * p.scale - lse) : 0.f;
      const float ds = pj * (pdot - dsum) * p.scale;
      dq.x += ds * kk[u].x; dq.y += ds * kk[u].y; dq.z += ds * kk[u].z; dq.w += ds * kk[u].w;
      if (grow && gl == 0 && j < Sk && (ds != 0.f || p.g_band_only)) {
        const int col = rho[u] - p.rho_lo;
        if (rho[u] >= 0 && rho[u] < p.R && col >= 0 && col < p.Rp) grow[col] = ds;   // (Ck = 1: one key per column of the zeroed row)
      }
    }
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    dq.x += __shfl_xor(dq.x, o); dq.y += __shfl_xor(dq.y, o); dq.z += __shfl_xor(dq.z, o); dq.w += __shfl_xor(dq.w, o);
  }
  if (lane < G) *reinterpret_cast<float4 *>(part + wave * HD + lane * 4) = dq;
  __syncthreads();
  if (tid < HD) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) acc += part[w * HD + tid];
    p.dq[(size_t)i * p.q_ss + (size_t)b * p.q_sb + (size_t)h * p.q_sh + tid] = acc;
  }
}

template <int HD>
__device__ __forceinline__ void attn_bwd_tail_key(const AttnBwdKArgs &p, float *part, int j) {
  constexpr int G = HD / 4, RPP = TAIL_THREADS / G, U = TAIL_U, NW = TAIL_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G, gl = tid % G;
  const int h = blockIdx.x, b = blockIdx.y;
  const float4 kk = *reinterpret_cast<const float4 *>(p.k + (size_t)j * p.k_ss + (size_t)b * p.k_sb + (size_t)h * p.k_sh + gl * 4);
  const float4 vv = *reinterpret_cast<const float4 *>(p.v + (size_t)j * p.v_ss + (size_t)b * p.v_sb + (size_t)h * p.v_sh + gl * 4);
  const int statb = (b * p.H + h) * p.Sq;
  const int i_begin = p.mask_mode == 1 ? j : 0;                      // causal: queries i >= j
  const int evk = j / p.Ck;
  const float *qb = p.q + (size_t)b * p.q_sb + (size_t)h * p.q_sh + gl * 4;
  const float *db = p.dout + (size_t)b * p.o_sb + (size_t)h * p.o_sh + gl * 4;
  const float *eb = p.e ? p.e + (size_t)h * p.R * HD + gl * 4 : nullptr;
  float4 dk = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = i_begin + grp; i0 < p.Sq; i0 += U * RPP) {
    float4 qv[U], dd[U], ee[U];
    float lse[U], dsum[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPP, ic = i < p.Sq ? i : i0;
      qv[u] = *reinterpret_cast<const float4 *>(qb + (size_t)ic * p.q_ss);
      dd[u] = *reinterpret_cast<const float4 *>(db + (size_t)ic * p.o_ss);
      lse[u] = p.lse[statb + ic];
      dsum[u] = p.dsum[statb + ic];
      ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int rho = ic / p.Cq - evk + p.Ek - 1;
      if (eb && rho >= 0 && rho < p.R) ee[u] = *reinterpret_cast<const float4 *>(eb + (size_t)rho * HD);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPP;
      const float4 kq = make_float4(kk.x + ee[u].x, kk.y + ee[u].y, kk.z + ee[u].z, kk.w + ee[u].w);
      float sdot = (qv[u].x * kq.x + qv[u].y * kq.y) + (qv[u].z * kq.z + qv[u].w * kq.w);
      float pdot = (dd[u].x * vv.x + dd[u].y * vv.y) + (dd[u].z * vv.z + dd[u].w * vv.w);
      sdot = G == 16 ? row16_sum(sdot) : G == 8 ? group8_sum(sdot) : group4_sum(sdot);
      pdot = G == 16 ? row16_sum(pdot) : G == 8 ? group8_sum(pdot) : group4_sum(pdot);
      const float pi = i < p.Sq ? __expf(sdot * p.scale - lse[u]) : 0.f;
      const float ds = pi * (pdot - dsum[u]) * p.scale;
      dv.x += pi * dd[u].x; dv.y += pi * dd[u].y; dv.z += pi * dd[u].z; dv.w += pi * dd[u].w;
      dk.x += ds * qv[u].x; dk.y += ds * qv[u].y; dk.z += ds * qv[u].z; dk.w += ds * qv[u].w;
      if (p.g_from_kv && gl == 0 && i < p.Sq)      // this key's column of dS for the query-stationary kernel (Cq = Ck = 1)
        p.g[(((size_t)h * p.Sq + i) * p.B + b) * p.Rp + (i - j + p.Ek - 1 - p.rho_lo)] = ds;
    }
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    dk.x += __shfl_xor(dk.x, o); dk.y += __shfl_xor(dk.y, o); dk.z += __shfl_xor(dk.z, o); dk.w += __shfl_xor(dk.w, o);
    dv.x += __shfl_xor(dv.x, o); dv.y += __shfl_xor(dv.y, o); dv.z += __shfl_xor(dv.z, o); dv.w += __shfl_xor(dv.w, o);
  }
  if (lane < G) {
    *reinterpret_cast<float4 *>(part + wave * HD + lane * 4) = dk;
    *reinterpret_cast<float4 *>(part + (NW + wave) * HD + lane * 4) = dv;
  }
  __syncthreads();
  if (tid < HD) {
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { ak += part[w * HD + tid]; av += part[(NW + w) * HD + tid]; }
    p.dk[(size_t)j * p.k_ss + (size_t)b * p.k_sb + (size_t)h * p.k_sh + tid] = ak;
    p.dv[(size_t)j * p.v_ss + (size_t)b * p.v_sb + (size_t)h * p.v_sh + tid] = av;
  }
}

__global__ __launch_bounds__(256) void attn_zero_f4_kernel(float4 *__restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// G zeroed only where it is read but not written: with Cq = Ck = 1 every (query, key) pair has its own element, the
// query-stationary kernel (and the one-row kernel) store the whole band of a row -- zeros included -- and both products
// over G read a band of chunks / queries around it (GemmExtra, WgradBand): what has to be zero are the MARGINS of the
// band, kMargin columns on either side: a 128-row / 128-column tile of either product plus its chunk rounding reaches at
// most kBandTileMax + kBandChunk - 2 = 158 columns beyond a row's own band (288 until the bound was tied to the kernels'
// tile constants: the margins were 150 MB of zeros per call, more than the band itself).
// One wave per row of G; rows = (head, query, batch).
constexpr int kMargin = 160;
static_assert(kMargin >= kBandTileMax + kBandChunk - 2, "margins of G cover a banded tile of either product + its chunk rounding");
__global__ __launch_bounds__(256) void attn_zero_margins_kernel(float *__restrict__ g, long rows, int Sq, int B, int Rp,
                                                                int lo_slope, int lo_base, int hi_slope, int hi_base) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int i = (int)((row / B) % Sq);
  const int lo = lo_slope * i + lo_base, hi = hi_slope * i + hi_base;      // the row's band [lo, hi]
  float4 *gr = reinterpret_cast<float4 *>(g + row * Rp);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  // quads [a0, a1) on the left of the band, [b0, b1) on its right (a quad that reaches into the band is harmless:
  // this launch runs before the kernels that store the band)
  const int a0 = max(lo - kMargin, 0) >> 2, a1 = (max(min(lo, Rp), 0) + 3) >> 2;
  const int b0 = max(min(hi + 1, Rp), 0) >> 2, b1 = (max(min(hi + 1 + kMargin, Rp), 0) + 3) >> 2;
  for (int q = a0 + lane; q < a1; q += 64) gr[q] = z;
  for (int q = b0 + lane; q < b1; q += 64) gr[q] = z;
}

// one launch for both roles (they are independent): blockIdx.z < tail_q: query row Sq - tail_q + z, else a key
template <int HD>
__global__ __launch_bounds__(TAIL_THREADS) void attn_bwd_tail_kernel(const AttnBwdKArgs p, int tail_q, int tail_k) {
  __shared__ __attribute__((aligned(16))) float part[2 * (TAIL_THREADS / 64) * HD];
  const int z = blockIdx.z;
  if (z < tail_q) attn_bwd_tail_row<HD>(p, part, p.Sq - tail_q + z);
  else attn_bwd_tail_key<HD>(p, part, p.Sk - tail_k + (z - tail_q));
}

template <int HD>
int launch_bwd_tails(const AttnBwdKArgs &a, int tail_q, int tail_k, hipStream_t stream) {
  hipLaunchKernelGGL(attn_bwd_tail_kernel<HD>, dim3(a.H, a.B, tail_q + tail_k), dim3(TAIL_THREADS), 0, stream, a, tail_q, tail_k);
  return check_launch("attn_bwd_tail");
}

int64_t span(int64_t S, int64_t ss, int64_t B, int64_t sb, int64_t H, int64_t sh, int64_t hd) {
  return (S - 1) * ss + (B - 1) * sb + (H - 1) * sh + hd;
}
}  // namespace

size_t rel_attention_bwd_workspace_floats(const isi_attn_args *g) {
  if (!g || g->B <= 0 || g->H <= 0 || g->Sq <= 0) return 0;
  int lo = 0, n = 0;
  if (g->rel_embeddings && g->rel_rows > 0) rho_range(g, &lo, &n);
  return bwd_layout(g->B, g->H, g->Sq, n, g->head_dim).total;
}

// which: 1 = the key-stationary kernel (dK, dV), 2 = the query-stationary one (dQ, G), 3 = both
template <int HD, bool ONE, bool SAVED = false>
static int launch_bwd_split(const AttnBwdKArgs &a, hipStream_t stream, int which = 3) {
  if constexpr (!SAVED) {
    if (a.logits) return launch_bwd_split<HD, ONE, true>(a, stream, which);
  }
  auto kq = rel_attention_bwd_q_split_kernel<HD, ONE, SAVED>;
  auto kkv = rel_attention_bwd_kv_split_kernel<HD, ONE, SAVED>;
  constexpr int VR = ((HD + 31) / 32) * 32;
  constexpr size_t smem_q = (size_t)(2 * (2 * 2 * 32 * HD) + 2 * 2 * VR * 32 + 2 * RING_S * HD) * sizeof(unsigned short) +
                            (size_t)(8 * 32 * SRL + 64) * sizeof(float);
  constexpr size_t smem_kv = (size_t)(2 * (2 * 2 * 32 * HD) + 2 * (2 * 2 * VR * 32) + 2 * RING_S * HD) * sizeof(unsigned short) +
                             (size_t)(8 * 32 * SRL + 192) * sizeof(float);
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv) != hipSuccess)
      return check_launch("hipFuncSetAttribute(rel_attention_bwd_split)");
    attr_set.mark();
  }
  const double pairs = (double)a.Sq * a.Sk * (a.mask_mode ? 0.5 : 1.0) * a.H * a.B;
  if (which & 1) {
    prof::Scope scope(prof::K_REL_ATTENTION_BWD, 2.0 * pairs * HD * (a.e ? 5 : 4),
                      4.0 * a.B * a.H * HD * (3.0 * a.Sq + 4.0 * a.Sk), stream);
    ISI_PROF_LAUNCH(scope, kkv, dim3(xcd_grid(a.nkb, a.H * a.B)), dim3(512), smem_kv, stream, a);
    const int rc = check_launch("rel_attention_bwd_kv_split");
    if (rc) return rc;
  }
  if (which & 2) {
    prof::Scope scope(prof::K_REL_ATTENTION_BWD, 2.0 * pairs * HD * (a.e ? 4 : 3),
                      4.0 * a.B * a.H * HD * (3.0 * a.Sq + 2.0 * a.Sk), stream);
    ISI_PROF_LAUNCH(scope, kq, dim3(xcd_grid(a.nqb, a.H * a.B)), dim3(512), smem_q, stream, a);
    return check_launch("rel_attention_bwd_q_split");
  }
  return ISI_OK;
}

template <int HD>
static int launch_bwd(const AttnBwdKArgs &a, hipStream_t stream) {
  auto kq = rel_attention_bwd_q_kernel<HD>;
  auto kkv = rel_attention_bwd_kv_kernel<HD>;
  constexpr size_t smem = (size_t)((128 + RING) * (HD + 4) + 4 * 32 * SRLD + 192) * sizeof(float);
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(rel_attention_bwd)");
    attr_set.mark();
  }
  const double pairs = (double)a.Sq * a.Sk * (a.mask_mode ? 0.5 : 1.0) * a.H * a.B;
  {
    prof::Scope scope(prof::K_REL_ATTENTION_BWD, 2.0 * pairs * HD * (a.e ? 5 : 4),
                      4.0 * a.B * a.H * HD * (3.0 * a.Sq + 4.0 * a.Sk), stream);
    ISI_PROF_LAUNCH(scope, kkv, dim3((a.Sk + QB - 1) / QB, a.H, a.B), dim3(256), smem, stream, a);
  }
  int rc = check_launch("rel_attention_bwd_kv");
  if (rc) return rc;
  {
    prof::Scope scope(prof::K_REL_ATTENTION_BWD, 2.0 * pairs * HD * (a.e ? 4 : 3),
                      4.0 * a.B * a.H * HD * (3.0 * a.Sq + 2.0 * a.Sk), stream);
    ISI_PROF_LAUNCH(scope, kq, dim3((a.Sq + QB - 1) / QB, a.H, a.B), dim3(256), smem, stream, a);
  }
  return check_launch("rel_attention_bwd_q");
}

int rel_attention_bwd_f32(const isi_attn_bwd_args *ga, hipStream_t stream) {
  if (!ga) return invalid("rel_attention_bwd: null pointer");
  const isi_attn_args *g = &ga->fwd;
  if (!g->q || !g->k || !g->v || !g->out || !g->lse || !ga->d_out || !ga->dq || !ga->dk || !ga->dv || !ga->workspace)
    return invalid("rel_attention_bwd: null pointer");
  if (g->Sq <= 0 || g->Sk <= 0 || g->B <= 0 || g->H <= 0) return invalid("rel_attention_bwd: bad shape");
  if (g->Cq <= 0 || g->Ck <= 0 || g->Ek <= 0) return invalid("rel_attention_bwd: bad event layout");
  if (g->mask_mode < 0 || g->mask_mode > 2) return invalid("rel_attention_bwd: bad mask mode");
  if (g->head_dim != 16 && g->head_dim != 32 && g->head_dim != 64)
    return unsupported("rel_attention_bwd: head_dim must be 16, 32 or 64");
  if (127 / g->Cq + 31 / g->Ck + 1 > BAND || 31 / g->Cq + 127 / g->Ck + 1 > BAND)
    return unsupported("rel_attention_bwd: band too wide");
  const bool has_e = g->rel_embeddings != nullptr;
  if (has_e && (g->rel_rows <= 0 || !ga->d_rel)) return invalid("rel_attention_bwd: rel_rows / d_rel missing");
  const int64_t lim = (int64_t)1 << 30;
  const int HD = g->head_dim;
  const int64_t eq = span(g->Sq, g->q_ss, g->B, g->q_sb, g->H, g->q_sh, HD);
  const int64_t ek = span(g->Sk, g->k_ss, g->B, g->k_sb, g->H, g->k_sh, HD);
  const int64_t ev = span(g->Sk, g->v_ss, g->B, g->v_sb, g->H, g->v_sh, HD);
  const int64_t eo = span(g->Sq, g->o_ss, g->B, g->o_sb, g->H, g->o_sh, HD);
  if (eq > lim || ek > lim || ev > lim || eo > lim) return unsupported("rel_attention_bwd: tensor spans 4 GiB or more");
  const int64_t all = g->q_ss | g->q_sb | g->q_sh | g->k_ss | g->k_sb | g->k_sh | g->v_ss | g->v_sb | g->v_sh |
                      g->o_ss | g->o_sb | g->o_sh;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(g->q) | reinterpret_cast<uintptr_t>(g->k) |
                         reinterpret_cast<uintptr_t>(g->v) | reinterpret_cast<uintptr_t>(g->out) |
                         reinterpret_cast<uintptr_t>(g->rel_embeddings) | reinterpret_cast<uintptr_t>(ga->d_out) |
                         reinterpret_cast<uintptr_t>(ga->dq) | reinterpret_cast<uintptr_t>(ga->dk) |
                         reinterpret_cast<uintptr_t>(ga->dv) | reinterpret_cast<uintptr_t>(ga->workspace);
  if ((all & 3) || (ptrs & 15))
    return invalid("rel_attention_bwd: strides must be multiples of 4 floats and pointers 16-byte aligned");
  int rho_lo = 0, rho_n = 0;
  if (has_e) rho_range(g, &rho_lo, &rho_n);
  const BwdLayout L = bwd_layout(g->B, g->H, g->Sq, rho_n, HD);
  if (ga->workspace_floats < L.total) { set_last_error("rel_attention_bwd: workspace too small"); return ISI_E_WORKSPACE; }
  if (has_e && (int64_t)g->H * g->B * g->Sq * L.Rp > lim) return unsupported("rel_attention_bwd: G spans 4 GiB or more");

  AttnBwdKArgs a;
  memset(&a, 0, sizeof a);
  a.q = g->q; a.k = g->k; a.v = g->v; a.e = g->rel_embeddings; a.mask = g->dense_mask;
  a.dout = ga->d_out; a.out = g->out; a.lse = g->lse;
  a.dsum = ga->workspace + L.dsum;
  a.dq = ga->dq; a.dk = ga->dk; a.dv = ga->dv;
  a.g = has_e ? ga->workspace + L.g : nullptr;
  a.q_bytes = (unsigned)(eq * 4); a.k_bytes = (unsigned)(ek * 4); a.v_bytes = (unsigned)(ev * 4);
  a.o_bytes = (unsigned)(eo * 4);
  a.R = g->rel_rows; a.Rp = L.Rp; a.rho_lo = rho_lo;
  a.e_bytes = (unsigned)((size_t)g->H * g->rel_rows * HD * 4);
  a.Sq = g->Sq; a.Sk = g->Sk; a.H = g->H; a.B = g->B;
  a.q_ss = (int)g->q_ss; a.q_sb = (int)g->q_sb; a.q_sh = (int)g->q_sh;
  a.k_ss = (int)g->k_ss; a.k_sb = (int)g->k_sb; a.k_sh = (int)g->k_sh;
  a.v_ss = (int)g->v_ss; a.v_sb = (int)g->v_sb; a.v_sh = (int)g->v_sh;
  a.o_ss = (int)g->o_ss; a.o_sb = (int)g->o_sb; a.o_sh = (int)g->o_sh;
  a.Cq = g->Cq; a.Ck = g->Ck; a.Ek = g->Ek;
  a.mask_mode = g->mask_mode; a.scale = g->scale;
  if (g->logits) {      // the forward's logits (isi_attn_args.logits): read by the split kernels
    if (g->precision < 1) return unsupported("rel_attention_bwd: kept logits go with the 16-bit modes (precision >= 1)");
    if (g->logits_ld < ((g->Sk + 31) & ~31) || (g->logits_ld & 3) || (reinterpret_cast<uintptr_t>(g->logits) & 15) ||
        g->logits_ld > ((int64_t)1 << 30))
      return invalid("rel_attention_bwd: logits_ld must be a multiple of 4, at least Sk rounded up to 32; logits 16-byte aligned");
    a.logits = g->logits; a.ldl = (int)g->logits_ld;
  }

  const int nstat = g->B * g->H * g->Sq;
  hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)(((int64_t)nstat * (HD / 4) + 255) / 256)), dim3(256), 0, stream, a, HD);
  int rc = check_launch("attn_dsum");
  if (rc) return rc;
  const bool split = g->precision >= 1;
  // G's band per query (columns rho - rho_lo; Cq = Ck = 1): [i + bot, i + top] unmasked, [top, i + top] causal,
  // [i + bot, top] anti-causal -- the two products over G skip what lies outside it
  const int KpT = (int)round_up((size_t)L.Rp, kBK);
  const int Mg = g->Sq * g->B;
  const bool unique = has_e && g->Cq == 1 && g->Ck == 1;
  const int top = g->Ek - 1 - rho_lo, bot = g->Ek - g->Sk - rho_lo;
  const int lo_slope = g->mask_mode == 1 ? 0 : 1, lo_base = g->mask_mode == 1 ? top : bot;
  const int hi_slope = g->mask_mode == 2 ? 0 : 1, hi_base = top;
  const bool gemm_path = has_e && split && KpT == L.Rp && g->q_ss == (int64_t)g->B * g->q_sb && gemm_split_applicable(Mg, HD, L.Rp, 1) &&
                         !knobs().no_gemm_kernel;
  // ... and then only the margins of the band have to be zeroed (attn_zero_margins_kernel)
  const bool band_only = unique && gemm_path && !knobs().attn_full_zero;
  a.g_band_only = band_only ? 1 : 0;
  if (has_e) {
    if (band_only) {
      const long rows = (long)g->H * g->Sq * g->B;
      hipLaunchKernelGGL(attn_zero_margins_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a.g, rows, g->Sq, g->B,
                         L.Rp, lo_slope, lo_base, hi_slope, hi_base);
      if ((rc = check_launch("attn_zero_margins"))) return rc;
    } else {
      // (a kernel, not hipMemsetAsync: a memset node inside a recorded training step -- a replayed HIP graph -- was followed by
      // reductions that read unwritten partial sums in ~3 % of the replays on ROCm 7.2 with graph packet capture)
      const size_t n4 = (size_t)g->H * g->B * g->Sq * L.Rp / 4;       // Rp is a multiple of 4
      hipLaunchKernelGGL(attn_zero_f4_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, stream,
                         reinterpret_cast<float4 *>(a.g), n4);
      if ((rc = check_launch("attn_zero(G)"))) return rc;
    }
  }
  const bool one = g->precision == 2;     // single-term bf16 products, like the forward of that mode
  // rows / keys beyond the last full block that go through the one-row kernels (the G row of such a query is written
  // with one key per column, which needs Ck = 1)
  const int tail_q = (split && (!has_e || g->Ck == 1)) ? attention_tail_rows(g->Sq, g->mask_mode, g->dense_mask != nullptr) : 0;
  const int tail_k = split ? attention_tail_rows(g->Sk, g->mask_mode, g->dense_mask != nullptr) : 0;
  a.nqb = tail_q ? g->Sq / QB : (g->Sq + QB - 1) / QB;
  a.nkb = tail_k ? g->Sk / QB : (g->Sk + QB - 1) / QB;
  // kept logits and a band-only G: the key-stationary kernel (and the one-key kernel) store dS into G, the query-stationary
  // kernel runs last and reads it back (AttnBwdKArgs.g_from_kv; ISI_ATTN_G_FROM_KV=0: every kernel forms its own dS)
  a.g_from_kv = (split && band_only && a.logits && knobs().attn_g_from_kv) ? 1 : 0;
  auto run_split = [&](int which) {
    switch (HD) {
      case 16: return one ? launch_bwd_split<16, true>(a, stream, which) : launch_bwd_split<16, false>(a, stream, which);
      case 32: return one ? launch_bwd_split<32, true>(a, stream, which) : launch_bwd_split<32, false>(a, stream, which);
      default: return one ? launch_bwd_split<64, true>(a, stream, which) : launch_bwd_split<64, false>(a, stream, which);
    }
  };
  auto run_tails = [&]() {
    return HD == 16 ? launch_bwd_tails<16>(a, tail_q, tail_k, stream) : HD == 32 ? launch_bwd_tails<32>(a, tail_q, tail_k, stream)
                                                                               : launch_bwd_tails<64>(a, tail_q, tail_k, stream);
  };
  if (!split) {
    rc = HD == 16 ? launch_bwd<16>(a, stream) : HD == 32 ? launch_bwd<32>(a, stream) : launch_bwd<64>(a, stream);
    if (!rc && (tail_q || tail_k)) rc = run_tails();
  } else if (a.g_from_kv) {
    rc = run_split(1);
    if (!rc && (tail_q || tail_k)) rc = run_tails();      // (the one-key kernel stores its column of G)
    if (!rc) rc = run_split(2);
  } else {
    rc = run_split(3);
    if (!rc && (tail_q || tail_k)) rc = run_tails();
  }
  if (rc || !has_e) return rc;

  // ---- dQ += G E  and  dE = G^T Q, head by head, on the GEMM kernels
  const int R = g->rel_rows;
  float *wT = ga->workspace + L.wT;
  {
    const int64_t total = (int64_t)g->H * HD * KpT;
    hipLaunchKernelGGL(pack_rel_T_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       g->rel_embeddings, wT, g->H, R, HD, KpT, rho_lo, rho_n);
    if ((rc = check_launch("pack_rel_T"))) return rc;
  }
  const int gemm_flags = g->precision >= 1 ? ISI_CONV_BF16X3 : 0;   // same product mode as the attention kernels
  {
    // all heads in one launch each (grid z = head): G_h [B*Sq, Rp] x E_h^T, accumulated into dq's head slice ...
    isi_src sg;
    memset(&sg, 0, sizeof sg);
    sg.ptr = a.g; sg.C = L.Rp; sg.sn = (int64_t)g->B * L.Rp; sg.sc = 1; sg.sh = L.Rp; sg.sw = L.Rp;   // "image" = [Sq][B][1]
    isi_src res;
    memset(&res, 0, sizeof res);
    res.ptr = ga->dq; res.C = HD; res.sn = g->q_ss; res.sc = 1; res.sh = g->q_sb; res.sw = g->q_sb;
    isi_dst dst;
    memset(&dst, 0, sizeof dst);
    dst.ptr = ga->dq; dst.sn = g->q_ss; dst.sc = 1; dst.sh = g->q_sb; dst.sw = g->q_sb;
    const int64_t zs_g = (int64_t)g->B * g->Sq * L.Rp;
    // Rows of G are (query, batch): with one channel per event on both sides the non-zero columns of query i are the
    // band [i + Ek - Sk, i + Ek - 1] - rho_lo (unmasked), [Ek - 1, i + Ek - 1] - rho_lo (causal) ... -- a 128-row
    // tile holds 128 / B queries, its band is about half of the table: the GEMM kernel skips the other K chunks
    GemmExtra gx;
    memset(&gx, 0, sizeof gx);
    gx.nz = g->H; gx.zs_a = zs_g; gx.zs_w = (int64_t)HD * KpT; gx.zs_res = g->q_sh; gx.zs_out = g->q_sh;
    if (g->Cq == 1 && g->Ck == 1) {
      gx.win_rpu = g->B;
      gx.lo_slope = lo_slope; gx.lo_base = lo_base; gx.hi_slope = hi_slope; gx.hi_base = hi_base;
    }
    if (gemm_path)
      rc = gemm_split_f32(a.g, L.Rp, wT, nullptr, ga->dq, g->q_sb, ga->dq, g->q_sb, Mg, HD, L.Rp, 0, 1, stream, nullptr, &gx);
    else
    rc = conv2d_batched_f32(&sg, nullptr, wT, nullptr, &res, &dst, g->Sq, g->B, 1, HD, 1, 1, 1, 0, gemm_flags, g->H,
                            zs_g, (int64_t)HD * KpT, g->q_sh, g->q_sh, stream);
    if (rc) return rc;
    // ... and dE_h = G_h^T Q_h, the pixel-reduction GEMM; column c of G is non-zero on the queries [c - top, c - bot]
    // (unmasked), from c - top on (causal), up to c - bot (anti-causal): the kernel reads only those rows
    const WgradBand wb{g->B, g->mask_mode == 2 ? 0 : 1, g->mask_mode == 2 ? 0 : -top,
                       g->mask_mode == 1 ? 0 : 1, g->mask_mode == 1 ? g->Sq - 1 : -bot, band_only ? 1 : 0};
    isi_src sq;
    memset(&sq, 0, sizeof sq);
    sq.ptr = g->q; sq.C = HD; sq.sn = g->q_ss; sq.sc = 1; sq.sh = g->q_sb; sq.sw = g->q_sb;
    rc = conv_wgrad_batched_f32(&sq, nullptr, a.g, ga->workspace + L.dw, nullptr, ga->workspace + L.wg, L.wg_floats,
                                g->Sq, g->B, 1, L.Rp, 1, 1, 1, 0, gemm_flags, g->H, g->q_sh, zs_g,
                                (int64_t)L.Rp * L.Kp, stream, 0, gx.win_rpu ? &wb : nullptr);
    if (rc) return rc;
  }
  {
    const int64_t total = (int64_t)g->H * R * HD;
    hipLaunchKernelGGL(unpack_drel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       ga->workspace + L.dw, ga->d_rel, g->H, R, HD, L.Rp, L.Kp, rho_lo, rho_n);
    rc = check_launch("unpack_drel");
  }
  return rc;
}

int rel_attention_bwd_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  if (n < 0)      // (n < 0: the query-stationary kernel's stamps, -n of them)
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_q_stamps), sizeof(long long) * (size_t)(-n < 512 ? -n : 512)) == hipSuccess ? 0 : -2;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_kv_stamps), sizeof(long long) * (size_t)(n < 512 ? n : 512)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

}  // namespace isi
