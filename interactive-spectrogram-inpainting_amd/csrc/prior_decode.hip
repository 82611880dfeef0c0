// Native key/value-cached sampling loop of the prior's decoder (gfx950, fp32).
//
// Launches per layer at batch 1 (a DEPENDENT launch costs 2.8-3.4 us on this part whatever its work --
// tools/probes/launch_chain_probe.hip -- and the stages of a decoder layer are a chain, so the count is the budget):
//   LN + q|k|v row-GEMV (k, v straight into the cache slot) -> self-attention over key splits -> merge of the splits
//   + out-projection + residual -> LN1 + cross-attention query -> cross-attention over key splits -> merge +
//   out-projection + residual -> LN2 + linear1 + ReLU -> linear2 + residual.
// Every stage needs ALL outputs of the one before (a GEMV row feeds every output of the next), so a stage boundary is
// a device-wide dependency; within one kernel such a dependency (partial rows + a ticket, the last workgroup finishes)
// was measured dearer than the launch it saves: the release/acquire fences cost 4.5 us and the finishing round trip
// 5.8 us on this multi-die part (attention + out-projection + residual in one kernel: 22.7 us against 16 us for the
// three launches; linear1 + linear2: 16.3 against 9.6 us).  What is folded is what needs no dependency of its own: the
// LayerNorms and the merge of the attention key splits run in the prologue of the consuming GEMV.
//
// One call runs, for every sequence position in [p_begin, p_end), the whole
// decoder stack on ONE new row (8 dependent launches per layer, below), the logits head, the
// categorical draw and the write of the sampled token's embedding into the next
// input row; the sampled index stays on the device.  Single-stream decoding is
// bound by the ~85 dependent launches per position, not by their work (a 512x1536
// GEMV reads 3 MB): every position-dependent address and size is therefore read on
// the device from a position counter, the launch sequence of one position is
// captured ONCE into a hipGraph (two variants: with / without the sampling tail)
// and replayed per position with a single graph launch.  The call returns after the
// last replay has finished (the graph executables are destroyed with the call).  This replaces the reference's per-token full decoder pass
// (sample.py:268-305 -> priors/transformer.py:763-774): decoder self-attention is
// causal and earlier inputs never change, so row p computed from cached keys /
// values equals row p of a full pass (tests/test_prior_gpu.py checks it).
//
// LayerNorms are folded into their consumers: each GEMV normalises its input
// rows (and, when the residual is a normalised tensor, its residual rows) on the
// fly from the stored pre-norm rows.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"

namespace isi {

namespace {
constexpr int NPB = 8;   // output features per workgroup of the row-GEMV (2 per wave, loads of both in flight)

struct RowLinArgs {
  const float *x; int x_stride;            // [M, K] input rows (pre-norm when ln_g)
  const float *ln_g, *ln_b;                // LayerNorm applied to x rows (nullable)
  const float *W, *bias;                   // [N, K] torch layout
  const float *res; int res_stride;        // [M, N] residual rows (nullable)
  const float *res_g, *res_b;              // LayerNorm applied to the residual rows (nullable)
  float *out; int out_stride;              // columns [0, split)
  float *out2; int out2_stride;            // columns [split, N) (nullable: split == N)
  int split, M, N, K, relu;
  float eps;
  // replayable launches (hipGraph): with pos != nullptr, x / res / out2 advance by these element
  // counts per sequence position read from device memory
  const int *pos;
  long x_pos, res_pos, out2_pos;
  // batched decoding on matrix tiles (row_mfma32_kernel): a pre-norm row is normalised by two launches of a position -- as the
  // INPUT of a projection and, one launch later, as the RESIDUAL of the out-projection / second feed-forward layer.  The
  // first writes the rows' statistics here ([M][2]: mean, 1 / std), the second reads them instead of loading the 32 x N
  // residual rows again in every workgroup and reducing them (6 % of a decoding step at B = 32, measured by ablation)
  float *stat_out;
  const float *res_stat;
};

__device__ __forceinline__ float wave_sum(float v) { return wave64_sum(v); }   // DPP path (isi_common.h)
// Every field of a launch's argument block is "used" by an empty asm statement at the head of the kernel: the compiler
// otherwise fetches the fields where they are first needed -- three or four DEPENDENT scalar-memory round trips (the later
// ones to a 64-byte line of the freshly written block that no earlier load touched) in front of the first weight request,
// in every one of the ~66 launches of a token.
__device__ __forceinline__ void touch_args(const RowLinArgs &a) {
  asm volatile("" ::"s"(a.x), "s"(a.x_stride), "s"(a.ln_g), "s"(a.ln_b), "s"(a.W), "s"(a.bias), "s"(a.res), "s"(a.res_stride),
               "s"(a.res_g), "s"(a.res_b), "s"(a.out), "s"(a.out_stride));
  asm volatile("" ::"s"(a.out2), "s"(a.out2_stride), "s"(a.split), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.relu), "s"(a.eps),
               "s"(a.pos), "s"(a.x_pos), "s"(a.res_pos), "s"(a.out2_pos));
  asm volatile("" ::"s"(a.stat_out), "s"(a.res_stat));
}

template <int MR>
__global__ __launch_bounds__(256) void row_linear_ln_kernel(RowLinArgs a) {
  touch_args(a);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (a.pos) {
    const long p = *a.pos;
    a.x += p * a.x_pos;
    if (a.res) a.res += p * a.res_pos;
    if (a.out2) a.out2 += p * a.out2_pos;
  }
  float *xs = sm;                    // [MR][K]
  float *stat = sm + MR * a.K;       // [MR][2] residual mean / rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.K >> 2;
  // weight rows of this wave's two output features: issue the loads first (they depend on nothing)
  constexpr int WPF = 8;  // float4 per lane and row held in registers: K <= 2048
  const int n0 = blockIdx.x * NPB + wave * 2;
  const bool two = n0 + 1 < a.N;
  const float4 *w0 = reinterpret_cast<const float4 *>(a.W + (size_t)(n0 < a.N ? n0 : 0) * a.K);
  const float4 *w1 = reinterpret_cast<const float4 *>(a.W + (size_t)(two ? n0 + 1 : (n0 < a.N ? n0 : 0)) * a.K);
  float4 wa[WPF], wb[WPF];
#pragma unroll
  for (int i = 0; i < WPF; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) { wa[i] = w0[qd]; wb[i] = w1[qd]; }
  }
  // ---- phase 0: stage (normalised) input rows; residual statistics
  for (int m = wave; m < a.M; m += 4) {
    const float4 *xr = reinterpret_cast<const float4 *>(a.x + (size_t)m * a.x_stride);
    float mean = 0.f, rstd = 1.f;
    if (a.ln_g) {
      float s = 0.f;
      for (int qd = lane; qd < nq; qd += 64) { const float4 v = xr[qd]; s += (v.x + v.y) + (v.z + v.w); }
      mean = wave_sum(s) / (float)a.K;
      float var = 0.f;
      for (int qd = lane; qd < nq; qd += 64) {
        const float4 v = xr[qd];
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      rstd = 1.0f / sqrtf(wave_sum(var) / (float)a.K + a.eps);
    }
    for (int qd = lane; qd < nq; qd += 64) {
      float4 v = xr[qd];
      if (a.ln_g) {
        const float4 g = reinterpret_cast<const float4 *>(a.ln_g)[qd], b = reinterpret_cast<const float4 *>(a.ln_b)[qd];
        v.x = (v.x - mean) * rstd * g.x + b.x; v.y = (v.y - mean) * rstd * g.y + b.y;
        v.z = (v.z - mean) * rstd * g.z + b.z; v.w = (v.w - mean) * rstd * g.w + b.w;
      }
      reinterpret_cast<float4 *>(xs + (size_t)m * a.K)[qd] = v;
    }
    if (a.res && a.res_g) {
      const float *rr = a.res + (size_t)m * a.res_stride;
      float s = 0.f;
      for (int i = lane; i < a.N; i += 64) s += rr[i];
      const float rm = wave_sum(s) / (float)a.N;
      float var = 0.f;
      for (int i = lane; i < a.N; i += 64) { const float d = rr[i] - rm; var += d * d; }
      const float vs = wave_sum(var);
      if (lane == 0) { stat[2 * m] = rm; stat[2 * m + 1] = 1.0f / sqrtf(vs / (float)a.N + a.eps); }
    }
  }
  __syncthreads();
  // ---- phase 1: NPB output features per workgroup, 2 per wave (their weight rows were requested at
  // kernel entry, before the input rows: the two global round trips overlap)
  if (n0 >= a.N) return;
  float acc0[MR], acc1[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
#pragma unroll
  for (int i = 0; i < WPF; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (m < a.M) {
          const float4 xv = reinterpret_cast<const float4 *>(xs + (size_t)m * a.K)[qd];
          acc0[m] += (wa[i].x * xv.x + wa[i].y * xv.y) + (wa[i].z * xv.z + wa[i].w * xv.w);
          acc1[m] += (wb[i].x * xv.x + wb[i].y * xv.y) + (wb[i].z * xv.z + wb[i].w * xv.w);
        }
      }
    }
  }
  for (int qd = lane + 64 * WPF; qd < nq; qd += 64) {   // K > 2048: the tail streams as before
    const float4 wa2 = w0[qd], wb2 = w1[qd];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < a.M) {
        const float4 xv = reinterpret_cast<const float4 *>(xs + (size_t)m * a.K)[qd];
        acc0[m] += (wa2.x * xv.x + wa2.y * xv.y) + (wa2.z * xv.z + wa2.w * xv.w);
        acc1[m] += (wb2.x * xv.x + wb2.y * xv.y) + (wb2.z * xv.z + wb2.w * xv.w);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
  if (lane < 2 && (lane == 0 || two)) {
    const int n = n0 + lane;
    const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < a.M) {
        float v = (lane == 0 ? acc0[m] : acc1[m]) + b;
        if (a.res) {
          float r = a.res[(size_t)m * a.res_stride + n];
          if (a.res_g) r = (r - stat[2 * m]) * stat[2 * m + 1] * a.res_g[n] + a.res_b[n];
          v += r;
        }
        if (a.relu) v = fmaxf(v, 0.f);
        if (n < a.split) a.out[(size_t)m * a.out_stride + n] = v;
        else a.out2[(size_t)m * a.out2_stride + (n - a.split)] = v;
      }
    }
  }
}

// ---- batch 1: one row.  No LDS, no barrier: every load of the kernel (weight rows, input row, LayerNorm parameters,
// bias, residual) is issued before anything is computed, so the kernel is one memory round trip long.  The arithmetic
// is that of row_linear_ln_kernel operation for operation (a row of a batch and the same row alone give the same bits).
struct Gemv1Args {
  RowLinArgs r;
  const float *part;   // nullable: attention key-split partials [K / HD heads][NS][HD + 4] merged into the input row
  int NS, HD;
  int nt;              // non-temporal weight loads (ISI_DECODE_NT, default on)
};

template <int KQ>   // float4 per lane of a K-long row: K <= 256 KQ
__global__ __launch_bounds__(256) void row_gemv1_kernel(Gemv1Args g) {
  touch_args(g.r);
  asm volatile("" ::"s"(g.part), "s"(g.NS), "s"(g.HD), "s"(g.nt));
  RowLinArgs &a = g.r;
  // replayable launches: the position comes from device memory.  The load is issued here and only waited for where an
  // offset actually depends on it (most launches of a position have none, or only the cache slot of their store)
  long ppos = 0;
  if (a.pos) ppos = *a.pos;
  if (a.x_pos) a.x += ppos * a.x_pos;
  if (a.res && a.res_pos) a.res += ppos * a.res_pos;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.K >> 2;
  // two output features per wave; ONE for long rows (K > 512: a wave then streams 4-8 KB instead of 8-16 KB and the
  // launch has twice the workgroups -- the K = 2048 launch of a layer went from 6.7 to ~5.5 us)
  constexpr int RPW = KQ >= 4 ? 1 : 2;
  const int n0 = blockIdx.x * (4 * RPW) + wave * RPW;
  if (n0 >= a.N) return;
  const bool two = RPW == 2 && n0 + 1 < a.N;
  const float4 *w0 = reinterpret_cast<const float4 *>(a.W + (size_t)n0 * a.K);
  const float4 *w1 = reinterpret_cast<const float4 *>(a.W + (size_t)(two ? n0 + 1 : n0) * a.K);
  float4 wa[KQ], wb[KQ], xv[KQ], gq[KQ], bq[KQ];
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      // weight rows are streamed once per token by exactly one wave: non-temporal loads (MI355X guide, nt-weights:
      // issued -> landed ~18 % sooner for such streams)
      typedef float f32x4nt __attribute__((ext_vector_type(4)));
      if (g.nt) {
        wa[i] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4nt *>(w0) + qd));
        if constexpr (RPW == 2) wb[i] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4nt *>(w1) + qd));
      } else {
        wa[i] = w0[qd];
        if constexpr (RPW == 2) wb[i] = w1[qd];
      }
      if constexpr (RPW != 2) wb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  constexpr int PS = KQ <= 2 ? 8 : 1;       // key splits held in registers (merged input: K <= 512)
  float4 pv[KQ][PS];
  float pm[KQ][PS], pl[KQ][PS];
  if (g.part) {
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int qd = lane + 64 * i;
      if (qd < nq) {
        const int hh = (4 * qd) / g.HD, c = (4 * qd) % g.HD;
        const float *base = g.part + (size_t)hh * g.NS * (g.HD + 4);
        // (no condition around a split's loads: behind `if (s2 < NS)` the compiler waited for each split's pair before it
        // requested the next -- up to eight round trips in a row at the head of both out-projection launches of a layer,
        // hipcc -S; splits beyond NS re-read the last one and are left out of the merge below)
#pragma unroll
        for (int s2 = 0; s2 < PS; ++s2) {
          const int s2c = min(s2, g.NS - 1);
          pv[i][s2] = *reinterpret_cast<const float4 *>(base + (size_t)s2c * (g.HD + 4) + c);
          const float2 ml = *reinterpret_cast<const float2 *>(base + (size_t)s2c * (g.HD + 4) + g.HD);
          pm[i][s2] = ml.x;
          pl[i][s2] = ml.y;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int qd = lane + 64 * i;
      xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qd < nq) {
        xv[i] = reinterpret_cast<const float4 *>(a.x)[qd];
        if (a.ln_g) { gq[i] = reinterpret_cast<const float4 *>(a.ln_g)[qd]; bq[i] = reinterpret_cast<const float4 *>(a.ln_b)[qd]; }
      }
    }
  }
  const int n = n0 + (lane & 1);
  float bias_v = 0.f, res_v = 0.f, rg = 1.f, rb = 0.f;
  if (lane < 2 && (lane == 0 || two)) {
    if (a.bias) bias_v = a.bias[n];
    if (a.res) res_v = a.res[n];
    if (a.res && a.res_g) { rg = a.res_g[n]; rb = a.res_b[n]; }
  }
  constexpr int RS = 8;                      // residual row for its LayerNorm statistics: N <= 512
  float rrow[RS];
  float2 rstat = make_float2(0.f, 1.f);      // the residual row's statistics from the launch that formed them (RowLinArgs.res_stat)
  if (a.res && a.res_g && a.res_stat) rstat = *reinterpret_cast<const float2 *>(a.res_stat);
  else if (a.res && a.res_g) {
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      const int c = lane + 64 * i;
      rrow[i] = c < a.N ? a.res[c] : 0.f;
    }
  }
  // ---- everything is in flight; compute
  if (g.part) {
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        float M = -1e30f;
#pragma unroll
        for (int s2 = 0; s2 < PS; ++s2) if (s2 < g.NS) M = fmaxf(M, pm[i][s2]);
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float den = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < PS; ++s2) {
          if (s2 < g.NS) {
            const float w = __expf(pm[i][s2] - M);     // (the combine kernel's arithmetic: transformer_ops.hip)
            num.x += w * pv[i][s2].x; num.y += w * pv[i][s2].y; num.z += w * pv[i][s2].z; num.w += w * pv[i][s2].w;
            den += w * pl[i][s2];
          }
        }
        const float rden = 1.0f / den;
        xv[i] = make_float4(num.x * rden, num.y * rden, num.z * rden, num.w * rden);
      }
    }
  } else if (a.ln_g) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KQ; ++i) if (lane + 64 * i < nq) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    const float mean = wave_sum(s) / (float)a.K;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        const float d0 = xv[i].x - mean, d1 = xv[i].y - mean, d2 = xv[i].z - mean, d3 = xv[i].w - mean;
        var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)a.K + a.eps);
    // (for the launch that normalises this row again as its residual: one dependent chain of two reductions, two
    // divisions and a square root less on the one wave its SIMD holds)
    if (a.stat_out && blockIdx.x == 0 && tid == 0) *reinterpret_cast<float2 *>(a.stat_out) = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        xv[i].x = (xv[i].x - mean) * rstd * gq[i].x + bq[i].x; xv[i].y = (xv[i].y - mean) * rstd * gq[i].y + bq[i].y;
        xv[i].z = (xv[i].z - mean) * rstd * gq[i].z + bq[i].z; xv[i].w = (xv[i].w - mean) * rstd * gq[i].w + bq[i].w;
      }
    }
  }
  float rmean = rstat.x, rrstd = rstat.y;
  if (a.res && a.res_g && !a.res_stat) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) s += rrow[i];
    rmean = wave_sum(s) / (float)a.N;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) { const float dv = rrow[i] - rmean; var += dv * dv; }
    rrstd = 1.0f / sqrtf(wave_sum(var) / (float)a.N + a.eps);
  }
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    if (lane + 64 * i < nq) {
      acc0 += (wa[i].x * xv[i].x + wa[i].y * xv[i].y) + (wa[i].z * xv[i].z + wa[i].w * xv[i].w);
      acc1 += (wb[i].x * xv[i].x + wb[i].y * xv[i].y) + (wb[i].z * xv[i].z + wb[i].w * xv[i].w);
    }
  }
  acc0 = wave_sum(acc0);
  acc1 = wave_sum(acc1);
  if (lane < 2 && (lane == 0 || two)) {
    float v = (lane == 0 ? acc0 : acc1) + bias_v;
    if (a.res) {
      float r = res_v;
      if (a.res_g) r = (r - rmean) * rrstd * rg + rb;
      v += r;
    }
    if (a.relu) v = fmaxf(v, 0.f);
    if (n < a.split) a.out[n] = v;
    else a.out2[ppos * a.out2_pos + (n - a.split)] = v;
  }
}

// ---- several rows (batched decoding): the same kernel shape with the rows' inputs in registers -- weights are loaded
// once and used for every row, the rows pass through in groups of MR (round 5: ALL rows of a stage in one launch; groups of
// 8 rows used to be launches of their own, 4 per stage at batch 32).  Per row the operations of row_gemv1_kernel /
// row_linear_ln_kernel, in their order: a row's result does not depend on the rows it shares a launch with.
template <int KQ, int MR>
__global__ __launch_bounds__(256) void row_gemvm_kernel(RowLinArgs a) {
  touch_args(a);
  long ppos = 0;
  if (a.pos) ppos = *a.pos;
  if (a.x_pos) a.x += ppos * a.x_pos;
  if (a.res && a.res_pos) a.res += ppos * a.res_pos;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.K >> 2;
  constexpr int RPW = KQ >= 4 ? 1 : 2;
  const int n0 = blockIdx.x * (4 * RPW) + wave * RPW;
  if (n0 >= a.N) return;
  const bool two = RPW == 2 && n0 + 1 < a.N;
  const float4 *w0 = reinterpret_cast<const float4 *>(a.W + (size_t)n0 * a.K);
  const float4 *w1 = reinterpret_cast<const float4 *>(a.W + (size_t)(two ? n0 + 1 : n0) * a.K);
  float4 wa[KQ], wb[KQ], gq[KQ], bq[KQ];
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      wa[i] = w0[qd];
      if constexpr (RPW == 2) wb[i] = w1[qd];
      else wb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.ln_g) { gq[i] = reinterpret_cast<const float4 *>(a.ln_g)[qd]; bq[i] = reinterpret_cast<const float4 *>(a.ln_b)[qd]; }
    }
  }
  const int n = n0 + (lane & 1);
  const bool writer = lane < 2 && (lane == 0 || two);
  float bias_v = 0.f, rg = 1.f, rb = 0.f;
  if (writer) {
    if (a.bias) bias_v = a.bias[n];
    if (a.res && a.res_g) { rg = a.res_g[n]; rb = a.res_b[n]; }
  }
  for (int m0 = 0; m0 < a.M; m0 += MR) {
  const float *xg = a.x + (size_t)m0 * a.x_stride;
  const float *resg = a.res ? a.res + (size_t)m0 * a.res_stride : nullptr;
  const int Mg = a.M - m0 < MR ? a.M - m0 : MR;
  float4 xv[MR][KQ];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int qd = lane + 64 * i;
      xv[m][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < Mg && qd < nq) xv[m][i] = reinterpret_cast<const float4 *>(xg + (size_t)m * a.x_stride)[qd];
    }
  float res_v[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) res_v[m] = (writer && resg && m < Mg) ? resg[(size_t)m * a.res_stride + n] : 0.f;
  constexpr int RS = 8;
  float rrow[MR][RS];
  float2 rstat[MR];        // the residual rows' statistics from the launch that formed them (RowLinArgs.res_stat)
  const bool res_handed = resg && a.res_g && a.res_stat;
  if (res_handed) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
      rstat[m] = m < Mg ? *reinterpret_cast<const float2 *>(a.res_stat + 2 * (size_t)(m0 + m)) : make_float2(0.f, 1.f);
  } else if (resg && a.res_g) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int c = lane + 64 * i;
        rrow[m][i] = (m < Mg && c < a.N) ? resg[(size_t)m * a.res_stride + c] : 0.f;
      }
  }
  // the group's rows go through each phase TOGETHER (statistics, normalisation, products, reductions): MR independent chains
  // of wave reductions interleave, where row after row every reduction waited for the one before (per row the operations
  // and their order are unchanged)
  if (a.ln_g) {
    float mean[MR], rstd[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) if (lane + 64 * i < nq) s += (xv[m][i].x + xv[m][i].y) + (xv[m][i].z + xv[m][i].w);
      mean[m] = wave_sum(s) / (float)a.K;
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) {
        if (lane + 64 * i < nq) {
          const float d0 = xv[m][i].x - mean[m], d1 = xv[m][i].y - mean[m], d2 = xv[m][i].z - mean[m], d3 = xv[m][i].w - mean[m];
          var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      rstd[m] = 1.0f / sqrtf(wave_sum(var) / (float)a.K + a.eps);
    }
    if (a.stat_out && blockIdx.x == 0 && tid == 0) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
        if (m < Mg) *reinterpret_cast<float2 *>(a.stat_out + 2 * (size_t)(m0 + m)) = make_float2(mean[m], rstd[m]);
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int i = 0; i < KQ; ++i) {
        if (lane + 64 * i < nq) {
          xv[m][i].x = (xv[m][i].x - mean[m]) * rstd[m] * gq[i].x + bq[i].x; xv[m][i].y = (xv[m][i].y - mean[m]) * rstd[m] * gq[i].y + bq[i].y;
          xv[m][i].z = (xv[m][i].z - mean[m]) * rstd[m] * gq[i].z + bq[i].z; xv[m][i].w = (xv[m][i].w - mean[m]) * rstd[m] * gq[i].w + bq[i].w;
        }
      }
  }
  float rmean[MR], rrstd[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { rmean[m] = res_handed ? rstat[m].x : 0.f; rrstd[m] = res_handed ? rstat[m].y : 1.f; }
  if (resg && a.res_g && !res_handed) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) s += rrow[m][i];
      rmean[m] = wave_sum(s) / (float)a.N;
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) { const float dv = rrow[m][i] - rmean[m]; var += dv * dv; }
      rrstd[m] = 1.0f / sqrtf(wave_sum(var) / (float)a.N + a.eps);
    }
  }
  float acc0[MR], acc1[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        a0 += (wa[i].x * xv[m][i].x + wa[i].y * xv[m][i].y) + (wa[i].z * xv[m][i].z + wa[i].w * xv[m][i].w);
        a1 += (wb[i].x * xv[m][i].x + wb[i].y * xv[m][i].y) + (wb[i].z * xv[m][i].z + wb[i].w * xv[m][i].w);
      }
    }
    acc0[m] = a0; acc1[m] = a1;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (writer && m < Mg) {
      float v = (lane == 0 ? acc0[m] : acc1[m]) + bias_v;
      if (resg) {
        float r = res_v[m];
        if (a.res_g) r = (r - rmean[m]) * rrstd[m] * rg + rb;
        v += r;
      }
      if (a.relu) v = fmaxf(v, 0.f);
      const size_t mr = (size_t)(m0 + m);
      if (n < a.split) a.out[mr * a.out_stride + n] = v;
      else a.out2[ppos * a.out2_pos + mr * a.out2_stride + (n - a.split)] = v;
    }
  }
  }
}

// rows x float4-per-lane held in registers: at most 16 (64 VGPRs)
// ---- many rows (batched decoding with more than `decode_mfma_rows` = 16 sequences): the stage as 32-row GEMM tiles on the
// fp32 matrix pipe.  A workgroup = 32 rows x 32 output features; K passes through LDS in chunks of 512 (LayerNorm applied
// while staging: the row statistics are formed first, the next chunk travels under the current one's matrix work); the four
// waves each take a quarter of a chunk's k range (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) and
// their accumulators meet in LDS.  Lane (i, kk) feeds four consecutive k of its row / feature per 16-byte read --
// k = 8 j + 4 kk -- to four MFMAs: A from LDS, B (the weight row of feature n0 + i) straight from memory.  A row's result
// does not depend on the rows it shares the tile with, but its summation order differs from the one-row kernels': batch-1
// decoding and batches of up to `decode_mfma_rows` rows keep the GEMV kernels above (one operation order: equal up to the
// compiler's contraction, a few ulp).
// Measured (tools/kt_sampling_b32.sh, B = 32): 17.9 us per launch on average against 23.5 for the GEMV kernel looping over
// four groups of 8 rows (and 4 x ~10 for the four launches of round 4); 10.7 us of it is the skeleton (requests, staging,
// three barriers, epilogue), 4.2 the LayerNorm statistics, 3.0 the matrix instructions.  At 8 rows the GEMV kernel wins
// (10 against 15 us); beyond 32 rows the tiles of a stage run side by side (grid y) where the GEMV loop grows linearly.
constexpr int MF_KC = 512;
constexpr int MF_LD = MF_KC + 4;         // padded LDS row: 16-byte reads of 32 consecutive rows are conflict-free
constexpr size_t kRowMfmaLds = (size_t)(32 * MF_LD + 128 + 4 * 32 * 33) * sizeof(float);
typedef float mf_f32x16 __attribute__((ext_vector_type(16)));

// LayerNorm statistics of this wave's 8 rows of the tile (KQS float4 per lane and row, 16 / KQS rows per batch of loads)
template <int KQS>
__device__ __forceinline__ void mfma_row_stats(const RowLinArgs &a, float *__restrict__ stat, int m0, int Mt, int nq, int lane,
                                               int wave) {
  constexpr int RB = 16 / KQS;
  for (int r0 = 8 * wave; r0 < 8 * wave + 8; r0 += RB) {
    float4 xr[RB][KQS];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
#pragma unroll
      for (int i = 0; i < KQS; ++i) {
        const int r = r0 + rr, qd = lane + 64 * i;
        xr[rr][i] = (r < Mt && qd < nq) ? reinterpret_cast<const float4 *>(a.x + (size_t)(m0 + r) * a.x_stride)[qd]
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      float s_ = 0.f;
#pragma unroll
      for (int i = 0; i < KQS; ++i) s_ += (xr[rr][i].x + xr[rr][i].y) + (xr[rr][i].z + xr[rr][i].w);
      const float mean = wave_sum(s_) / (float)a.K;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < KQS; ++i) {
        if (lane + 64 * i < nq) {
          const float d0 = xr[rr][i].x - mean, d1 = xr[rr][i].y - mean, d2 = xr[rr][i].z - mean, d3 = xr[rr][i].w - mean;
          var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)a.K + a.eps);
      if (lane == 0) { stat[4 * (r0 + rr)] = mean; stat[4 * (r0 + rr) + 1] = rstd; }
    }
  }
}

// The statistics of the 32 RAW rows the workgroup has just staged in LDS (one K chunk: K <= MF_KC = 128 float4 per row), EIGHT
// LANES PER ROW: a wave takes its eight rows side by side -- 16 float4 per lane, four partial sums, a three-step DPP
// reduction, ONE division / square-root sequence for the eight rows.  The form above (a wave per row, row after row) is a
// chain of dependent instructions per row -- LDS read, sums, two wave reductions through an SGPR, two IEEE divisions, a
// square root, ~1300 cycles -- eight times over for the one wave a SIMD holds: 5 us of a 14-16 us launch with a LayerNorm
// in front (ablation + per-grid kernel trace, tools/kt_rowtiles.sh), in every workgroup of 24 launches per decoding step.
// (Summation order differs from the batch-1 kernels' by construction; so does the tile product's.)
__device__ __forceinline__ void mfma_row_stats_lds(const float *__restrict__ xs, float *__restrict__ stat, int Mt, int nq, int K,
                                                   float eps, int lane, int wave) {
  const int r = 8 * wave + (lane >> 3), sub = lane & 7;
  float4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int qd = sub + 8 * j;
    v[j] = (r < Mt && qd < nq) ? *reinterpret_cast<const float4 *>(xs + r * MF_LD + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) s4[j & 3] += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = group8_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) / (float)K;
  float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (sub + 8 * j < nq) {
      const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
      q4[j & 3] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.0f / sqrtf(group8_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) / (float)K + eps);
  if (sub == 0) { stat[4 * r] = mean; stat[4 * r + 1] = rstd; }
}

// `ksplit_ws` != nullptr (K beyond one chunk on a grid too small to fill the chip: linear2 of a feed-forward block, 16 tiles at
// N = 512): grid z = the K chunk, a workgroup multiplies ONE chunk and leaves its raw 32 x 32 sums in ksplit_ws[z][M][N];
// row_mfma_finish_kernel adds the chunks in order and applies bias / residual / ReLU (33.8 -> ~13 us for that stage at B = 32).
__global__ __launch_bounds__(256) void row_mfma32_kernel(RowLinArgs a, float *__restrict__ ksplit_ws, int knobs_decode_stats_global) {
  touch_args(a);
  asm volatile("" ::"s"(ksplit_ws));
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *xs = sm;                         // [32][MF_LD] normalised input rows of the chunk
  float *stat = xs + 32 * MF_LD;          // [32][4] input mean / rstd, residual mean / rstd
  float *red = stat + 128;                // [4][32][33] the waves' accumulators
  long ppos = 0;
  if (a.pos) ppos = *a.pos;
  if (a.x_pos) a.x += ppos * a.x_pos;
  if (a.res && a.res_pos) a.res += ppos * a.res_pos;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int Mt = a.M - m0 < 32 ? a.M - m0 : 32;
  const int nq = a.K >> 2;
  const int idx = lane & 31, kk = lane >> 5;            // A: row idx of the tile; B: feature n0 + idx
  const float *wrow = a.W + (size_t)(n0 + idx < a.N ? n0 + idx : 0) * a.K;
  // one chunk's operands: the wave's 16 weight pieces (k = 8 (wave + 4 t) + 4 kk) and the thread's 16 pieces of the 32 input
  // rows (piece f = tid + 256 u: row f / cq, float4 f % cq) -- all requested before anything waits
  float4 bw[16], sv[16];
  auto request_chunk = [&](int kc) {
    const int kcn = a.K - kc < MF_KC ? a.K - kc : MF_KC;
    const int cq = kcn >> 2;
    const int csh = (cq & (cq - 1)) == 0 ? __builtin_ctz(cq) : -1;       // (a chunk of 512: f / cq is a shift)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = 8 * (wave + 4 * t) + 4 * kk;
      bw[t] = k < kcn ? *reinterpret_cast<const float4 *>(wrow + kc + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int f = tid + 256 * u, r = csh >= 0 ? f >> csh : f / cq, c4 = f - r * cq;
      sv[u] = (f < 32 * cq && r < Mt) ? *reinterpret_cast<const float4 *>(a.x + (size_t)(m0 + r) * a.x_stride + kc + 4 * c4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const int kc_begin = ksplit_ws ? (int)blockIdx.z * MF_KC : 0;
  const int kc_end = ksplit_ws ? (kc_begin + MF_KC < a.K ? kc_begin + MF_KC : a.K) : a.K;
  request_chunk(kc_begin);
  // the epilogue's per-feature values and residual entries (output o = tid + 256 q: row o >> 5, feature n0 + (tid & 31))
  const int nj = n0 + (tid & 31);
  const bool nok = nj < a.N;
  const float bias_v = (a.bias && nok) ? a.bias[nj] : 0.f;
  float rg = 1.f, rb = 0.f, resv[4];
  const bool rln = a.res && a.res_g && !ksplit_ws;
  if (rln && nok) { rg = a.res_g[nj]; rb = a.res_b[nj]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = (tid >> 5) + 8 * q;
    resv[q] = (a.res && !ksplit_ws && nok && i < Mt) ? a.res[(size_t)(m0 + i) * a.res_stride + nj] : 0.f;
  }
  // ---- row statistics (wave w: rows 8 w .. 8 w + 7)
  // one K chunk of at most 128 float4 per row (the prior's d_model-wide LayerNorm inputs): statistics from the staged rows
#ifdef ISI_ROWMFMA_ABL_STATS
  const bool stats_from_lds = false;
#else
  const bool stats_from_lds = a.ln_g && a.K <= MF_KC && nq <= 128 && !knobs_decode_stats_global;
#endif
#ifdef ISI_ROWMFMA_ABL_STATS
  if (a.ln_g && tid < 32) { stat[4 * tid] = 0.f; stat[4 * tid + 1] = 1.f; }
  if (false) {
#else
  if (a.ln_g && !stats_from_lds) {
#endif
    const int kq = (nq + 63) >> 6;                       // float4 per lane and row
    if (kq <= 2) mfma_row_stats<2>(a, stat, m0, Mt, nq, lane, wave);
    else if (kq <= 4) mfma_row_stats<4>(a, stat, m0, Mt, nq, lane, wave);
    else mfma_row_stats<8>(a, stat, m0, Mt, nq, lane, wave);
  }
#ifdef ISI_ROWMFMA_ABL_RES
  if (rln && tid < 32) { stat[4 * tid + 2] = 0.f; stat[4 * tid + 3] = 1.f; }
  if (false) {
#else
  if (rln && a.res_stat) {       // written by the launch that normalised these rows as its input
    if (tid < Mt) {
      const float2 st2 = *reinterpret_cast<const float2 *>(a.res_stat + 2 * (size_t)(m0 + tid));
      stat[4 * tid + 2] = st2.x;
      stat[4 * tid + 3] = st2.y;
    }
  } else if (rln) {       // residual rows (N <= 512 where they are normalised: 8 floats per lane and row)
#endif
    float rv[8][8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 8 * wave + rr, c = lane + 64 * i;
        rv[rr][i] = (r < Mt && c < a.N) ? a.res[(size_t)(m0 + r) * a.res_stride + c] : 0.f;
      }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float s_ = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s_ += rv[rr][i];
      const float rmean = wave_sum(s_) / (float)a.N;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) if (lane + 64 * i < a.N) { const float d = rv[rr][i] - rmean; var += d * d; }
      const float rrstd = 1.0f / sqrtf(wave_sum(var) / (float)a.N + a.eps);
      if (lane == 0) { stat[4 * (8 * wave + rr) + 2] = rmean; stat[4 * (8 * wave + rr) + 3] = rrstd; }
    }
  }
  mf_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kc = kc_begin; kc < kc_end; kc += MF_KC) {
    const int kcn = a.K - kc < MF_KC ? a.K - kc : MF_KC;
    const int cq = kcn >> 2;
    const int csh = (cq & (cq - 1)) == 0 ? __builtin_ctz(cq) : -1;
    float4 g_pre = make_float4(1.f, 1.f, 1.f, 1.f), b_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool gb_pre = stats_from_lds && cq == 128;
    if (stats_from_lds) {   // (kc == kc_begin == 0: one chunk)  raw rows -> LDS, statistics from there, then the rows again, normalised
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = tid + 256 * u, r = csh >= 0 ? f >> csh : f / cq, c4 = f - r * cq;
        if (f < 32 * cq) *reinterpret_cast<float4 *>(xs + r * MF_LD + 4 * c4) = sv[u];
      }
      __syncthreads();
      // (LayerNorm weight / bias of this thread's columns -- float4 number tid % 128 of a 512-wide row, whatever the piece --
      // travel under the statistics; requested where they are applied their L2 round trip was exposed)
      if (cq == 128) { g_pre = reinterpret_cast<const float4 *>(a.ln_g)[tid & 127]; b_pre = reinterpret_cast<const float4 *>(a.ln_b)[tid & 127]; }
#ifdef ISI_ROWMFMA_ABL_STATS2
      if (tid < 32) { stat[4 * tid] = 0.f; stat[4 * tid + 1] = 1.f; }
#else
      mfma_row_stats_lds(xs, stat, Mt, nq, a.K, a.eps, lane, wave);
#endif
    }
    __syncthreads();       // the statistics are written / the previous chunk's rows have been read
    if (a.stat_out && a.ln_g && kc == kc_begin && blockIdx.x == 0 && blockIdx.z == 0 && tid < Mt)
      *reinterpret_cast<float2 *>(a.stat_out + 2 * (size_t)(m0 + tid)) = make_float2(stat[4 * tid], stat[4 * tid + 1]);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int f = tid + 256 * u, r = csh >= 0 ? f >> csh : f / cq, c4 = f - r * cq;
      if (f < 32 * cq) {
        float4 v = sv[u];
        if (a.ln_g && r < Mt) {
#ifdef ISI_ROWMFMA_ABL_LNLOAD
          const float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
#else
          // (requested with the tile at the head of the kernel the launch was 1.5 us LONGER, tools/kt_rowtiles.sh)
          float4 g = g_pre, b = b_pre;
          if (!gb_pre) { g = reinterpret_cast<const float4 *>(a.ln_g + kc)[c4]; b = reinterpret_cast<const float4 *>(a.ln_b + kc)[c4]; }
#endif
          const float mean = stat[4 * r], rstd = stat[4 * r + 1];
          v.x = (v.x - mean) * rstd * g.x + b.x; v.y = (v.y - mean) * rstd * g.y + b.y;
          v.z = (v.z - mean) * rstd * g.z + b.z; v.w = (v.w - mean) * rstd * g.w + b.w;
        }
        *reinterpret_cast<float4 *>(xs + r * MF_LD + 4 * c4) = v;
      }
    }
    float4 bc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) bc[t] = bw[t];
    __syncthreads();
    if (kc + MF_KC < kc_end) request_chunk(kc + MF_KC);   // the next chunk travels under this chunk's matrix work
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (8 * (wave + 4 * t) < kcn) {        // (uniform per wave)
        const float4 av = *reinterpret_cast<const float4 *>(xs + idx * MF_LD + 8 * (wave + 4 * t) + 4 * kk);
#ifdef ISI_ROWMFMA_ABL_MFMA   // measurement only (wrong results): the tile WITHOUT its matrix instructions.  Round 6, batched decoding at
        // B = 32 / 128: 27.0 -> 28.6 / 58.2 -> 60.2 k codes/s -- the fp32 matrix pipe (1/16 of the 16-bit rate) is worth 6 % / 3.5 %
        // of a decoding step; a launch is a chain of dependent memory round trips (position, rows + weights, statistics, store),
        // so the three-term 16-bit form VERDICT r05 asked for was not built
        acc[t & 15] += av.x * bc[t].x + av.y * bc[t].y + av.z * bc[t].z + av.w * bc[t].w;
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bc[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bc[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bc[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bc[t].w, acc, 0, 0, 0);
#endif
      }
    }
  }
  // ---- the four k quarters meet in LDS: C[i][j], j = lane & 31, i = 4 kk + 8 (r >> 2) + (r & 3)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + 4 * kk + 8 * (r >> 2) + (r & 3)) * 33 + idx] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = (tid >> 5) + 8 * q, j = tid & 31;
    if (i < Mt && nok) {
      float v = (red[i * 33 + j] + red[(32 + i) * 33 + j]) + (red[(64 + i) * 33 + j] + red[(96 + i) * 33 + j]);
      if (ksplit_ws) {       // raw sums of this K chunk; row_mfma_finish_kernel does the rest
        ksplit_ws[((size_t)blockIdx.z * a.M + (m0 + i)) * a.N + nj] = v;
        continue;
      }
      v += bias_v;
      if (a.res) {
        float r = resv[q];
        if (rln) r = (r - stat[4 * i + 2]) * stat[4 * i + 3] * rg + rb;
        v += r;
      }
      if (a.relu) v = fmaxf(v, 0.f);
      const size_t m = (size_t)(m0 + i);
      if (nj < a.split) a.out[m * a.out_stride + nj] = v;
      else a.out2[ppos * a.out2_pos + m * a.out2_stride + (nj - a.split)] = v;
    }
  }
}

// one workgroup per row: the K chunks' sums in chunk order, then bias / (normalised) residual / ReLU as in the tile kernel
__global__ __launch_bounds__(256) void row_mfma_finish_kernel(RowLinArgs a, const float *__restrict__ ws, int nz) {
  touch_args(a);
  __shared__ float red[8];
  long ppos = 0;
  if (a.pos) ppos = *a.pos;
  if (a.res && a.res_pos) a.res += ppos * a.res_pos;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool rln = a.res && a.res_g;
  float rmean = 0.f, rrstd = 1.f;
  if (rln && a.res_stat) {
    rmean = a.res_stat[2 * (size_t)m];
    rrstd = a.res_stat[2 * (size_t)m + 1];
  } else if (rln) {
    const float *rr = a.res + (size_t)m * a.res_stride;
    float s_ = 0.f;
    for (int i = tid; i < a.N; i += 256) s_ += rr[i];
    s_ = wave_sum(s_);
    if (lane == 0) red[wave] = s_;
    __syncthreads();
    rmean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.N;
    float var = 0.f;
    for (int i = tid; i < a.N; i += 256) { const float d = rr[i] - rmean; var += d * d; }
    var = wave_sum(var);
    if (lane == 0) red[4 + wave] = var;
    __syncthreads();
    rrstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)a.N + a.eps);
  }
  for (int n = tid; n < a.N; n += 256) {
    float v = 0.f;
    for (int z = 0; z < nz; ++z) v += ws[((size_t)z * a.M + m) * a.N + n];
    v += a.bias ? a.bias[n] : 0.f;
    if (a.res) {
      float r = a.res[(size_t)m * a.res_stride + n];
      if (rln) r = (r - rmean) * rrstd * a.res_g[n] + a.res_b[n];
      v += r;
    }
    if (a.relu) v = fmaxf(v, 0.f);
    if (n < a.split) a.out[(size_t)m * a.out_stride + n] = v;
    else a.out2[ppos * a.out2_pos + (size_t)m * a.out2_stride + (n - a.split)] = v;
  }
}

bool row_mfma_supported(const RowLinArgs &a) {
  return a.M >= 2 && (a.K & 7) == 0 && (a.x_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && (!a.ln_g || ((reinterpret_cast<uintptr_t>(a.ln_g) | reinterpret_cast<uintptr_t>(a.ln_b)) & 15) == 0);
}

int launch_row_mfma(const RowLinArgs &a, hipStream_t st, float *ksplit_ws = nullptr, size_t ksplit_floats = 0) {
  static DeviceOnce attr_set;       // once, outside any stream capture (the first position runs direct)
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(row_mfma32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kRowMfmaLds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(row_mfma32)");
    attr_set.mark();
  }
  const int tiles = ((a.N + 31) / 32) * ((a.M + 31) / 32), nz = (a.K + MF_KC - 1) / MF_KC;
  if (nz > 1 && tiles < 128 && ksplit_ws && (size_t)nz * a.M * a.N <= ksplit_floats) {
    hipLaunchKernelGGL(row_mfma32_kernel, dim3((a.N + 31) / 32, (a.M + 31) / 32, nz), dim3(256), kRowMfmaLds, st, a, ksplit_ws, knobs().decode_stats_global);
    int rc = check_launch("row_mfma32 (K chunks)");
    if (rc) return rc;
    hipLaunchKernelGGL(row_mfma_finish_kernel, dim3(a.M), dim3(256), 0, st, a, ksplit_ws, nz);
    return check_launch("row_mfma_finish");
  }
  hipLaunchKernelGGL(row_mfma32_kernel, dim3((a.N + 31) / 32, (a.M + 31) / 32), dim3(256), kRowMfmaLds, st, a, (float *)nullptr, knobs().decode_stats_global);
  return check_launch("row_mfma32");
}

// rows of a group: as many as fit (kq * mr <= 16), at most 8
static int row_gemvm_group(int M, int kq) {
  int mr = M <= 2 ? 2 : M <= 4 ? 4 : 8;
  while (kq * mr > 16) mr >>= 1;
  return mr;
}
bool row_gemvm_supported(const RowLinArgs &a) {
  if (a.M < 1 || a.M > 256 || (a.K & 3) || a.K > 2048) return false;
  if (a.res && a.res_g && a.N > 512) return false;
  const int kq = a.K <= 256 ? 1 : a.K <= 512 ? 2 : a.K <= 1024 ? 4 : 8;
  return row_gemvm_group(a.M, kq) >= 2;
}

int launch_row_gemvm(const RowLinArgs &a, hipStream_t st) {
  const int kq = a.K <= 256 ? 1 : a.K <= 512 ? 2 : a.K <= 1024 ? 4 : 8;
  const int mr = row_gemvm_group(a.M, kq);
  dim3 grid(kq >= 4 ? (a.N + 3) / 4 : (a.N + NPB - 1) / NPB), block(256);
#define ISI_GM(KQ_, MR_) hipLaunchKernelGGL((row_gemvm_kernel<KQ_, MR_>), grid, block, 0, st, a)
  if (kq == 1) { if (mr == 2) ISI_GM(1, 2); else if (mr == 4) ISI_GM(1, 4); else ISI_GM(1, 8); }
  else if (kq == 2) { if (mr == 2) ISI_GM(2, 2); else if (mr == 4) ISI_GM(2, 4); else ISI_GM(2, 8); }
  else if (kq == 4) { if (mr == 2) ISI_GM(4, 2); else ISI_GM(4, 4); }
  else ISI_GM(8, 2);
#undef ISI_GM
  return check_launch("row_gemvm");
}

bool row_gemv1_supported(const RowLinArgs &a, bool merged) {
  if (a.M != 1 || (a.K & 3) || a.K > 2048) return false;
  if (a.res && a.res_g && a.N > 512) return false;
  return !merged || a.K <= 512;
}

int launch_row_gemv1(const RowLinArgs &a, const float *part, int NS, int HD, hipStream_t st) {
  Gemv1Args g{a, part, NS, HD, knobs().decode_nt};
  dim3 grid((a.N + NPB - 1) / NPB), grid1((a.N + 3) / 4), block(256);   // grid1: one output feature per wave
  if (a.K <= 256) hipLaunchKernelGGL(row_gemv1_kernel<1>, grid, block, 0, st, g);
  else if (a.K <= 512) hipLaunchKernelGGL(row_gemv1_kernel<2>, grid, block, 0, st, g);
  else if (a.K <= 1024) hipLaunchKernelGGL(row_gemv1_kernel<4>, grid1, block, 0, st, g);
  else hipLaunchKernelGGL(row_gemv1_kernel<8>, grid1, block, 0, st, g);
  return check_launch("row_gemv1");
}

int launch_row_linear_8(const RowLinArgs &a, hipStream_t st);

// more than 8 rows (batched serving): groups of 8 rows, one launch each (the kernel stages its rows in LDS)
int launch_row_linear(const RowLinArgs &a0, hipStream_t st) {
  for (int m0 = 0; m0 < a0.M; m0 += 8) {
    RowLinArgs a = a0;
    a.M = a0.M - m0 < 8 ? a0.M - m0 : 8;
    a.x += (size_t)m0 * a.x_stride;
    if (a.res) a.res += (size_t)m0 * a.res_stride;
    a.out += (size_t)m0 * a.out_stride;
    if (a.out2) a.out2 += (size_t)m0 * a.out2_stride;
    const int rc = launch_row_linear_8(a, st);
    if (rc) return rc;
  }
  return ISI_OK;
}

int launch_row_linear_8(const RowLinArgs &a, hipStream_t st) {
  const int mr = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;   // the kernel's row capacity (template MR)
  const size_t smem = ((size_t)mr * a.K + 2 * mr) * sizeof(float);
  dim3 grid((a.N + NPB - 1) / NPB), block(256);
#define ISI_RL(MR)                                                                                      \
  do {                                                                                                  \
    auto kern = row_linear_ln_kernel<MR>;                                                               \
    static DeviceOnce attr_set;  /* once, outside any stream capture (the first position runs direct) */ \
    if (!attr_set.done()) {                                                                                    \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)   \
        return check_launch("hipFuncSetAttribute(row_linear)");                                         \
      attr_set.mark();                                                                                  \
    }                                                                                                   \
    if (smem > 160 * 1024) return unsupported("row_linear: rows do not fit in LDS");                    \
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);                                                 \
  } while (0)
  if (a.M <= 1) ISI_RL(1);
  else if (a.M <= 2) ISI_RL(2);
  else if (a.M <= 4) ISI_RL(4);
  else ISI_RL(8);
#undef ISI_RL
  return check_launch("row_linear_ln");
}

__global__ void set_pos_kernel(int *pos, int value, int add) { *pos = add ? *pos + value : value; }

// One stage of the decoding loop on M rows: the kernel by row count.  `part` != nullptr (one row): the input row is the
// merge of that attention's key-split partials.
int launch_stage_rows(const RowLinArgs &a, const float *part, int ns, int hd, float *mf_ws, size_t mf_ws_floats, hipStream_t q_st) {
  if (row_gemv1_supported(a, part != nullptr)) return launch_row_gemv1(a, part, ns, hd, q_st);
  // batches: beyond `decode_mfma_rows` rows the stage is a tile GEMM on the fp32 matrix pipe; up to there every row in one
  // launch of the register-resident GEMV kernel (the rows pass through in groups)
  if (a.M > knobs().decode_mfma_rows && row_mfma_supported(a)) return launch_row_mfma(a, q_st, mf_ws, mf_ws_floats);
  if (row_gemvm_supported(a)) return launch_row_gemvm(a, q_st);
  for (int m0 = 0; m0 < a.M; m0 += 8) {
    RowLinArgs g8 = a;
    g8.M = a.M - m0 < 8 ? a.M - m0 : 8;
    g8.x += (size_t)m0 * a.x_stride;
    if (g8.res) g8.res += (size_t)m0 * a.res_stride;
    g8.out += (size_t)m0 * a.out_stride;
    if (g8.out2) g8.out2 += (size_t)m0 * a.out2_stride;
    const int rc8 = row_gemvm_supported(g8) ? launch_row_gemvm(g8, q_st) : launch_row_linear(g8, q_st);
    if (rc8) return rc8;
  }
  return ISI_OK;
}

}  // namespace

// A decoding-loop stage on its own (isi_decode_stage_f32): out[M][N] = act(LN(x)[M][K] W[N][K]^T + bias + LN_res(res)), the
// kernels prior_sample_run launches for M rows.  workspace: decode_stage_workspace_floats(M, N, K) floats, 16-byte aligned.
size_t decode_stage_workspace_floats(int M, int N, int K) { return 4 + (size_t)M * ((K + MF_KC - 1) / MF_KC) * N; }
int decode_stage_f32(const float *x, int x_stride, const float *ln_g, const float *ln_b, const float *W, const float *bias,
                     const float *res, int res_stride, const float *res_g, const float *res_b, float *out, int out_stride,
                     int M, int N, int K, int relu, float eps, float *workspace, size_t workspace_floats, hipStream_t st) {
  if (!x || !W || !out || M <= 0 || M > 256 || N <= 0 || K <= 0) return invalid("decode_stage: bad argument");
  if ((K & 3) || (x_stride & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15))
    return invalid("decode_stage: K and x_stride must be multiples of 4, x and W 16-byte aligned");
  if ((ln_g == nullptr) != (ln_b == nullptr) || (res_g == nullptr) != (res_b == nullptr) || (res_g && !res))
    return invalid("decode_stage: LayerNorm weight and bias come together (and the residual's with a residual)");
  if (res_g && N > 512) return unsupported("decode_stage: a normalised residual has at most 512 features");
  if (K > 2048) return unsupported("decode_stage: K <= 2048");
  RowLinArgs a{x, x_stride, ln_g, ln_b, W, bias, res, res_stride, res_g, res_b, out, out_stride, nullptr, 0, N, M, N, K,
               relu ? 1 : 0, eps, nullptr, 0, 0, 0};
  float *ws = (workspace && !(reinterpret_cast<uintptr_t>(workspace) & 15)) ? workspace : nullptr;
  return launch_stage_rows(a, nullptr, 1, 64, ws, ws ? workspace_floats : 0, st);
}

// partial sums of the K chunks of linear2 (K = dim_feedforward, N = d_model) when a batch decodes on matrix tiles
static size_t mfma_ksplit_floats(const isi_prior_w *w, int B) {
  return 4 + (size_t)B * ((w->dim_feedforward + MF_KC - 1) / MF_KC) * w->d_model;
}
size_t prior_decode_scratch_floats(const isi_prior_w *w, int B) {
  if (!w || B <= 0) return 0;
  const size_t d = w->d_model;
  // q, attn out, y1, y2, y3(a), y3(b), hidden, logits, sampled(int64)
  return (size_t)B * (6 * d + w->dim_feedforward + w->n_class) + 2 * (size_t)B + 64 + 32 +
         rel_attention_decode_workspace_floats(B, w->nhead, w->d_model / w->nhead) + mfma_ksplit_floats(w, B) +
         2 * (size_t)B + 4;     // (+ the rows' LayerNorm statistics handed from launch to launch, RowLinArgs.stat_out)
}

// ---- cache of the decode loop's graph executables (prior_sample_run)
constexpr size_t kDecodeGraphCacheMax = 8;
struct DecodeGraphs {
  std::vector<unsigned char> key;
  hipGraph_t graphs[2] = {nullptr, nullptr};
  hipGraphExec_t execs[2] = {nullptr, nullptr};
  bool failed[2] = {false, false};
  uint64_t used = 0;
  hipEvent_t done = nullptr;      // recorded behind the last replay: what dropping this entry has to wait for
};
static std::mutex &decode_graph_mutex() { static std::mutex m; return m; }
// (called with the mutex held)  nullptr: no device / allocation failure -- the caller then launches directly
static DecodeGraphs *decode_graphs_for(const isi_prior_w *w, const isi_prior_state *s, float temperature, int top_k, float top_p,
                                       int W) {
  static std::vector<DecodeGraphs *> cache;
  static uint64_t clock_ = 0;
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  isi_prior_state sk = *s;
  sk.mask = nullptr;                                   // a HOST array that only steers which windows replay
  const Knobs kn = knobs();
  struct Tail { float temperature, top_p; int top_k, W, device; } tail{temperature, top_p, top_k, W, device};
  std::vector<unsigned char> key(sizeof(isi_prior_w) + sizeof(isi_prior_state) + sizeof(Knobs) + sizeof(Tail));
  unsigned char *k = key.data();
  std::memcpy(k, w, sizeof(isi_prior_w)); k += sizeof(isi_prior_w);
  std::memcpy(k, &sk, sizeof(isi_prior_state)); k += sizeof(isi_prior_state);
  std::memcpy(k, &kn, sizeof(Knobs)); k += sizeof(Knobs);
  std::memcpy(k, &tail, sizeof(Tail));
  for (DecodeGraphs *g : cache)
    if (g->key == key) { g->used = ++clock_; return g; }
  if (cache.size() >= kDecodeGraphCacheMax) {
    size_t lru = 0;
    for (size_t i = 1; i < cache.size(); ++i) if (cache[i]->used < cache[lru]->used) lru = i;
    if (cache[lru]->done) {                            // its last replay may still be running
      (void)hipEventSynchronize(cache[lru]->done);
      (void)hipEventDestroy(cache[lru]->done);
    } else {
      (void)hipDeviceSynchronize();
    }
    for (int j = 0; j < 2; ++j) {
      if (cache[lru]->execs[j]) (void)hipGraphExecDestroy(cache[lru]->execs[j]);
      if (cache[lru]->graphs[j]) (void)hipGraphDestroy(cache[lru]->graphs[j]);
    }
    delete cache[lru];
    cache.erase(cache.begin() + (long)lru);
  }
  DecodeGraphs *g = new (std::nothrow) DecodeGraphs();
  if (!g) return nullptr;
  g->key = std::move(key);
  g->used = ++clock_;
  cache.push_back(g);
  return g;
}

int prior_sample_run(const isi_prior_w *w, const isi_prior_state *s, int p_begin, int p_end, float temperature,
                     int top_k, float top_p, hipStream_t st) {
  if (!w || !s) return invalid("prior_sample_run: null pointer");
  if (w->n_layers <= 0 || w->n_layers > ISI_MAX_LAYERS) return invalid("prior_sample_run: bad layer count");
  if (s->B <= 0 || s->B > 256) return unsupported("prior_sample_run: batch size must be 1..256");
  if (p_begin < 0 || p_end > s->S_t || p_begin > p_end) return invalid("prior_sample_run: bad position range");
  if (!s->x_seq || !s->kv_cache || !s->memory_kv || !s->codes || !s->mask || !s->uniforms || !s->scratch)
    return invalid("prior_sample_run: null state pointer");
  if (s->scratch_floats < prior_decode_scratch_floats(w, s->B)) {
    set_last_error("prior_sample_run: scratch too small");
    return ISI_E_WORKSPACE;
  }
  if (w->d_model % 4 || w->dim_feedforward % 4 || w->d_model % w->nhead) return invalid("prior_sample_run: bad dims");
  const int d = w->d_model, B = s->B, hd = d / w->nhead, ff = w->dim_feedforward;
  float *q = s->scratch, *ao = q + (size_t)B * d, *y1 = ao + (size_t)B * d, *y2 = y1 + (size_t)B * d;
  float *y3a = y2 + (size_t)B * d, *y3b = y3a + (size_t)B * d, *hid = y3b + (size_t)B * d;
  float *logits = hid + (size_t)B * ff;
  int64_t *sampled = reinterpret_cast<int64_t *>(logits + (size_t)B * w->n_class + ((B * w->n_class) & 1));
  float *attn_ws = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(sampled + B) + 8 + 15) & ~(uintptr_t)15);   // 16-byte aligned partial rows
  const size_t cache_layer = (size_t)s->S_t * B * 2 * d, mem_layer = (size_t)s->S_src * B * 2 * d;
  const float scale = 1.0f / sqrtf((float)hd);

  int *pos = reinterpret_cast<int *>(attn_ws + rel_attention_decode_workspace_floats(B, w->nhead, hd));
  pos = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(pos) + 15) & ~(uintptr_t)15);
  float *mf_ws = reinterpret_cast<float *>(pos + 4);                     // (16-byte aligned: K-chunk sums of row_mfma32_kernel)
  const size_t mf_ws_floats = mfma_ksplit_floats(w, B) - 4;
  float *rowstat = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(mf_ws + mf_ws_floats) + 7) & ~(uintptr_t)7);   // [B][2]
  const int i_off = s->start_len - 1;   // token index predicted from position p is p - i_off

  // Every launch of one position; all position-dependent addresses and sizes are derived on the device
  // from *pos, so the same sequence can be captured once into a hipGraph and replayed per position.
  // `p` >= 0: direct launches, the position is passed by value (no dependent load of the counter at the head of
  // every kernel, no counter update); p < 0: replayable launches reading the device counter.
  auto enqueue_position = [&](bool sample, int p, hipStream_t q_st) -> int {
    const int *pos_arg = p < 0 ? pos : nullptr;
    // `part` != nullptr: the input row is the merge of that attention's key-split partials
    auto launch_rows = [&](RowLinArgs a, const float *part = nullptr, int ns = 1) -> int {
      if (p >= 0) {
        a.x += (long)p * a.x_pos;
        if (a.res) a.res += (long)p * a.res_pos;
        if (a.out2) a.out2 += (long)p * a.out2_pos;
        a.pos = nullptr;
      }
      if (!a.x_pos && !a.res_pos && !a.out2_pos) a.pos = nullptr;      // nothing of this launch depends on the position
      return launch_stage_rows(a, part, ns, hd, mf_ws, mf_ws_floats, q_st);
    };
    // statistics hand-off between the two launches that normalise the same rows (tile path only: both must run there)
    auto on_tiles = [&](const RowLinArgs &a) { return a.M > knobs().decode_mfma_rows && row_mfma_supported(a); };
    auto hand_stats = [&](RowLinArgs &producer, RowLinArgs &consumer, bool consumer_merges = false) {
      const bool tiles = on_tiles(producer) && on_tiles(consumer);
      const bool one_row = row_gemv1_supported(producer, false) && row_gemv1_supported(consumer, consumer_merges);   // batch 1
      // batches of up to `decode_mfma_rows` rows: the rows-in-registers kernel (launch_stage_rows picks it in this order)
      const bool few_rows = producer.M > 1 && producer.M <= knobs().decode_mfma_rows && row_gemvm_supported(producer) &&
                            row_gemvm_supported(consumer) && !consumer_merges;
      if ((tiles || one_row || few_rows) && producer.ln_g && consumer.res_g && !knobs().decode_no_stat_handoff) {
        producer.stat_out = rowstat;
        consumer.res_stat = rowstat;
      }
    };
    // the attention's splits are merged by the out-projection when that runs as the one-row kernel
    const int ns_self = rel_attention_decode_splits(s->S_t, B * w->nhead), ns_cross = rel_attention_decode_splits(s->S_src, B * w->nhead);
    const bool merge_in_gemv = B == 1 && d <= 512 && (d & 3) == 0;
    const float *yin = s->x_seq;     // + p * B * d through x_pos / res_pos
    long yin_pos = (long)B * d;
    const float *ln_g = nullptr, *ln_b = nullptr;
    for (int l = 0; l < w->n_layers; ++l) {
      const isi_decoder_layer_w &L = w->layers[l];
      float *cache = s->kv_cache + l * cache_layer;
      const float *memkv = s->memory_kv + l * mem_layer;
      float *y3 = (l & 1) ? y3b : y3a;
      RowLinArgs a;
      int rc;
      // q | k,v  (k,v straight into the cache slot of this position)
      a = RowLinArgs{yin, d, ln_g, ln_b, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias, nullptr, 0, nullptr,
                     nullptr, q, d, cache, 2 * d, d, B, 3 * d, d, 0, 1e-5f, pos, yin_pos, 0, (long)B * 2 * d};
      // y1 = LN_in(yin) + ao Wo^T + bo   (its launch follows the attention)
      RowLinArgs a_o{ao, d, nullptr, nullptr, L.self_attn.out_proj_weight, L.self_attn.out_proj_bias, yin, d, ln_g,
                     ln_b, y1, d, nullptr, 0, d, B, d, d, 0, 1e-5f, pos, 0, yin_pos, 0};
      if (ln_g) hand_stats(a, a_o, merge_in_gemv && ns_self > 1);
      if ((rc = launch_rows(a))) return rc;
      isi_attn_args g;
      memset(&g, 0, sizeof g);
      g.q = q; g.k = cache; g.v = cache + d; g.rel_embeddings = L.self_attn.rel_embeddings; g.out = ao;
      g.Sq = 1; g.Sk = s->S_t; g.B = B; g.H = w->nhead; g.head_dim = hd;
      g.q_sb = d; g.q_sh = hd; g.k_ss = (int64_t)B * 2 * d; g.k_sb = 2 * d; g.k_sh = hd;
      g.v_ss = g.k_ss; g.v_sb = g.k_sb; g.v_sh = hd; g.o_sb = d; g.o_sh = hd;
      g.Cq = w->Cd; g.Ck = w->Cd; g.Ek = w->Ed; g.rel_rows = L.self_attn.rel_rows; g.scale = scale;
      const bool defer_s = merge_in_gemv && ns_self > 1;
      if ((rc = rel_attention_decode_launch(&g, p < 0 ? 0 : p, pos_arg, 1, attn_ws, defer_s ? 0 : 1, q_st))) return rc;
      if ((rc = defer_s ? launch_rows(a_o, attn_ws, ns_self) : launch_rows(a_o))) return rc;
      // cross-attention query from LN1(y1)
      a = RowLinArgs{y1, d, L.norm1_w, L.norm1_b, L.cross_attn.in_proj_weight, L.cross_attn.in_proj_bias, nullptr, 0,
                     nullptr, nullptr, q, d, nullptr, 0, d, B, d, d, 0, 1e-5f, nullptr, 0, 0, 0};
      RowLinArgs a_o2{ao, d, nullptr, nullptr, L.cross_attn.out_proj_weight, L.cross_attn.out_proj_bias, y1, d,
                      L.norm1_w, L.norm1_b, y2, d, nullptr, 0, d, B, d, d, 0, 1e-5f, nullptr, 0, 0, 0};
      hand_stats(a, a_o2, merge_in_gemv && ns_cross > 1);
      if ((rc = launch_rows(a))) return rc;
      g.k = memkv; g.v = memkv + d; g.rel_embeddings = L.cross_attn.rel_embeddings; g.Sk = s->S_src;
      g.Ck = w->Ce; g.Ek = w->Ee; g.rel_rows = L.cross_attn.rel_rows;
      const bool defer_c = merge_in_gemv && ns_cross > 1;
      if ((rc = rel_attention_decode_launch(&g, p < 0 ? 0 : p, pos_arg, 0, attn_ws, defer_c ? 0 : 1, q_st))) return rc;
      if ((rc = defer_c ? launch_rows(a_o2, attn_ws, ns_cross) : launch_rows(a_o2))) return rc;
      // feed-forward on LN2(y2)
      a = RowLinArgs{y2, d, L.norm2_w, L.norm2_b, L.linear1_w, L.linear1_b, nullptr, 0, nullptr, nullptr, hid, ff,
                     nullptr, 0, ff, B, ff, d, 1, 1e-5f, nullptr, 0, 0, 0};
      RowLinArgs a_f2{hid, ff, nullptr, nullptr, L.linear2_w, L.linear2_b, y2, d, L.norm2_w, L.norm2_b, y3, d, nullptr,
                      0, d, B, d, ff, 0, 1e-5f, nullptr, 0, 0, 0};
      hand_stats(a, a_f2);
      if ((rc = launch_rows(a))) return rc;
      if ((rc = launch_rows(a_f2))) return rc;
      yin = y3; yin_pos = 0; ln_g = L.norm3_w; ln_b = L.norm3_b;
    }
    if (sample) {
      RowLinArgs a{yin, d, ln_g, ln_b, w->logits_w, w->logits_b, nullptr, 0, nullptr, nullptr, logits, w->n_class,
                   nullptr, 0, w->n_class, B, w->n_class, d, 0, 1e-5f, nullptr, 0, 0, 0};
      int rc;
      if ((rc = launch_rows(a))) return rc;
      // (a replayed position of ONE sequence: the commit also advances the position counter)
      const bool fold_advance = p < 0 && B == 1;
      const SampleCommit cm{w->embed_table, w->eff_dim, s->codes, s->S, p, i_off, s->S_t, s->x_seq, d, fold_advance ? pos : nullptr};
      if ((rc = sample_row_commit_f32(logits, w->n_class, B, w->n_class, temperature, top_k, top_p,
                                      p < 0 ? s->uniforms : s->uniforms + (size_t)(p - i_off) * B, sampled, nullptr, pos_arg,
                                      i_off, cm, q_st)))
        return rc;
      if (fold_advance) return ISI_OK;
    }
    if (p >= 0) return ISI_OK;
    hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, q_st, pos, 1, 1);
    return check_launch("advance_position");
  };
  auto sampled_at = [&](int p) {
    const int i = p - i_off;
    return i >= 0 && i < s->S && s->mask[i];
  };

  if (p_begin == p_end) return ISI_OK;

  // the first position runs directly (it also performs the one-time kernel attribute set-up) ...
  int rc;
  if ((rc = enqueue_position(sampled_at(p_begin), p_begin, st))) return rc;
  int p = p_begin + 1;
  // ... the rest can replay captured graphs of W = ISI_PRIOR_GRAPH consecutive positions (round 5; 1 = the round-2 form, one
  // position per graph): the ~66 launches of a position -- W x 66 per graph -- collapse into one graph launch and the host
  // leaves the loop (direct launches cost it ~230 us per token against ~280 us of kernels: on a slower host the loop was
  // host-bound).  Every position-dependent address is read from a device counter the last node of a
  // position advances.  Two variants exist -- windows whose positions are all sampled / all kept --, each captured when
  // first needed; windows of mixed positions and the tail run as direct launches (the counter is re-set after them).
  // (not while the CALLER is capturing `st`: the synchronisation below would invalidate that capture -- direct launches then,
  // which a capture takes as they are)
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap_status) != hipSuccess) { (void)hipGetLastError(); cap_status = hipStreamCaptureStatusNone; }
  const int W = cap_status == hipStreamCaptureStatusNone ? knobs().prior_graph : 0;
  if (W > 0 && p_end - p >= (W > 2 ? 2 * W : 4)) {
    // The executables are CACHED: a graph bakes in nothing but the launches' arguments -- every pointer and size
    // of *w and *state (the host mask aside), the sampling parameters, W and the library switches --, so a call whose
    // arguments compare equal byte for byte replays the graphs an earlier call captured.  Nothing is destroyed at the end of
    // a call, hence nothing to wait for: the call returns as soon as its launches are enqueued, like every other entry
    // point.  At most kDecodeGraphCacheMax argument sets are kept; the least recently used one is dropped once the event
    // recorded behind its last replay has passed.
    std::lock_guard<std::mutex> lock(decode_graph_mutex());
    DecodeGraphs *G = decode_graphs_for(w, s, temperature, top_k, top_p, W);
    hipStream_t cap = nullptr;
    bool ok = G != nullptr;
    auto build = [&](int k) -> bool {      // k = 1: every position of the window samples
      if (G->execs[k] || G->failed[k]) return G->execs[k] != nullptr;
      bool good = (cap != nullptr || hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess) &&
                  hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
      int erc = ISI_OK;
      for (int i = 0; good && i < W && erc == ISI_OK; ++i) erc = enqueue_position(k == 1, -1, cap);
      hipGraph_t gph = nullptr;
      const bool ended = good && hipStreamEndCapture(cap, &gph) == hipSuccess;
      G->graphs[k] = gph;
      good = good && erc == ISI_OK && ended && gph != nullptr && hipGraphInstantiate(&G->execs[k], gph, nullptr, nullptr, 0) == hipSuccess;
      if (!good) { G->failed[k] = true; G->execs[k] = nullptr; (void)hipGetLastError(); }
      return good;
    };
    int counter = -1;                      // value of the device counter (-1: unknown)
    while (ok && p < p_end && rc == ISI_OK) {
      bool uniform = p + W <= p_end;
      const bool flag = sampled_at(p);
      for (int i = 1; uniform && i < W; ++i) uniform = sampled_at(p + i) == flag;
      if (uniform && build(flag ? 1 : 0)) {
        if (counter != p) {
          hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, st, pos, p, 0);
          if ((rc = check_launch("set_position"))) break;
        }
        if (hipGraphLaunch(G->execs[flag ? 1 : 0], st) != hipSuccess) { rc = check_launch("hipGraphLaunch(prior positions)"); break; }
        p += W;
        counter = p;
      } else {
        if ((rc = enqueue_position(flag, p, st))) break;
        ++p;
      }
    }
    if (cap) (void)hipStreamDestroy(cap);
    if (G && (G->execs[0] || G->execs[1])) {
      if (!G->done && hipEventCreateWithFlags(&G->done, hipEventDisableTiming) != hipSuccess) { G->done = nullptr; (void)hipGetLastError(); }
      if (G->done && hipEventRecord(G->done, st) != hipSuccess) (void)hipGetLastError();
    }
    if (rc) return rc;
  }
  for (; p < p_end; ++p)
    if ((rc = enqueue_position(sampled_at(p), p, st))) return rc;
  return ISI_OK;
}

}  // namespace isi
