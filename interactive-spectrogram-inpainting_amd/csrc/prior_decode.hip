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
  if (a.res && a.res_g) {
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
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        xv[i].x = (xv[i].x - mean) * rstd * gq[i].x + bq[i].x; xv[i].y = (xv[i].y - mean) * rstd * gq[i].y + bq[i].y;
        xv[i].z = (xv[i].z - mean) * rstd * gq[i].z + bq[i].z; xv[i].w = (xv[i].w - mean) * rstd * gq[i].w + bq[i].w;
      }
    }
  }
  float rmean = 0.f, rrstd = 1.f;
  if (a.res && a.res_g) {
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

// ---- a few rows (batched decoding, up to 8 sequences per launch): the same kernel shape with the rows' inputs in
// registers -- weights are loaded once and used for every row.  Per row the operations of row_gemv1_kernel /
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
  float4 wa[KQ], wb[KQ], gq[KQ], bq[KQ], xv[MR][KQ];
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
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int qd = lane + 64 * i;
      xv[m][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M && qd < nq) xv[m][i] = reinterpret_cast<const float4 *>(a.x + (size_t)m * a.x_stride)[qd];
    }
  const int n = n0 + (lane & 1);
  const bool writer = lane < 2 && (lane == 0 || two);
  float bias_v = 0.f, rg = 1.f, rb = 0.f, res_v[MR];
  if (writer) {
    if (a.bias) bias_v = a.bias[n];
    if (a.res && a.res_g) { rg = a.res_g[n]; rb = a.res_b[n]; }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) res_v[m] = (writer && a.res && m < a.M) ? a.res[(size_t)m * a.res_stride + n] : 0.f;
  constexpr int RS = 8;
  float rrow[MR][RS];
  if (a.res && a.res_g) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int c = lane + 64 * i;
        rrow[m][i] = (m < a.M && c < a.N) ? a.res[(size_t)m * a.res_stride + c] : 0.f;
      }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m >= a.M) break;
    if (a.ln_g) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) if (lane + 64 * i < nq) s += (xv[m][i].x + xv[m][i].y) + (xv[m][i].z + xv[m][i].w);
      const float mean = wave_sum(s) / (float)a.K;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) {
        if (lane + 64 * i < nq) {
          const float d0 = xv[m][i].x - mean, d1 = xv[m][i].y - mean, d2 = xv[m][i].z - mean, d3 = xv[m][i].w - mean;
          var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)a.K + a.eps);
#pragma unroll
      for (int i = 0; i < KQ; ++i) {
        if (lane + 64 * i < nq) {
          xv[m][i].x = (xv[m][i].x - mean) * rstd * gq[i].x + bq[i].x; xv[m][i].y = (xv[m][i].y - mean) * rstd * gq[i].y + bq[i].y;
          xv[m][i].z = (xv[m][i].z - mean) * rstd * gq[i].z + bq[i].z; xv[m][i].w = (xv[m][i].w - mean) * rstd * gq[i].w + bq[i].w;
        }
      }
    }
    float rmean = 0.f, rrstd = 1.f;
    if (a.res && a.res_g) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) s += rrow[m][i];
      rmean = wave_sum(s) / (float)a.N;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < RS; ++i) if (lane + 64 * i < a.N) { const float dv = rrow[m][i] - rmean; var += dv * dv; }
      rrstd = 1.0f / sqrtf(wave_sum(var) / (float)a.N + a.eps);
    }
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (lane + 64 * i < nq) {
        acc0 += (wa[i].x * xv[m][i].x + wa[i].y * xv[m][i].y) + (wa[i].z * xv[m][i].z + wa[i].w * xv[m][i].w);
        acc1 += (wb[i].x * xv[m][i].x + wb[i].y * xv[m][i].y) + (wb[i].z * xv[m][i].z + wb[i].w * xv[m][i].w);
      }
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    if (writer) {
      float v = (lane == 0 ? acc0 : acc1) + bias_v;
      if (a.res) {
        float r = res_v[m];
        if (a.res_g) r = (r - rmean) * rrstd * rg + rb;
        v += r;
      }
      if (a.relu) v = fmaxf(v, 0.f);
      if (n < a.split) a.out[(size_t)m * a.out_stride + n] = v;
      else a.out2[ppos * a.out2_pos + (size_t)m * a.out2_stride + (n - a.split)] = v;
    }
  }
}

// rows x float4-per-lane held in registers: at most 16 (64 VGPRs)
bool row_gemvm_supported(const RowLinArgs &a) {
  if (a.M < 1 || a.M > 8 || (a.K & 3) || a.K > 2048) return false;
  if (a.res && a.res_g && a.N > 512) return false;
  const int kq = a.K <= 256 ? 1 : a.K <= 512 ? 2 : a.K <= 1024 ? 4 : 8;
  const int mr = a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
  return kq * mr <= 16;
}

int launch_row_gemvm(const RowLinArgs &a, hipStream_t st) {
  const int kq = a.K <= 256 ? 1 : a.K <= 512 ? 2 : a.K <= 1024 ? 4 : 8;
  const int mr = a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
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

}  // namespace

size_t prior_decode_scratch_floats(const isi_prior_w *w, int B) {
  if (!w || B <= 0) return 0;
  const size_t d = w->d_model;
  // q, attn out, y1, y2, y3(a), y3(b), hidden, logits, sampled(int64)
  return (size_t)B * (6 * d + w->dim_feedforward + w->n_class) + 2 * (size_t)B + 64 + 32 +
         rel_attention_decode_workspace_floats(B, w->nhead, w->d_model / w->nhead);
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
      if (row_gemv1_supported(a, part != nullptr)) return launch_row_gemv1(a, part, ns, hd, q_st);
      // batches: groups of up to 8 rows through the register-resident kernel where the rows fit
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
    };
    // the attention's splits are merged by the out-projection when that runs as the one-row kernel
    const int ns_self = rel_attention_decode_splits(s->S_t), ns_cross = rel_attention_decode_splits(s->S_src);
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
      // y1 = LN_in(yin) + ao Wo^T + bo
      a = RowLinArgs{ao, d, nullptr, nullptr, L.self_attn.out_proj_weight, L.self_attn.out_proj_bias, yin, d, ln_g,
                     ln_b, y1, d, nullptr, 0, d, B, d, d, 0, 1e-5f, pos, 0, yin_pos, 0};
      if ((rc = defer_s ? launch_rows(a, attn_ws, ns_self) : launch_rows(a))) return rc;
      // cross-attention query from LN1(y1)
      a = RowLinArgs{y1, d, L.norm1_w, L.norm1_b, L.cross_attn.in_proj_weight, L.cross_attn.in_proj_bias, nullptr, 0,
                     nullptr, nullptr, q, d, nullptr, 0, d, B, d, d, 0, 1e-5f, nullptr, 0, 0, 0};
      if ((rc = launch_rows(a))) return rc;
      g.k = memkv; g.v = memkv + d; g.rel_embeddings = L.cross_attn.rel_embeddings; g.Sk = s->S_src;
      g.Ck = w->Ce; g.Ek = w->Ee; g.rel_rows = L.cross_attn.rel_rows;
      const bool defer_c = merge_in_gemv && ns_cross > 1;
      if ((rc = rel_attention_decode_launch(&g, p < 0 ? 0 : p, pos_arg, 0, attn_ws, defer_c ? 0 : 1, q_st))) return rc;
      a = RowLinArgs{ao, d, nullptr, nullptr, L.cross_attn.out_proj_weight, L.cross_attn.out_proj_bias, y1, d,
                     L.norm1_w, L.norm1_b, y2, d, nullptr, 0, d, B, d, d, 0, 1e-5f, nullptr, 0, 0, 0};
      if ((rc = defer_c ? launch_rows(a, attn_ws, ns_cross) : launch_rows(a))) return rc;
      // feed-forward on LN2(y2)
      a = RowLinArgs{y2, d, L.norm2_w, L.norm2_b, L.linear1_w, L.linear1_b, nullptr, 0, nullptr, nullptr, hid, ff,
                     nullptr, 0, ff, B, ff, d, 1, 1e-5f, nullptr, 0, 0, 0};
      if ((rc = launch_rows(a))) return rc;
      a = RowLinArgs{hid, ff, nullptr, nullptr, L.linear2_w, L.linear2_b, y2, d, L.norm2_w, L.norm2_b, y3, d, nullptr,
                     0, d, B, d, ff, 0, 1e-5f, nullptr, 0, 0, 0};
      if ((rc = launch_rows(a))) return rc;
      yin = y3; yin_pos = 0; ln_g = L.norm3_w; ln_b = L.norm3_b;
    }
    if (sample) {
      RowLinArgs a{yin, d, ln_g, ln_b, w->logits_w, w->logits_b, nullptr, 0, nullptr, nullptr, logits, w->n_class,
                   nullptr, 0, w->n_class, B, w->n_class, d, 0, 1e-5f, nullptr, 0, 0, 0};
      int rc;
      if ((rc = launch_rows(a))) return rc;
      const SampleCommit cm{w->embed_table, w->eff_dim, s->codes, s->S, p, i_off, s->S_t, s->x_seq, d};
      if ((rc = sample_row_commit_f32(logits, w->n_class, B, w->n_class, temperature, top_k, top_p,
                                      p < 0 ? s->uniforms : s->uniforms + (size_t)(p - i_off) * B, sampled, nullptr, pos_arg,
                                      i_off, cm, q_st)))
        return rc;
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
  const int W = knobs().prior_graph;
  if (W > 0 && p_end - p >= (W > 2 ? 2 * W : 4)) {
    hipStream_t cap = nullptr;
    hipGraph_t graphs[2] = {nullptr, nullptr};
    hipGraphExec_t execs[2] = {nullptr, nullptr};
    bool failed[2] = {false, false};
    bool ok = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess;
    auto build = [&](int k) -> bool {      // k = 1: every position of the window samples
      if (execs[k] || failed[k]) return execs[k] != nullptr;
      bool good = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
      int erc = ISI_OK;
      for (int i = 0; good && i < W && erc == ISI_OK; ++i) erc = enqueue_position(k == 1, -1, cap);
      hipGraph_t gph = nullptr;
      const bool ended = good && hipStreamEndCapture(cap, &gph) == hipSuccess;
      graphs[k] = gph;
      good = good && erc == ISI_OK && ended && gph != nullptr && hipGraphInstantiate(&execs[k], gph, nullptr, nullptr, 0) == hipSuccess;
      if (!good) { failed[k] = true; (void)hipGetLastError(); }
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
        if (hipGraphLaunch(execs[flag ? 1 : 0], st) != hipSuccess) { rc = check_launch("hipGraphLaunch(prior positions)"); break; }
        p += W;
        counter = p;
      } else {
        if ((rc = enqueue_position(flag, p, st))) break;
        ++p;
      }
    }
    if (execs[0] || execs[1]) (void)hipStreamSynchronize(st);   // executables must outlive their launches
    for (int k = 0; k < 2; ++k) {
      if (execs[k]) (void)hipGraphExecDestroy(execs[k]);
      if (graphs[k]) (void)hipGraphDestroy(graphs[k]);
    }
    if (cap) (void)hipStreamDestroy(cap);
    if (rc) return rc;
  }
  for (; p < p_end; ++p)
    if ((rc = enqueue_position(sampled_at(p), p, st))) return rc;
  return ISI_OK;
}

}  // namespace isi
