// Native key/value-cached sampling loop of the prior's decoder (gfx950, fp32).
//
// One call enqueues, for every sequence position in [p_begin, p_end), the whole
// decoder stack on ONE new row (8 launches per layer), the logits head, the
// categorical draw and the write of the sampled token's embedding into the next
// input row -- with no host synchronisation: the sampled index stays on the
// device.  This replaces the reference's per-token full decoder pass
// (sample.py:268-305 -> priors/transformer.py:763-774): decoder self-attention is
// causal and earlier inputs never change, so row p computed from cached keys /
// values equals row p of a full pass (tests/test_prior_gpu.py checks it).
//
// LayerNorms are folded into their consumers: each GEMV normalises its input
// rows (and, when the residual is a normalised tensor, its residual rows) on the
// fly from the stored pre-norm rows.
#include "isi_common.h"
#include "isi_internal.h"

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
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int MR>
__global__ __launch_bounds__(256) void row_linear_ln_kernel(const RowLinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *xs = sm;                    // [MR][K]
  float *stat = sm + MR * a.K;       // [MR][2] residual mean / rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.K >> 2;
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
  // ---- phase 1: NPB output features per workgroup, 2 per wave, both weight rows in flight
  const int n0 = blockIdx.x * NPB + wave * 2;
  if (n0 >= a.N) return;
  const bool two = n0 + 1 < a.N;
  const float4 *w0 = reinterpret_cast<const float4 *>(a.W + (size_t)n0 * a.K);
  const float4 *w1 = reinterpret_cast<const float4 *>(a.W + (size_t)(two ? n0 + 1 : n0) * a.K);
  float acc0[MR], acc1[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
  for (int qd = lane; qd < nq; qd += 64) {
    const float4 wa = w0[qd], wb = w1[qd];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < a.M) {
        const float4 xv = reinterpret_cast<const float4 *>(xs + (size_t)m * a.K)[qd];
        acc0[m] += (wa.x * xv.x + wa.y * xv.y) + (wa.z * xv.z + wa.w * xv.w);
        acc1[m] += (wb.x * xv.x + wb.y * xv.y) + (wb.z * xv.z + wb.w * xv.w);
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

int launch_row_linear(const RowLinArgs &a, hipStream_t st) {
  const size_t smem = ((size_t)a.M * a.K + 2 * a.M) * sizeof(float);
  dim3 grid((a.N + NPB - 1) / NPB), block(256);
#define ISI_RL(MR)                                                                                      \
  do {                                                                                                  \
    auto kern = row_linear_ln_kernel<MR>;                                                               \
    if (smem > 48 * 1024) {                                                                             \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)    \
        return check_launch("hipFuncSetAttribute(row_linear)");                                         \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);                                                 \
  } while (0)
  if (a.M <= 1) ISI_RL(1);
  else if (a.M <= 2) ISI_RL(2);
  else if (a.M <= 4) ISI_RL(4);
  else ISI_RL(8);
#undef ISI_RL
  return check_launch("row_linear_ln");
}

// codes[b, i] = sampled[b]; x_next[b, 0:eff] = table[sampled[b], :]
__global__ void commit_token_kernel(const int64_t *__restrict__ sampled, const float *__restrict__ table,
                                    int eff, int64_t *__restrict__ codes, int codes_stride, int i,
                                    float *__restrict__ x_next, int x_stride) {
  const int b = blockIdx.x;
  const int64_t tok = sampled[b];
  if (threadIdx.x == 0) codes[(size_t)b * codes_stride + i] = tok;
  for (int e = threadIdx.x; e < eff; e += blockDim.x) x_next[(size_t)b * x_stride + e] = table[(size_t)tok * eff + e];
}

}  // namespace

size_t prior_decode_scratch_floats(const isi_prior_w *w, int B) {
  if (!w || B <= 0) return 0;
  const size_t d = w->d_model;
  // q, attn out, y1, y2, y3(a), y3(b), hidden, logits, sampled(int64)
  return (size_t)B * (6 * d + w->dim_feedforward + w->n_class) + 2 * (size_t)B + 64 +
         rel_attention_decode_workspace_floats(B, w->nhead, w->d_model / w->nhead);
}

int prior_sample_run(const isi_prior_w *w, const isi_prior_state *s, int p_begin, int p_end, float temperature,
                     int top_k, float top_p, hipStream_t st) {
  if (!w || !s) return invalid("prior_sample_run: null pointer");
  if (w->n_layers <= 0 || w->n_layers > ISI_MAX_LAYERS) return invalid("prior_sample_run: bad layer count");
  if (s->B <= 0 || s->B > 8) return unsupported("prior_sample_run: batch size must be 1..8");
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
  float *attn_ws = reinterpret_cast<float *>(sampled + B) + 2;
  const size_t cache_layer = (size_t)s->S_t * B * 2 * d, mem_layer = (size_t)s->S_src * B * 2 * d;
  const float scale = 1.0f / sqrtf((float)hd);

  for (int p = p_begin; p < p_end; ++p) {
    const float *yin = s->x_seq + (size_t)p * B * d;
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
                     nullptr, q, d, cache + (size_t)p * B * 2 * d, 2 * d, d, B, 3 * d, d, 0, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      isi_attn_args g;
      memset(&g, 0, sizeof g);
      g.q = q; g.k = cache; g.v = cache + d; g.rel_embeddings = L.self_attn.rel_embeddings; g.out = ao;
      g.Sq = 1; g.Sk = p + 1; g.B = B; g.H = w->nhead; g.head_dim = hd;
      g.q_sb = d; g.q_sh = hd; g.k_ss = (int64_t)B * 2 * d; g.k_sb = 2 * d; g.k_sh = hd;
      g.v_ss = g.k_ss; g.v_sb = g.k_sb; g.v_sh = hd; g.o_sb = d; g.o_sh = hd;
      g.Cq = w->Cd; g.Ck = w->Cd; g.Ek = w->Ed; g.rel_rows = L.self_attn.rel_rows; g.scale = scale;
      if ((rc = rel_attention_decode_f32(&g, p, attn_ws, st))) return rc;
      // y1 = LN_in(yin) + ao Wo^T + bo
      a = RowLinArgs{ao, d, nullptr, nullptr, L.self_attn.out_proj_weight, L.self_attn.out_proj_bias, yin, d, ln_g,
                     ln_b, y1, d, nullptr, 0, d, B, d, d, 0, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      // cross-attention query from LN1(y1)
      a = RowLinArgs{y1, d, L.norm1_w, L.norm1_b, L.cross_attn.in_proj_weight, L.cross_attn.in_proj_bias, nullptr, 0,
                     nullptr, nullptr, q, d, nullptr, 0, d, B, d, d, 0, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      g.k = memkv; g.v = memkv + d; g.rel_embeddings = L.cross_attn.rel_embeddings; g.Sk = s->S_src;
      g.Ck = w->Ce; g.Ek = w->Ee; g.rel_rows = L.cross_attn.rel_rows;
      if ((rc = rel_attention_decode_f32(&g, p, attn_ws, st))) return rc;
      a = RowLinArgs{ao, d, nullptr, nullptr, L.cross_attn.out_proj_weight, L.cross_attn.out_proj_bias, y1, d,
                     L.norm1_w, L.norm1_b, y2, d, nullptr, 0, d, B, d, d, 0, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      // feed-forward on LN2(y2)
      a = RowLinArgs{y2, d, L.norm2_w, L.norm2_b, L.linear1_w, L.linear1_b, nullptr, 0, nullptr, nullptr, hid, ff,
                     nullptr, 0, ff, B, ff, d, 1, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      a = RowLinArgs{hid, ff, nullptr, nullptr, L.linear2_w, L.linear2_b, y2, d, L.norm2_w, L.norm2_b, y3, d, nullptr,
                     0, d, B, d, ff, 0, 1e-5f};
      if ((rc = launch_row_linear(a, st))) return rc;
      yin = y3; ln_g = L.norm3_w; ln_b = L.norm3_b;
    }
    const int i = p - (s->start_len - 1);  // token predicted from position p
    if (i < 0 || i >= s->S || !s->mask[i]) continue;
    RowLinArgs a{yin, d, ln_g, ln_b, w->logits_w, w->logits_b, nullptr, 0, nullptr, nullptr, logits, w->n_class,
                 nullptr, 0, w->n_class, B, w->n_class, d, 0, 1e-5f};
    int rc;
    if ((rc = launch_row_linear(a, st))) return rc;
    if ((rc = sample_row_f32(logits, w->n_class, B, w->n_class, temperature, top_k, top_p,
                             s->uniforms + (size_t)i * B, sampled, nullptr, st)))
      return rc;
    if (i + s->start_len < s->S_t) {
      hipLaunchKernelGGL(commit_token_kernel, dim3(B), dim3(256), 0, st, sampled, w->embed_table, w->eff_dim,
                         s->codes, s->S, i, s->x_seq + (size_t)(i + s->start_len) * B * d, d);
      if ((rc = check_launch("commit_token"))) return rc;
    }
  }
  return ISI_OK;
}

}  // namespace isi
