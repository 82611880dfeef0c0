// One sequence position of the prior's decoder stack in ONE persistent kernel (gfx950, fp32).
//
// EXPERIMENTAL, opt-in (ISI_PRIOR_PERSISTENT=1): correct (tests run it against the default path) but
// slower on MI355X -- 0.89 ms per position against 0.50 ms -- see the note at grid_barrier().
//
// The multi-kernel path of prior_decode.hip spends a position's 0.5 ms in ~85 dependent
// kernels whose own duration (4.5-9 us each for <= 4 MB of weights) is mostly dispatch and
// drain (profiles/r01_prior_sampling_kernel_trace.txt).  Here a cooperative grid of one
// workgroup per CU walks the same phases -- per layer: q|k|v GEMV, self-attention over the
// cached keys (split over workgroups), out-projection, cross-query GEMV, cross-attention,
// out-projection, feed-forward 1 and 2 -- separated by grid barriers (a release/acquire pair
// at agent scope around one atomic counter) instead of kernel boundaries.  LayerNorms stay
// folded into their consumers, the merge of the attention's key splits is done by the
// out-projection while it stages its input row.  Same arithmetic and summation order per
// output element as the multi-kernel path (tests compare both with the full-pass rows).
//
// Replaces the per-token decoder pass of the reference's sampling loop
// (sample.py:268-305 -> priors/transformer.py:763-774) together with prior_decode.hip.
#include <hip/hip_cooperative_groups.h>

#include "isi_common.h"
#include "isi_internal.h"

namespace isi {

namespace {
constexpr int NPB = 8;        // output features per GEMV work item (2 per wave)
constexpr int NSPLIT = 8;     // key splits of the attention phases
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct PersistArgs {
  isi_prior_w w;
  float *x_seq, *kv_cache;
  const float *memory_kv;
  float *q, *y1, *y2, *y3a, *y3b, *hid, *logits, *part;   // scratch rows / attention partials
  unsigned *bar;                                           // [0..1] barrier counters (by position parity), [2] error flag
  int S_t, S_src, B, p, want_logits;
  float scale;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Rows exchanged between workgroups inside one launch (q, y1..y3, hidden, attention partials, the
// new key/value row) are written and read with agent-scope relaxed atomics: write-through stores and
// cache-bypassing loads (sc1), so that the barrier needs NO cache write-back / invalidate -- an
// agent-scope release/acquire fence pair costs ~20 us per barrier on the eight L2s of this part.
// Weights, biases, LayerNorm parameters, older cache rows and embeddings are read-only in a launch
// and use ordinary cached loads.
__device__ __forceinline__ float ldc(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All workgroups of the grid arrive; rows written with stc() before are visible to ldc() after.
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target, unsigned *err) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned n = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (++n > SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

struct Gemv {
  const float *x; int x_stride;       // [M,K] input rows (nullable when part)
  const float *ln_g, *ln_b;           // LayerNorm on the input rows
  const float *part;                  // attention partials [M][H][NSPLIT][HD+2] to merge into the input rows
  const float *W, *bias;              // [N,K]
  const float *res; int res_stride;   // residual rows
  const float *res_g, *res_b;         // LayerNorm on the residual rows
  float *out; int out_stride;         // columns [0, split)
  float *out2; int out2_stride;       // columns [split, N)
  int split, N, K, relu;
};

// out[m, n] = [relu]( LN?(x[m,:]) . W[n,:] + bias[n] + LN?(res[m,n]) ) for M <= MR rows.
template <int MR, int HD>
__device__ __attribute__((noinline)) void gemv_phase(const Gemv &g, int M, int H, float *sm) {
  const int n_items = (g.N + NPB - 1) / NPB;
  if ((int)blockIdx.x >= n_items) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = g.K >> 2;
  float *xs = sm;                 // [MR][K]
  float *stat = sm + MR * g.K;    // [MR][2]
  constexpr int WPF = 8;
  // first item's weight rows: requested before the input rows are staged
  int item = blockIdx.x;
  int n0 = item * NPB + wave * 2;
  float4 wa[WPF], wb[WPF];
#pragma unroll
  for (int i = 0; i < WPF; ++i) { wa[i] = make_float4(0.f, 0.f, 0.f, 0.f); wb[i] = wa[i]; }
  auto load_w = [&](int n0_) {
    const bool ok0 = n0_ < g.N, ok1 = n0_ + 1 < g.N;
    const float4 *w0 = reinterpret_cast<const float4 *>(g.W + (size_t)(ok0 ? n0_ : 0) * g.K);
    const float4 *w1 = reinterpret_cast<const float4 *>(g.W + (size_t)(ok1 ? n0_ + 1 : (ok0 ? n0_ : 0)) * g.K);
#pragma unroll
    for (int i = 0; i < WPF; ++i) {
      const int qd = lane + 64 * i;
      if (qd < nq) { wa[i] = w0[qd]; wb[i] = w1[qd]; }
    }
  };
  load_w(n0);
  // ---- stage the input rows
  if (g.part) {   // merge of the key splits: all loads first, then the arithmetic
    constexpr int ST = HD + 2;
    for (int i = tid; i < M * g.K; i += 256) {
      const int m = i / g.K, c = i - m * g.K;
      const int h = c / HD, dd = c - h * HD;
      const float *pp = g.part + ((size_t)m * H + h) * NSPLIT * ST;
      float vmax[NSPLIT], vsum[NSPLIT], val[NSPLIT];
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s) { vmax[s] = ldc(pp + s * ST + HD); vsum[s] = ldc(pp + s * ST + HD + 1); val[s] = ldc(pp + s * ST + dd); }
      float mx = -1e30f;
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s) mx = fmaxf(mx, vmax[s]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s) {
        const float wgt = expf(vmax[s] - mx);
        num += wgt * val[s];
        den += wgt * vsum[s];
      }
      xs[i] = num / den;
    }
  }
  for (int m = wave; m < M; m += 4) {
    if (!g.part) {
      const float *xr = g.x + (size_t)m * g.x_stride;
      float *xm = xs + (size_t)m * g.K;
      float s = 0.f;
      for (int c = lane; c < g.K; c += 64) { const float v = ldc(xr + c); xm[c] = v; s += v; }
      if (g.ln_g) {   // two-pass statistics on the staged row, then normalise in place
        const float mean = wave_sum(s) / (float)g.K;
        float var = 0.f;
        for (int c = lane; c < g.K; c += 64) { const float dlt = xm[c] - mean; var += dlt * dlt; }
        const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)g.K + 1e-5f);
        for (int c = lane; c < g.K; c += 64) xm[c] = (xm[c] - mean) * rstd * g.ln_g[c] + g.ln_b[c];
      }
    }
    if (g.res && g.res_g) {
      const float *rr = g.res + (size_t)m * g.res_stride;
      float s = 0.f;
      for (int i = lane; i < g.N; i += 64) s += ldc(rr + i);
      const float rm = wave_sum(s) / (float)g.N;
      float var = 0.f;
      for (int i = lane; i < g.N; i += 64) { const float d = ldc(rr + i) - rm; var += d * d; }
      const float vs = wave_sum(var);
      if (lane == 0) { stat[2 * m] = rm; stat[2 * m + 1] = 1.0f / sqrtf(vs / (float)g.N + 1e-5f); }
    }
  }
  __syncthreads();
  // ---- work items
  for (;;) {
    if (n0 < g.N) {
      float acc0[MR], acc1[MR];
#pragma unroll
      for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
#pragma unroll
      for (int i = 0; i < WPF; ++i) {
        const int qd = lane + 64 * i;
        if (qd < nq) {
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            if (m < M) {
              const float4 xv = reinterpret_cast<const float4 *>(xs + (size_t)m * g.K)[qd];
              acc0[m] += (wa[i].x * xv.x + wa[i].y * xv.y) + (wa[i].z * xv.z + wa[i].w * xv.w);
              acc1[m] += (wb[i].x * xv.x + wb[i].y * xv.y) + (wb[i].z * xv.z + wb[i].w * xv.w);
            }
          }
        }
      }
      const bool two = n0 + 1 < g.N;
#pragma unroll
      for (int m = 0; m < MR; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
      if (lane < 2 && (lane == 0 || two)) {
        const int n = n0 + lane;
        const float b = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          if (m < M) {
            float v = (lane == 0 ? acc0[m] : acc1[m]) + b;
            if (g.res) {
              float r = ldc(g.res + (size_t)m * g.res_stride + n);
              if (g.res_g) r = (r - stat[2 * m]) * stat[2 * m + 1] * g.res_g[n] + g.res_b[n];
              v += r;
            }
            if (g.relu) v = fmaxf(v, 0.f);
            if (n < g.split) stc(g.out + (size_t)m * g.out_stride + n, v);
            else stc(g.out2 + (size_t)m * g.out2_stride + (n - g.split), v);
          }
        }
      }
    }
    item += gridDim.x;
    if (item >= n_items) break;
    n0 = item * NPB + wave * 2;
    load_w(n0);
  }
}

// Partials of one query row per (batch, head) against keys [0, Sk): item = (b, h, split).
template <int HD>
__device__ __attribute__((noinline)) void attn_phase(const float *q, const float *k, const float *v, const float *e, float *part_out, int Sk,
                           int B, int H, int d, int64_t kv_ss, int q_pos, int Cq, int Ck, int Ek, int R, float scale,
                           float *sm) {
  constexpr int G = HD / 4, RPP = 256 / G, ST = HD + 2;
  float *red = sm, *part = red + 8, *sc = part + RPP * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G, gl = tid % G;
  const int chunk = (Sk + NSPLIT - 1) / NSPLIT;
  const int n_items = B * H * NSPLIT;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int z = item % NSPLIT, h = (item / NSPLIT) % H, b = item / (NSPLIT * H);
    float *pp = part_out + (((size_t)b * H + h) * NSPLIT + z) * ST;
    const int kbeg = z * chunk;
    const int n = min(Sk - kbeg, chunk);
    if (n <= 0) {   // empty split: neutral partial
      if (tid < HD) stc(pp + tid, 0.f);
      if (tid == 0) { stc(pp + HD, -1e30f); stc(pp + HD + 1, 0.f); }
      continue;
    }
    __syncthreads();   // the previous item's readers of red / part / sc are done
    const float *qp = q + (size_t)b * d + h * HD + gl * 4;
    const float4 qq = make_float4(ldc(qp), ldc(qp + 1), ldc(qp + 2), ldc(qp + 3));
    const int evq = q_pos / Cq;
    const float *kb = k + (size_t)kbeg * kv_ss + (size_t)b * 2 * d + h * HD + gl * 4;
    const float *vb = v + (size_t)kbeg * kv_ss + (size_t)b * 2 * d + h * HD + gl * 4;
    const float *eb = e ? e + (size_t)h * R * HD + gl * 4 : nullptr;
    float lmax = -1e30f;
    for (int j0 = grp; j0 < n; j0 += 4 * RPP) {
      float part_s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * RPP;
        float acc = 0.f;
        if (j < n) {
          float4 kk = *reinterpret_cast<const float4 *>(kb + (size_t)j * kv_ss);
          if (eb) {
            int r = evq - (kbeg + j) / Ck + Ek - 1;
            r = r < 0 ? 0 : (r >= R ? R - 1 : r);
            const float4 ee = *reinterpret_cast<const float4 *>(eb + (size_t)r * HD);
            kk.x += ee.x; kk.y += ee.y; kk.z += ee.z; kk.w += ee.w;
          }
          acc = (qq.x * kk.x + qq.y * kk.y) + (qq.z * kk.z + qq.w * kk.w);
        }
        part_s[u] = acc;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float acc = part_s[u];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        const int j = j0 + u * RPP;
        if (j < n) {
          acc *= scale;
          if (gl == 0) sc[j] = acc;
          lmax = fmaxf(lmax, acc);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int j = tid; j < n; j += 256) {
      const float pj = expf(sc[j] - gmax);
      sc[j] = pj;
      lsum += pj;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float gsum = (red[4] + red[5]) + (red[6] + red[7]);
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0, o2 = o0, o3 = o0;
    for (int j0 = grp; j0 < n; j0 += 4 * RPP) {
      const int ja = j0, jb = j0 + RPP, jc = j0 + 2 * RPP, jd = j0 + 3 * RPP;
      const float pa = sc[ja], pb = jb < n ? sc[jb] : 0.f, pc = jc < n ? sc[jc] : 0.f, pd = jd < n ? sc[jd] : 0.f;
      const float4 va = *reinterpret_cast<const float4 *>(vb + (size_t)ja * kv_ss);
      const float4 vbb = *reinterpret_cast<const float4 *>(vb + (size_t)(jb < n ? jb : ja) * kv_ss);
      const float4 vc = *reinterpret_cast<const float4 *>(vb + (size_t)(jc < n ? jc : ja) * kv_ss);
      const float4 vd = *reinterpret_cast<const float4 *>(vb + (size_t)(jd < n ? jd : ja) * kv_ss);
      o0.x += pa * va.x; o0.y += pa * va.y; o0.z += pa * va.z; o0.w += pa * va.w;
      o1.x += pb * vbb.x; o1.y += pb * vbb.y; o1.z += pb * vbb.z; o1.w += pb * vbb.w;
      o2.x += pc * vc.x; o2.y += pc * vc.y; o2.z += pc * vc.z; o2.w += pc * vc.w;
      o3.x += pd * vd.x; o3.y += pd * vd.y; o3.z += pd * vd.z; o3.w += pd * vd.w;
    }
    o0.x = (o0.x + o1.x) + (o2.x + o3.x); o0.y = (o0.y + o1.y) + (o2.y + o3.y);
    o0.z = (o0.z + o1.z) + (o2.z + o3.z); o0.w = (o0.w + o1.w) + (o2.w + o3.w);
    *reinterpret_cast<float4 *>(part + grp * HD + gl * 4) = o0;
    __syncthreads();
    if (tid < HD) {
      float acc = 0.f;
      for (int gI = 0; gI < RPP; ++gI) acc += part[gI * HD + tid];
      stc(pp + tid, acc);
      if (tid == 0) { stc(pp + HD, gmax); stc(pp + HD + 1, gsum); }
    }
  }
}

template <int MR, int HD>
__global__ __launch_bounds__(256) void prior_position_kernel(const PersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const isi_prior_w &w = a.w;
  const int d = w.d_model, B = a.B, H = w.nhead, ff = w.dim_feedforward, p = a.p;
  unsigned *counter = a.bar + (p & 1), *err = a.bar + 2;
  unsigned epoch = 0;
  const unsigned G = gridDim.x;
  auto sync = [&]() { ++epoch; grid_barrier(counter, epoch * G, err); };
  const size_t cache_layer = (size_t)a.S_t * B * 2 * d, mem_layer = (size_t)a.S_src * B * 2 * d;
  const int64_t kv_ss = (int64_t)B * 2 * d;

  const float *yin = a.x_seq + (size_t)p * B * d;
  const float *ln_g = nullptr, *ln_b = nullptr;
  for (int l = 0; l < w.n_layers; ++l) {
    const isi_decoder_layer_w &L = w.layers[l];
    float *cache = a.kv_cache + l * cache_layer;
    const float *memkv = a.memory_kv + l * mem_layer;
    float *y3 = (l & 1) ? a.y3b : a.y3a;
    Gemv g;
    // q | k,v (k, v straight into the cache slot of this position)
    g = Gemv{yin, d, ln_g, ln_b, nullptr, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias, nullptr, 0, nullptr, nullptr,
             a.q, d, cache + (size_t)p * B * 2 * d, 2 * d, d, 3 * d, d, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    attn_phase<HD>(a.q, cache, cache + d, L.self_attn.rel_embeddings, a.part, p + 1, B, H, d, kv_ss, p, w.Cd, w.Cd, w.Ed,
                   L.self_attn.rel_rows, a.scale, sm);
    sync();
    // y1 = LN_in(yin) + attn Wo^T + bo
    g = Gemv{nullptr, 0, nullptr, nullptr, a.part, L.self_attn.out_proj_weight, L.self_attn.out_proj_bias, yin, d, ln_g, ln_b,
             a.y1, d, nullptr, 0, d, d, d, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    // cross-attention query from LN1(y1)
    g = Gemv{a.y1, d, L.norm1_w, L.norm1_b, nullptr, L.cross_attn.in_proj_weight, L.cross_attn.in_proj_bias, nullptr, 0,
             nullptr, nullptr, a.q, d, nullptr, 0, d, d, d, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    attn_phase<HD>(a.q, memkv, memkv + d, L.cross_attn.rel_embeddings, a.part, a.S_src, B, H, d, kv_ss, p, w.Cd, w.Ce, w.Ee,
                   L.cross_attn.rel_rows, a.scale, sm);
    sync();
    g = Gemv{nullptr, 0, nullptr, nullptr, a.part, L.cross_attn.out_proj_weight, L.cross_attn.out_proj_bias, a.y1, d,
             L.norm1_w, L.norm1_b, a.y2, d, nullptr, 0, d, d, d, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    // feed-forward on LN2(y2)
    g = Gemv{a.y2, d, L.norm2_w, L.norm2_b, nullptr, L.linear1_w, L.linear1_b, nullptr, 0, nullptr, nullptr, a.hid, ff,
             nullptr, 0, ff, ff, d, 1};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    g = Gemv{a.hid, ff, nullptr, nullptr, nullptr, L.linear2_w, L.linear2_b, a.y2, d, L.norm2_w, L.norm2_b, y3, d, nullptr,
             0, d, d, ff, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
    sync();
    yin = y3; ln_g = L.norm3_w; ln_b = L.norm3_b;
  }
  if (a.want_logits) {
    Gemv g{yin, d, ln_g, ln_b, nullptr, w.logits_w, w.logits_b, nullptr, 0, nullptr, nullptr, a.logits, w.n_class, nullptr, 0,
           w.n_class, w.n_class, d, 0};
    gemv_phase<MR, HD>(g, B, H, sm);
  }
  // the other parity's counter is idle during this launch: reset it for the next position
  if (blockIdx.x == 0 && threadIdx.x == 0) a.bar[(p + 1) & 1] = 0u;
}

template <int MR, int HD>
int launch_position(const PersistArgs &a, int grid, size_t smem, hipStream_t st) {
  auto kern = prior_position_kernel<MR, HD>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return check_launch("hipFuncSetAttribute(prior_position)");
    attr_set = true;
  }
  void *args[] = {const_cast<PersistArgs *>(&a)};
  if (hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3(grid), dim3(256), args, (unsigned)smem,
                                 st) != hipSuccess)
    return check_launch("hipLaunchCooperativeKernel(prior_position)");
  return ISI_OK;
}
}  // namespace

// Floats of scratch the persistent path needs on top of the multi-kernel layout: none (it reuses the
// same rows); the barrier words live in the 16 spare floats after the attention workspace.
bool prior_position_supported(const isi_prior_w *w, int B) {
  if (!w || B < 1 || B > 8) return false;
  const int hd = w->d_model / w->nhead;
  if (hd != 16 && hd != 32 && hd != 64) return false;
  if (w->d_model % 4 || w->dim_feedforward % 4 || w->d_model > 2048 || w->dim_feedforward > 2048) return false;
  return true;
}

int prior_position_run(const isi_prior_w *w, const isi_prior_state *s, float *q, float *y1, float *y2, float *y3a,
                       float *y3b, float *hid, float *logits, float *part, unsigned *bar, int p, int want_logits,
                       hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return check_launch("hipGetDeviceProperties");
    n_cu = prop.multiProcessorCount;
  }
  PersistArgs a;
  a.w = *w;
  a.x_seq = s->x_seq; a.kv_cache = s->kv_cache; a.memory_kv = s->memory_kv;
  a.q = q; a.y1 = y1; a.y2 = y2; a.y3a = y3a; a.y3b = y3b; a.hid = hid; a.logits = logits; a.part = part;
  a.bar = bar;
  a.S_t = s->S_t; a.S_src = s->S_src; a.B = s->B; a.p = p; a.want_logits = want_logits;
  const int hd = w->d_model / w->nhead;
  a.scale = 1.0f / sqrtf((float)hd);
  const int B = s->B;
  const int kmax = w->d_model > w->dim_feedforward ? w->d_model : w->dim_feedforward;
  const int smax = s->S_t > s->S_src ? s->S_t : s->S_src;
  const size_t smem_gemv = ((size_t)B * kmax + 2 * B) * sizeof(float);
  const size_t smem_attn = (size_t)(8 + 256 * 4 + (smax + NSPLIT - 1) / NSPLIT) * sizeof(float);
  const size_t smem = smem_gemv > smem_attn ? smem_gemv : smem_attn;
  if (smem > 160 * 1024) return unsupported("prior_position: rows do not fit in LDS");
#define ISI_PP(MR)                                                          \
  switch (hd) {                                                             \
    case 16: return launch_position<MR, 16>(a, n_cu, smem, st);             \
    case 32: return launch_position<MR, 32>(a, n_cu, smem, st);             \
    default: return launch_position<MR, 64>(a, n_cu, smem, st);             \
  }
  if (B <= 1) { ISI_PP(1) }
  else if (B <= 2) { ISI_PP(2) }
  else if (B <= 4) { ISI_PP(4) }
  else { ISI_PP(8) }
#undef ISI_PP
}

}  // namespace isi
