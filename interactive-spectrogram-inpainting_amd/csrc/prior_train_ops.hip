// Training kernels of the transformer prior that are not GEMMs (gfx950):
//   layernorm_bwd_f32        backward of layernorm_f32 (transformer_ops.hip)
//   label_smoothing_loss_f32 the criterion of the prior, forward and gradient in one pass
//                            (reference utils/losses/prediction.py:5-20 behind
//                            train_autoregressive_model.py:254-257)
// All HBM-bound: one pass over the activations, 16-byte accesses, a wave per row.
#include <algorithm>
#include <cmath>

#include "isi_common.h"
#include "isi_internal.h"

namespace isi {

namespace {
constexpr int MAXV = 8;  // float4 per lane and row: D <= 2048
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
}  // namespace

// z = x + res ; xhat = (z - mean) * rstd ; y = xhat * gamma + beta
// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// partial[blk][0][c] = sum over the block's rows of dy * xhat ; partial[blk][1][c] = sum of dy
// With a fused dropout (drop_thresh != 0: z = drop(x) + res, transformer_ops.hip) the mask is formed again from (seed, index)
// and the gradient of x -- dz where the element was kept, times the inverse keep probability -- goes to `dx`.
// MAXV: float4 per lane and row the registers are laid out for (D <= 256 MAXV).  With the one layout for D <= 2048 a row
// of the prior's D = 512 kept 160 VGPRs for 40 live ones, i.e. two or three waves per SIMD for a kernel that is a chain of
// load -> reduce -> reduce -> reduce -> store per row: 37 us where its 84 MB take 17 at the HBM's rate.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ dy, float *__restrict__ dz,
                                                            float *__restrict__ partial, int M, int D, float eps,
                                                            int rows_per_block, float *__restrict__ dx,
                                                            unsigned drop_thresh, float drop_scale, uint64_t seed0,
                                                            const uint64_t *seed_base) {
  const uint64_t seed = seed0 + (drop_thresh && seed_base ? *seed_base : 0);
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  float4 ag[MAXV], ab[MAXV], gm[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int qd = lane + 64 * i;
    gm[i] = qd < nq ? reinterpret_cast<const float4 *>(gamma)[qd] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(M, row0 + rows_per_block);
  for (int row = row0 + wave; row < row1; row += 4) {
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * D);
    const float4 *rr = res ? reinterpret_cast<const float4 *>(res + (size_t)row * D) : nullptr;
    const float4 *gr = reinterpret_cast<const float4 *>(dy + (size_t)row * D);
    float4 v[MAXV], g[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int qd = lane + 64 * i;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f), u = t;
      if (qd < nq) {
        t = xr[qd];
        if (drop_thresh) {
          const unsigned e0 = (unsigned)row * (unsigned)D + 4u * qd;
          t.x = dropout_keep(seed, e0, drop_thresh) ? t.x * drop_scale : 0.f;
          t.y = dropout_keep(seed, e0 + 1, drop_thresh) ? t.y * drop_scale : 0.f;
          t.z = dropout_keep(seed, e0 + 2, drop_thresh) ? t.z * drop_scale : 0.f;
          t.w = dropout_keep(seed, e0 + 3, drop_thresh) ? t.w * drop_scale : 0.f;
        }
        if (rr) { const float4 w = rr[qd]; t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w; }
        u = gr[qd];
        sum += (t.x + t.y) + (t.z + t.w);
      }
      v[i] = t; g[i] = u;
    }
    const float mean = wave_sum(sum) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int qd = lane + 64 * i;
      if (qd < nq) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        var += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;  // sum g, sum g * xhat
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int qd = lane + 64 * i;
      if (qd < nq) {
        float4 xh;
        xh.x = (v[i].x - mean) * rstd; xh.y = (v[i].y - mean) * rstd;
        xh.z = (v[i].z - mean) * rstd; xh.w = (v[i].w - mean) * rstd;
        ag[i].x += g[i].x * xh.x; ag[i].y += g[i].y * xh.y; ag[i].z += g[i].z * xh.z; ag[i].w += g[i].w * xh.w;
        ab[i].x += g[i].x; ab[i].y += g[i].y; ab[i].z += g[i].z; ab[i].w += g[i].w;
        g[i].x *= gm[i].x; g[i].y *= gm[i].y; g[i].z *= gm[i].z; g[i].w *= gm[i].w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh.x + g[i].y * xh.y) + (g[i].z * xh.z + g[i].w * xh.w);
        v[i] = xh;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    float4 *orow = reinterpret_cast<float4 *>(dz + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int qd = lane + 64 * i;
      if (qd < nq) {
        float4 o;
        o.x = rstd * (g[i].x - s1 - v[i].x * s2); o.y = rstd * (g[i].y - s1 - v[i].y * s2);
        o.z = rstd * (g[i].z - s1 - v[i].z * s2); o.w = rstd * (g[i].w - s1 - v[i].w * s2);
        orow[qd] = o;
        if (drop_thresh) {
          const unsigned e0 = (unsigned)row * (unsigned)D + 4u * qd;
          float4 m;
          m.x = dropout_keep(seed, e0, drop_thresh) ? o.x * drop_scale : 0.f;
          m.y = dropout_keep(seed, e0 + 1, drop_thresh) ? o.y * drop_scale : 0.f;
          m.z = dropout_keep(seed, e0 + 2, drop_thresh) ? o.z * drop_scale : 0.f;
          m.w = dropout_keep(seed, e0 + 3, drop_thresh) ? o.w * drop_scale : 0.f;
          reinterpret_cast<float4 *>(dx + (size_t)row * D)[qd] = m;
        }
      }
    }
  }
  // combine the four waves in a fixed order
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      reinterpret_cast<float4 *>(red + (size_t)(wave * 2 + 0) * D)[qd] = ag[i];
      reinterpret_cast<float4 *>(red + (size_t)(wave * 2 + 1) * D)[qd] = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int which = c / D, col = c - which * D;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[(size_t)(w * 2 + which) * D + col];
    partial[((size_t)blockIdx.x * 2 + which) * D + col] = s;
  }
}

// dgamma | dbeta [2][D] = sum over the blocks' partials [nblk][2][D]: 64 columns per workgroup, its sixteen waves
// take interleaved blocks (fixed order -> deterministic; with four waves a thread's chain of 256 dependent adds made
// this launch as long as the backward kernel itself: 22 -> ~7 us, 36 launches per training step)
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_kernel(const float *__restrict__ partial,
                                                                    float *__restrict__ dgamma,
                                                                    float *__restrict__ dbeta, int nblk, int D) {
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    const int which = c / D, col = c - which * D;
    const float *pp = partial + (size_t)which * D + col;
    int b = g;
    for (; b + 48 < nblk; b += 64) {
      s0 += pp[(size_t)b * 2 * D]; s1 += pp[(size_t)(b + 16) * 2 * D];
      s2 += pp[(size_t)(b + 32) * 2 * D]; s3 += pp[(size_t)(b + 48) * 2 * D];
    }
    for (; b < nblk; b += 16) s0 += pp[(size_t)b * 2 * D];
  }
  red[g][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < 2 * D) {
    const int which = c / D, col = c - which * D;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][e];
    (which ? dbeta : dgamma)[col] = t;
  }
}

// 8 rows per workgroup (2 per wave): enough waves per SIMD to hide a row's dependent load -> reduce -> store chain
static int ln_bwd_blocks(int64_t M) { return (int)std::min<int64_t>(1024, (M + 7) / 8); }

size_t layernorm_bwd_workspace_floats(int64_t M, int D) { return M > 0 ? (size_t)ln_bwd_blocks(M) * 2 * D : 0; }

int layernorm_bwd_f32(const float *x, const float *res, const float *gamma, const float *dy, float *dz,
                      float *dgamma, float *dbeta, float *workspace, int64_t M, int D, float eps,
                      hipStream_t stream, float *dx, float drop_p, uint64_t drop_seed) {
  unsigned thresh = 0; float scale = 1.f;
  if (drop_p != 0.f) {
    if (!(drop_p > 0.f && drop_p < 1.f) || M * D > ((int64_t)1 << 32) || !dx || (reinterpret_cast<uintptr_t>(dx) & 15))
      return invalid("layernorm_bwd: dropout needs 0 <= p < 1, fewer than 2^32 elements and a 16-byte aligned dx");
    const double t = (double)drop_p * 4294967296.0;
    thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (t < 1.0 ? 1u : (unsigned)t);
    scale = 1.f / (1.f - drop_p);
  }
  if (!x || !gamma || !dy || !dz || !dgamma || !dbeta || !workspace || M <= 0 || M > INT32_MAX || D <= 0)
    return invalid("layernorm_bwd: bad argument");
  if ((D & 3) || D > 2048) return unsupported("layernorm_bwd: need D % 4 == 0 and D <= 2048");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(gamma) |
       reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15)
    return invalid("layernorm_bwd: pointers must be 16-byte aligned");
  const int nblk = ln_bwd_blocks(M);
  const int rpb = (int)((M + nblk - 1) / nblk);
#define ISI_LN_BWD(V_) hipLaunchKernelGGL(layernorm_bwd_kernel<V_>, dim3(nblk), dim3(256), (size_t)8 * D * sizeof(float), stream, x, res, \
                     gamma, dy, dz, workspace, (int)M, D, eps, rpb, dx, thresh, scale, drop_seed, dropout_seed_base())
  if (D <= 512) ISI_LN_BWD(2);
  else if (D <= 1024) ISI_LN_BWD(4);
  else ISI_LN_BWD(8);
#undef ISI_LN_BWD
  int rc = check_launch("layernorm_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, stream, workspace, dgamma,
                     dbeta, nblk, D);
  return check_launch("layernorm_bwd_reduce");
}

// Label smoothing (prediction.py:5-20): true_dist = smoothing / (num_classes - 1) everywhere, 1 - smoothing at the
// target; row loss = sum_k -true_dist[k] * log_softmax(logits)[k].  One wave per row; writes the row loss
// and d(mean loss)/d(logits) = (softmax - true_dist) * grad_scale  (grad_scale = upstream / rows).
__global__ __launch_bounds__(256) void label_smoothing_kernel(const float *__restrict__ logits,
                                                              const int64_t *__restrict__ target,
                                                              float *__restrict__ row_loss, float *__restrict__ dlogits,
                                                              int M, int K, int num_classes, float smoothing, float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float *lr = logits + (size_t)row * K;
  float mx = -INFINITY;
  for (int c = lane; c < K; c += 64) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  float se = 0.f, sl = 0.f;
  for (int c = lane; c < K; c += 64) {
    const float v = lr[c] - mx;
    se += expf(v);
    sl += v;
  }
  se = wave_sum(se);
  sl = wave_sum(sl);
  const float lse = logf(se);
  const int t = (int)target[row];
  const float off = smoothing / (float)(num_classes - 1), on = 1.f - smoothing;
  // sum_k -td[k] (v_k - lse) = -off * (sl - K lse) - (on - off) * (v_t - lse)
  if (lane == 0) {
    const float vt = lr[t] - mx;
    row_loss[row] = -off * (sl - (float)K * lse) - (on - off) * (vt - lse);
  }
  if (dlogits) {
    float *dr = dlogits + (size_t)row * K;
    const float inv = 1.f / se;
    for (int c = lane; c < K; c += 64) {
      const float pr = expf(lr[c] - mx) * inv;
      dr[c] = (pr - (c == t ? on : off)) * grad_scale;
    }
  }
}

int label_smoothing_loss_f32(const float *logits, const int64_t *target, float *row_loss, float *dlogits, int64_t M,
                             int K, int num_classes, float smoothing, float grad_scale, hipStream_t stream) {
  if (!logits || !target || !row_loss || M <= 0 || M > INT32_MAX || K <= 1 || num_classes <= 1)
    return invalid("label_smoothing_loss: bad argument");
  hipLaunchKernelGGL(label_smoothing_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, logits, target,
                     row_loss, dlogits, (int)M, K, num_classes, smoothing, grad_scale);
  return check_launch("label_smoothing_loss");
}

}  // namespace isi
