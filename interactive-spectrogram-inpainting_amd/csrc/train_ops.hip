// Element-wise / reduction operators of the VQ-VAE training step (gfx950, fp32).
// They replace the autograd kernels behind `loss.backward()` and the in-forward
// EMA codebook update of the reference (train_vqvae.py:174-189,
// vqvae/bottleneck.py:79-95).
#include <algorithm>

#include "isi_common.h"
#include "isi_internal.h"

namespace isi {

// dy *= (y > 0): ReLU backward through an output that was rectified in the producer's epilogue.
__global__ void relu_bwd_kernel(float *__restrict__ dy, const float *__restrict__ y, int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = first; i < n4; i += stride) {
    float4 g = reinterpret_cast<float4 *>(dy)[i];
    const float4 v = reinterpret_cast<const float4 *>(y)[i];
    g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f;
    g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
    reinterpret_cast<float4 *>(dy)[i] = g;
  }
  const int64_t t = 4 * n4 + first;  // tail of a length that is not a multiple of 4
  if (t < n) dy[t] = y[t] > 0.f ? dy[t] : 0.f;
}

// a += alpha * b
__global__ void axpy_kernel(float *__restrict__ a, const float *__restrict__ b, float alpha, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 x = reinterpret_cast<float4 *>(a)[i];
    const float4 y = reinterpret_cast<const float4 *>(b)[i];
    x.x += alpha * y.x; x.y += alpha * y.y; x.z += alpha * y.z; x.w += alpha * y.w;
    reinterpret_cast<float4 *>(a)[i] = x;
  }
}

// Quantiser backward: straight-through + commitment term.
//   quantize = z + (q - z).detach()  ->  d/dz = dq ;  diff = mean((q.detach() - z)^2) -> d/dz = 2 (z - q) g_diff / (N D)
//   q_st = z + (q - z) is what the forward stored, so (z - q) = z - (q_st) up to rounding: use z and q_st.
__global__ void vq_bwd_kernel(float *__restrict__ dz, const float *__restrict__ dq, const float *__restrict__ z,
                              const float *__restrict__ q_st, const float *__restrict__ g_diff, float inv_numel,
                              int64_t n4) {
  const float coef = 2.f * g_diff[0] * inv_numel;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4 *>(dq)[i];
    const float4 a = reinterpret_cast<const float4 *>(z)[i];
    const float4 b = reinterpret_cast<const float4 *>(q_st)[i];
    float4 o;
    o.x = g.x + coef * (a.x - b.x); o.y = g.y + coef * (a.y - b.y);
    o.z = g.z + coef * (a.z - b.z); o.w = g.w + coef * (a.w - b.w);
    reinterpret_cast<float4 *>(dz)[i] = o;
  }
}

// Column sums of a dense [M, C] matrix (bias gradients): per-block partials [nblk][C].
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, float *__restrict__ partial,
                                                             int64_t M, int C, int64_t x_stride, int rows_per_block) {
  __shared__ float red[4][64];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + e;
    float s = 0.f;
    if (c < C)
      for (int64_t r = r0 + g; r < r1; r += 4) s += x[r * x_stride + c];
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && c < C) partial[(size_t)blockIdx.x * C + c] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    __syncthreads();
  }
}

// EMA update of the codebook buffers (bottleneck.py:79-92), layout [D][K] like the reference:
//   cluster_size = g cs + (1-g) counts ; embed_avg = g ea + (1-g) embed_sum^T
//   n = sum(cluster_size) ; cs' = (cs + eps) / (n + K eps) n ; embed = embed_avg / cs'
__global__ __launch_bounds__(1024) void vq_ema_update_kernel(float *__restrict__ embed, float *__restrict__ cluster_size,
                                                             float *__restrict__ embed_avg,
                                                             const float *__restrict__ counts,
                                                             const float *__restrict__ embed_sum_dk, int D, int K,
                                                             float decay, float eps) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int k = tid; k < K; k += blockDim.x) {
    const float cs = cluster_size[k] * decay + (1.f - decay) * counts[k];
    cluster_size[k] = cs;
    s += cs;
  }
  red[tid] = s;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float n = red[0];
  for (int i = tid; i < D * K; i += blockDim.x) {
    const int k = i % K;
    const float ea = embed_avg[i] * decay + (1.f - decay) * embed_sum_dk[i];
    embed_avg[i] = ea;
    const float csn = (cluster_size[k] + eps) / (n + K * eps) * n;
    embed[i] = ea / csn;
  }
}

static unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 8192); }

int relu_bwd_f32(float *dy, const float *y, int64_t n, hipStream_t st) {
  if (!dy || !y || n < 0) return invalid("relu_bwd: bad argument");
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y)) & 15)
    return invalid("relu_bwd: pointers must be 16-byte aligned");
  if (n == 0) return ISI_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, dy, y, n / 4, n);
  return check_launch("relu_bwd_f32");
}

int axpy_f32(float *a, const float *b, float alpha, int64_t n, hipStream_t st) {
  if (!a || !b || n < 0 || (n & 3)) return invalid("axpy: bad argument (n % 4 == 0 required)");
  if (n == 0) return ISI_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, a, b, alpha, n / 4);
  return check_launch("axpy_f32");
}

int vq_bwd_f32(float *dz, const float *dq, const float *z, const float *q_st, const float *g_diff, int64_t n,
               hipStream_t st) {
  if (!dz || !dq || !z || !q_st || !g_diff || n <= 0 || (n & 3)) return invalid("vq_bwd: bad argument");
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, dz, dq, z, q_st, g_diff,
                     1.0f / (float)n, n / 4);
  return check_launch("vq_bwd_f32");
}

int colsum_num_partials(int64_t M) { return (int)std::min<int64_t>((M + 255) / 256, 256); }

// out[C] = column sums of x [M, C] (row stride x_stride); workspace: colsum_num_partials(M) * C floats
int colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M, int C, hipStream_t st);

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *__restrict__ partial,
                                                          float *__restrict__ out, int C, int nblk) {
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + e;
  float s = 0.f;
  if (c < C)
    for (int b = g; b < nblk; b += 4) s += partial[(size_t)b * C + c];
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && c < C) out[c] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

int colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M, int C, hipStream_t st) {
  if (!x || !out || !workspace || M <= 0 || C <= 0) return invalid("colsum: bad argument");
  const int nblk = colsum_num_partials(M);
  const int rows = (int)((M + nblk - 1) / nblk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, x, workspace, M, C, x_stride, rows);
  int rc = check_launch("colsum_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 63) / 64), dim3(256), 0, st, workspace, out, C, nblk);
  return check_launch("colsum_reduce");
}

int vq_ema_update_f32(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                      const float *embed_sum_dk, int D, int K, float decay, float eps, hipStream_t st) {
  if (!embed || !cluster_size || !embed_avg || !counts || !embed_sum_dk || D <= 0 || K <= 0)
    return invalid("vq_ema_update: bad argument");
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(1024), 0, st, embed, cluster_size, embed_avg, counts,
                     embed_sum_dk, D, K, decay, eps);
  return check_launch("vq_ema_update_f32");
}

}  // namespace isi
