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
    if (c < C) {
      int64_t r = r0 + g;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (; r + 12 < r1; r += 16) {      // four independent loads in flight (a channel slice's rows are 2 x C floats apart)
        s += x[r * x_stride + c]; s1 += x[(r + 4) * x_stride + c];
        s2 += x[(r + 8) * x_stride + c]; s3 += x[(r + 12) * x_stride + c];
      }
      for (; r < r1; r += 4) s += x[r * x_stride + c];
      s = (s + s1) + (s2 + s3);
    }
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

// out[m][c] = gate(y[m][c]) * (a[m][c] + b[m][c]): the sum of two gradient contributions with the ReLU mask of the tensor
// they belong to, in one pass; `a` may be a channel slice of a wider channels-last tensor (row stride lda >= C): replaces a
// slice copy + axpy + relu_bwd (three passes over the bottom encoder's 134 MB output gradient).  y = nullptr: no mask.
__global__ void add_gate_rows_kernel(float *__restrict__ out, const float *__restrict__ a, int64_t lda,
                                     const float *__restrict__ b, const float *__restrict__ y, int64_t M, int C4) {
  const int64_t n4 = M * C4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t m = i / C4;
    const int c4 = (int)(i - m * C4);
    const float4 u = *reinterpret_cast<const float4 *>(a + m * lda + 4 * c4);
    float4 g = make_float4(u.x, u.y, u.z, u.w);
    if (b) {
      const float4 v = reinterpret_cast<const float4 *>(b)[i];
      g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    if (y) {
      const float4 r = reinterpret_cast<const float4 *>(y)[i];
      g.x = r.x > 0.f ? g.x : 0.f; g.y = r.y > 0.f ? g.y : 0.f; g.z = r.z > 0.f ? g.z : 0.f; g.w = r.w > 0.f ? g.w : 0.f;
    }
    reinterpret_cast<float4 *>(out)[i] = g;
  }
}

// vq_bwd_kernel with dq read through a row stride (a channel slice of the decoder's input gradient: no dense copy first)
__global__ void vq_bwd_rows_kernel(float *__restrict__ dz, const float *__restrict__ dq, int64_t ldq,
                                   const float *__restrict__ z, const float *__restrict__ q_st,
                                   const float *__restrict__ g_diff, float inv_numel, int64_t M, int D4) {
  const float coef = 2.f * g_diff[0] * inv_numel;
  const int64_t n4 = M * D4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t m = i / D4;
    const int c4 = (int)(i - m * D4);
    const float4 g = *reinterpret_cast<const float4 *>(dq + m * ldq + 4 * c4);
    const float4 a = reinterpret_cast<const float4 *>(z)[i];
    const float4 b = reinterpret_cast<const float4 *>(q_st)[i];
    float4 o;
    o.x = g.x + coef * (a.x - b.x); o.y = g.y + coef * (a.y - b.y);
    o.z = g.z + coef * (a.z - b.z); o.w = g.w + coef * (a.w - b.w);
    reinterpret_cast<float4 *>(dz)[i] = o;
  }
}

// [B, C <= 4, H, W] view (any strides) -> dense channels-last [B, H, W, 4], channels >= C zero: the 2-channel spectrogram
// side of the first / last layer as a 4-channel operand of the vectorised weight-gradient kernel, in one pass (was: a
// zero fill of the 4-channel tensor + a strided copy into it)
__global__ void pad_channels4_kernel(const float *__restrict__ x, float4 *__restrict__ out, int64_t npix, int C, int HW,
                                     int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t b = i / HW;
    const int rem = (int)(i - b * HW), y = rem / W, xx = rem - y * W;
    const float *p = x + b * sn + y * sh + xx * sw;
    float4 v = make_float4(p[0], 0.f, 0.f, 0.f);
    if (C > 1) v.y = p[sc];
    if (C > 2) v.z = p[2 * sc];
    if (C > 3) v.w = p[3 * sc];
    out[i] = v;
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

int pad_channels4_f32(const float *x, float *out, int B, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                      hipStream_t st) {
  if (!x || !out || B <= 0 || C < 1 || C > 4 || H <= 0 || W <= 0) return invalid("pad_channels4: bad argument");
  if (reinterpret_cast<uintptr_t>(out) & 15) return invalid("pad_channels4: output must be 16-byte aligned");
  const int64_t npix = (int64_t)B * H * W;
  hipLaunchKernelGGL(pad_channels4_kernel, dim3(grid_for(npix)), dim3(256), 0, st, x, reinterpret_cast<float4 *>(out), npix, C,
                     H * W, W, sn, sc, sh, sw);
  return check_launch("pad_channels4_f32");
}

int add_gate_rows_f32(float *out, const float *a, int64_t lda, const float *b, const float *y, int64_t M, int C, hipStream_t st) {
  if (!out || !a || M <= 0 || C <= 0 || (C & 3) || (lda & 3) || lda < C) return invalid("add_gate_rows: bad argument (C, lda multiples of 4)");
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15)
    return invalid("add_gate_rows: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(add_gate_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, st, out, a, lda, b, y, M, C / 4);
  return check_launch("add_gate_rows_f32");
}

int vq_bwd_rows_f32(float *dz, const float *dq, int64_t ldq, const float *z, const float *q_st, const float *g_diff, int64_t M,
                    int D, hipStream_t st) {
  if (!dz || !dq || !z || !q_st || !g_diff || M <= 0 || D <= 0 || (D & 3) || (ldq & 3) || ldq < D) return invalid("vq_bwd_rows: bad argument");
  if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(q_st)) & 15)
    return invalid("vq_bwd_rows: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(vq_bwd_rows_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, st, dz, dq, ldq, z, q_st, g_diff,
                     1.0f / (float)(M * D), M, D / 4);
  return check_launch("vq_bwd_rows_f32");
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

// (round 5: 16 columns x 16 row groups per block instead of 64 x 4 -- with C = 64 the old form was ONE block whose threads
// walked 64 partial rows each, 16 us of pure latency per bias gradient)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *__restrict__ partial,
                                                          float *__restrict__ out, int C, int nblk) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + e;
  float s = 0.f;
  if (c < C)
    for (int b = g; b < nblk; b += 16) s += partial[(size_t)b * C + c];
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][e];
    out[c] = t;
  }
}

// Dense [M, C] with C a power of two <= 1024: the matrix is streamed as one flat float4 array; a thread's
// stride (256 float4 = 1024 floats) is a multiple of C, so its four lanes always see the same four columns and
// every lane of the wave loads (C = 2 used 2 lanes of 64 in colsum_partial_kernel: 500 -> ~60 us on the
// last decoder layer's bias gradient).
__global__ __launch_bounds__(256) void colsum_flat_kernel(const float4 *__restrict__ x, float *__restrict__ partial,
                                                          int64_t n4, int C, int64_t f4_per_block) {
  __shared__ float red[256][4];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * f4_per_block, i1 = i0 + f4_per_block < n4 ? i0 + f4_per_block : n4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, d = a;
  int64_t i = i0 + tid;
  for (; i + 768 < i1; i += 1024) {
    const float4 v0 = x[i], v1 = x[i + 256], v2 = x[i + 512], v3 = x[i + 768];
    a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
    c.x += v2.x; c.y += v2.y; c.z += v2.z; c.w += v2.w;
    d.x += v3.x; d.y += v3.y; d.z += v3.z; d.w += v3.w;
  }
  for (; i < i1; i += 256) {
    const float4 v0 = x[i];
    a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
  }
  red[tid][0] = (a.x + b.x) + (c.x + d.x);
  red[tid][1] = (a.y + b.y) + (c.y + d.y);
  red[tid][2] = (a.z + b.z) + (c.z + d.z);
  red[tid][3] = (a.w + b.w) + (c.w + d.w);
  __syncthreads();
  for (int col = tid; col < C; col += 256) {
    float s = 0.f;
    if (C >= 4) {
      for (int t = col >> 2; t < 256; t += C >> 2) s += red[t][col & 3];
    } else {
      for (int t = 0; t < 256; ++t)
        for (int e = col; e < 4; e += C) s += red[t][e];
    }
    partial[(size_t)blockIdx.x * C + col] = s;
  }
}

int colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M, int C, hipStream_t st) {
  if (!x || !out || !workspace || M <= 0 || C <= 0) return invalid("colsum: bad argument");
  const int nblk = colsum_num_partials(M);
  if (x_stride == C && C <= 1024 && (C & (C - 1)) == 0 && ((M * C) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int64_t n4 = M * C / 4;
    const int64_t per = ((n4 + nblk - 1) / nblk + 255) / 256 * 256;   // block starts stay multiples of 1024 floats
    hipLaunchKernelGGL(colsum_flat_kernel, dim3(nblk), dim3(256), 0, st, reinterpret_cast<const float4 *>(x), workspace,
                       n4, C, per);
    int rc = check_launch("colsum_flat");
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, out, C, nblk);
    return check_launch("colsum_reduce");
  }
  const int rows = (int)((M + nblk - 1) / nblk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, x, workspace, M, C, x_stride, rows);
  int rc = check_launch("colsum_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, out, C, nblk);
  return check_launch("colsum_reduce");
}

// ---- mean squared error (train_vqvae.py:203 `nn.MSELoss()`; reference train_vqvae.py:168-176): mean((a - b)^2) and its
// gradient 2 (a - b) g / n.  Two launches forward (fixed-order partial sums per workgroup, then one workgroup), one backward --
// torch's path is an element-wise kernel, a memset of the reduction's semaphores, two reduction launches, and two element-wise
// kernels backward; the memset is what this replaces first of all: a hipMemsetAsync node inside a replayed HIP graph is not
// reliably ordered with its neighbours on ROCm 7.2.
constexpr int kMsePartials = 1024;
__global__ __launch_bounds__(256) void mse_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          float *__restrict__ partial, int64_t n) {
  __shared__ float red[4];
  const int64_t n4 = n >> 2;
  const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = a4[i], y = b4[i];
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // the last one to three elements
    const float d = a[4 * n4 + threadIdx.x] - b[4 * n4 + threadIdx.x];
    s += d * d;
  }
  s = wave64_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mse_finish_kernel(const float *__restrict__ partial, int np, float inv_n, float *__restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  s = wave64_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      const float *__restrict__ g, float two_over_n, float *__restrict__ da,
                                                      float *__restrict__ db, int64_t n) {
  const float c = two_over_n * g[0];
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
    const float4 d = make_float4((x.x - y.x) * c, (x.y - y.y) * c, (x.z - y.z) * c, (x.w - y.w) * c);
    if (da) reinterpret_cast<float4 *>(da)[i] = d;
    if (db) reinterpret_cast<float4 *>(db)[i] = make_float4(-d.x, -d.y, -d.z, -d.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = 4 * n4 + threadIdx.x;
    const float d = (a[i] - b[i]) * c;
    if (da) da[i] = d;
    if (db) db[i] = -d;
  }
}
int mse_loss_num_partials(int64_t n) {
  const int64_t nb = (n / 4 + 255) / 256;
  return (int)(nb < 1 ? 1 : nb > kMsePartials ? kMsePartials : nb);
}
int mse_loss_f32(const float *a, const float *b, int64_t n, float *workspace, float *out, hipStream_t st) {
  if (!a || !b || !workspace || !out || n <= 0) return invalid("mse_loss: bad argument");
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return invalid("mse_loss: operands must be 16-byte aligned");
  const int np = mse_loss_num_partials(n);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(np), dim3(256), 0, st, a, b, workspace, n);
  int rc = check_launch("mse_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, st, workspace, np, 1.0f / (float)n, out);
  return check_launch("mse_finish");
}
int mse_loss_bwd_f32(const float *a, const float *b, const float *g, int64_t n, float *da, float *db, hipStream_t st) {
  if (!a || !b || !g || (!da && !db) || n <= 0) return invalid("mse_loss_bwd: bad argument");
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(db)) & 15)
    return invalid("mse_loss_bwd: tensors must be 16-byte aligned");
  const int64_t nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)(nb < 1 ? 1 : nb > 8192 ? 8192 : nb)), dim3(256), 0, st, a, b, g,
                     2.0f / (float)n, da, db, n);
  return check_launch("mse_loss_bwd");
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
