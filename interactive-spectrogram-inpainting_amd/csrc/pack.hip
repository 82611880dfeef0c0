// Weight layout preparation kernels (run once per weight version, not on the hot path).
//   conv    [Cout,Cin,KH,KW]  -> [Cout][Kpad], k = (kh*KW + kw)*Cin + ci
//   convT   [Cin,Cout,4,4]    -> [phase][Cout][Kpad], k = (ty*2 + tx)*Cin + ci,
//                                torch tap (ky,kx) = (3 - py - 2 ty, 3 - px - 2 tx)
//   embed   [D,K]             -> [K][D] and e2[k] = sum_d embed[d,k]^2
// Reference weight layouts: torch.nn.Conv2d / ConvTranspose2d as instantiated at
// vqvae/encoder_decoder.py:95-112,138,199-215 and the `embed` buffer of
// vqvae/bottleneck.py:47-51.
#include "isi_common.h"
#include "isi_internal.h"
#include "split_bf16.h"
#include "split_f16.h"

namespace isi {

__global__ void pack_conv_kernel(const float *__restrict__ w, float *__restrict__ out, int Cout,
                                 int Cin, int KH, int KW, int Kpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Cout * Kpad) return;
  const int co = (int)(i / Kpad);
  const int k = (int)(i - (int64_t)co * Kpad);
  float v = 0.f;
  if (k < KH * KW * Cin) {
    const int tap = k / Cin, ci = k - tap * Cin;
    const int kh = tap / KW, kw = tap - kh * KW;
    v = w[(((int64_t)co * Cin + ci) * KH + kh) * KW + kw];
  }
  out[i] = v;
}

// pack_conv_kernel and split_weight_f16_kernel in one launch: a thread forms eight consecutive elements of a packed row
// (Kpad is a multiple of 32: a group never straddles rows), writes them and their split-f16 pair {hi[8] | lo[8]} behind
// the packed weight (a training step re-packs every convolution weight: 29 launches less per step)
__global__ void pack_conv_w16_kernel(const float *__restrict__ w, float *__restrict__ out, int Cout,
                                     int Cin, int KH, int KW, int Kpad) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * Kpad;
  if (gi * 8 >= total) return;
  const int co = (int)((gi * 8) / Kpad);
  const int kb = (int)(gi * 8 - (int64_t)co * Kpad);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kb + j;
    v[j] = 0.f;
    if (k < KH * KW * Cin) {
      const int tap = k / Cin, ci = k - tap * Cin;
      const int kh = tap / KW, kw = tap - kh * KW;
      v[j] = w[(((int64_t)co * Cin + ci) * KH + kh) * KW + kw];
    }
  }
  const float4 a = make_float4(v[0], v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
  reinterpret_cast<float4 *>(out)[2 * gi] = a;
  reinterpret_cast<float4 *>(out)[2 * gi + 1] = b;
  uint4 hi, lo;
  f16s::weight8_encode(a, b, hi, lo);
  uint4 *pairs = reinterpret_cast<uint4 *>(out + total);
  pairs[2 * gi] = hi;
  pairs[2 * gi + 1] = lo;
}

// Packed weight of the INPUT-GRADIENT convolution of a stride-1 Conv2d with weight w [Cout][Cin][KH][KW]: the
// convolution dY -> dX has Cin output channels, Cout input channels and the 180-degree rotated window:
// out[ci][(kh, kw, co)] = w[co][ci][KH-1-kh][KW-1-kw]  (one launch instead of flip + transpose + contiguous + pack).
__global__ void pack_conv_dgrad_kernel(const float *__restrict__ w, float *__restrict__ out, int Cout,
                                       int Cin, int KH, int KW, int Kpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Cin * Kpad) return;
  const int ci = (int)(i / Kpad);
  const int k = (int)(i - (int64_t)ci * Kpad);
  float v = 0.f;
  if (k < KH * KW * Cout) {
    const int tap = k / Cout, co = k - tap * Cout;
    const int kh = tap / KW, kw = tap - kh * KW;
    v = w[(((int64_t)co * Cin + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
  }
  out[i] = v;
}

__global__ void pack_convT_kernel(const float *__restrict__ w, float *__restrict__ out, int Cin,
                                  int Cout, int Kpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)4 * Cout * Kpad) return;
  const int ph = (int)(i / ((int64_t)Cout * Kpad));
  const int64_t rem = i - (int64_t)ph * Cout * Kpad;
  const int co = (int)(rem / Kpad);
  const int k = (int)(rem - (int64_t)co * Kpad);
  const int py = ph >> 1, px = ph & 1;
  float v = 0.f;
  if (k < 4 * Cin) {
    const int tap = k / Cin, ci = k - tap * Cin;
    const int ty = tap >> 1, tx = tap & 1;
    const int ky = 3 - py - 2 * ty, kx = 3 - px - 2 * tx;
    v = w[(((int64_t)ci * Cout + co) * 4 + ky) * 4 + kx];
  }
  out[i] = v;
}

__global__ void pack_codebook_kernel(const float *__restrict__ embed, float *__restrict__ codes,
                                     float *__restrict__ e2, int D, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int d0 = 0; d0 < D; d0 += 16) {   // 16 loads in flight (one code per thread: the chain of 64 dependent loads was 29 us)
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = d0 + i < D ? embed[(int64_t)(d0 + i) * K + k] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (d0 + i < D) {
        codes[(int64_t)k * D + d0 + i] = v[i];
        s += v[i] * v[i];  // sequential over d like a dim-0 reduction
      }
    }
  }
  e2[k] = s;
}

// fp32 packed weight -> split-f16 pair format of ISI_CONV_W16: every group of eight consecutive floats becomes 32 bytes
// {hi0 .. hi7 | lo0 .. lo7}, the f16 pieces of 1024 w exactly as the kernels of ISI_CONV_F16X3 compute them while
// staging (split_f16.h) -- a 16-byte piece is an MFMA operand fragment: staging is a plain copy
__global__ void split_weight_f16_kernel(const float4 *__restrict__ in, uint4 *__restrict__ out, int64_t ng) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ng) {
    uint4 hi, lo;
    f16s::weight8_encode(in[2 * i], in[2 * i + 1], hi, lo);
    out[2 * i] = hi;
    out[2 * i + 1] = lo;
  }
}

// nn.Linear weight w [N][K] -> the operand of its input-gradient GEMM dX = dY W, i.e. W^T as a [K][N] "weight": the fp32
// transpose followed by its split-bf16 pair copy (groups of 8 consecutive n as {hi[8] | lo[8]} bf16, 32 bytes where the
// 8 floats would be; pieces exactly as split_bf16.h forms them while staging).  One launch per weight and step instead
// of a transposing copy, and the GEMM kernel stages the weight tile by plain copies.  N, K multiples of 32.
__global__ __launch_bounds__(256) void linear_wT_bf16_kernel(const float *__restrict__ w, float *__restrict__ outT,
                                                             float *__restrict__ out16, int N, int K) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) {
    const int r = i >> 5, c = i & 31;                      // row n0 + r, column k0 + c: 128-byte row segments
    tile[r][c] = w[(size_t)(n0 + r) * K + k0 + c];
  }
  __syncthreads();
  if (tid < 128) {
    const int kk = tid >> 2, g = tid & 3;                  // output row k0 + kk, group of 8 n
    float4 a, b;
    a.x = tile[8 * g + 0][kk]; a.y = tile[8 * g + 1][kk]; a.z = tile[8 * g + 2][kk]; a.w = tile[8 * g + 3][kk];
    b.x = tile[8 * g + 4][kk]; b.y = tile[8 * g + 5][kk]; b.z = tile[8 * g + 6][kk]; b.w = tile[8 * g + 7][kk];
    const size_t o = (size_t)(k0 + kk) * N + n0 + 8 * g;
    *reinterpret_cast<float4 *>(outT + o) = a;
    *reinterpret_cast<float4 *>(outT + o + 4) = b;
    uint2 h0, l0, h1, l1;
    split_f4(a, h0, l0);
    split_f4(b, h1, l1);
    *reinterpret_cast<uint4 *>(out16 + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(out16 + o + 4) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}

// The same for MANY weights in one launch (a training step of the prior re-packs ~80 weights: 80 launches of ~5 us, most
// of it launch overhead): table[t] = {w, out, N, K} as four 64-bit words in device memory, grid (tiles of the largest
// weight, number of weights).
__global__ __launch_bounds__(256) void linear_wT_bf16_multi_kernel(const long long *__restrict__ table) {
  const long long *e = table + 4 * (size_t)blockIdx.y;
  const int N = (int)e[2], K = (int)e[3];
  const int tn = N / 32, tiles = tn * (K / 32);
  __shared__ float tile[32][33];
  const float *w = reinterpret_cast<const float *>(e[0]);
  float *outT = reinterpret_cast<float *>(e[1]);
  float *out16 = outT + (size_t)N * K;
  const int tid = threadIdx.x;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {     // (uniform per workgroup)
    const int n0 = (t % tn) * 32, k0 = (t / tn) * 32;
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
      const int r = i >> 5, c = i & 31;
      tile[r][c] = w[(size_t)(n0 + r) * K + k0 + c];
    }
    __syncthreads();
    if (tid < 128) {
      const int kk = tid >> 2, g = tid & 3;
      float4 a, b;
      a.x = tile[8 * g + 0][kk]; a.y = tile[8 * g + 1][kk]; a.z = tile[8 * g + 2][kk]; a.w = tile[8 * g + 3][kk];
      b.x = tile[8 * g + 4][kk]; b.y = tile[8 * g + 5][kk]; b.z = tile[8 * g + 6][kk]; b.w = tile[8 * g + 7][kk];
      const size_t o = (size_t)(k0 + kk) * N + n0 + 8 * g;
      *reinterpret_cast<float4 *>(outT + o) = a;
      *reinterpret_cast<float4 *>(outT + o + 4) = b;
      uint2 h0, l0, h1, l1;
      split_f4(a, h0, l0);
      split_f4(b, h1, l1);
      *reinterpret_cast<uint4 *>(out16 + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4 *>(out16 + o + 4) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
  }
}
int pack_linear_wT_bf16_multi(const void *table_dev, int n, int blocks_per_weight, hipStream_t stream) {
  if (!table_dev || n <= 0 || n > 65535 || blocks_per_weight <= 0) return invalid("pack_linear_wT_bf16_multi: bad argument");
  hipLaunchKernelGGL(linear_wT_bf16_multi_kernel, dim3(blocks_per_weight, n), dim3(256), 0, stream,
                     static_cast<const long long *>(table_dev));
  return check_launch("pack_linear_wT_bf16_multi");
}

// ---- every weight layout of a VQ-VAE training step in ONE launch (round 5).  A step re-packs each convolution weight for
// the forward (packed + split-f16 pair copy) and for its input-gradient convolution, plus both codebooks: 63 launches of
// ~5 us in round 4.  table[t] = 8 x int64 {kind, src, dst, d0, d1, KH, KW, aux}; grid (blocks per entry, entries); a thread
// forms 8 consecutive elements of a packed row (Kpad is a multiple of 32: a group never straddles rows).
//   kind 0 / 1  Conv2d weight [d0 = Cout][d1 = Cin][KH][KW] -> [Cout][Kpad]            (1: + pair copy behind it)
//   kind 2      its stride-1 input-gradient weight          -> [Cin][Kpad'], rotated   (pack_conv_dgrad_kernel)
//   kind 3 / 4  ConvTranspose2d(k4,s2,p1) weight [d0 = Cin][d1 = Cout][4][4] -> four phase matrices (4: + pair copy)
//   kind 5      the few-channel layout of the same           -> [16 Cout][Cin] + pair copy (convT_small_f32.hip)
//   kind 6      codebook `embed` [d0 = D][d1 = K] -> codes [K][D] at dst, |e|^2 [K] at aux (pack_codebook_kernel)
__device__ __forceinline__ float pack_multi_elem(const int kind, const float *__restrict__ w, const int64_t i, const int d0,
                                                 const int d1, const int KH, const int KW, const int Kpad) {
  if (kind <= 1) {
    const int co = (int)(i / Kpad), k = (int)(i - (int64_t)co * Kpad);
    if (k >= KH * KW * d1) return 0.f;
    const int tap = k / d1, ci = k - tap * d1, kh = tap / KW, kw = tap - kh * KW;
    return w[(((int64_t)co * d1 + ci) * KH + kh) * KW + kw];
  }
  if (kind == 2) {
    const int ci = (int)(i / Kpad), k = (int)(i - (int64_t)ci * Kpad);
    if (k >= KH * KW * d0) return 0.f;
    const int tap = k / d0, co = k - tap * d0, kh = tap / KW, kw = tap - kh * KW;
    return w[(((int64_t)co * d1 + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
  }
  if (kind <= 4) {
    const int Cin = d0, Cout = d1;
    const int ph = (int)(i / ((int64_t)Cout * Kpad));
    const int64_t rem = i - (int64_t)ph * Cout * Kpad;
    const int co = (int)(rem / Kpad), k = (int)(rem - (int64_t)co * Kpad);
    if (k >= 4 * Cin) return 0.f;
    const int tap = k / Cin, ci = k - tap * Cin, ty = tap >> 1, tx = tap & 1;
    const int ky = 3 - (ph >> 1) - 2 * ty, kx = 3 - (ph & 1) - 2 * tx;
    return w[(((int64_t)ci * Cout + co) * 4 + ky) * 4 + kx];
  }
  {   // kind 5
    const int Cin = d0, Cout = d1;
    const int ci = (int)(i % Cin), n = (int)(i / Cin), co = n % Cout, t = n / Cout;
    return w[((ci * Cout + co) * 4 + (t >> 2)) * 4 + (t & 3)];
  }
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const long long *__restrict__ table) {
  const long long *e = table + 8 * (size_t)blockIdx.y;
  const int kind = (int)e[0];
  const float *w = reinterpret_cast<const float *>(e[1]);
  float *out = reinterpret_cast<float *>(e[2]);
  const int d0 = (int)e[3], d1 = (int)e[4], KH = (int)e[5], KW = (int)e[6];
  if (kind == 6) {
    float *e2 = reinterpret_cast<float *>(e[7]);
    const int D = d0, K = d1;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int c0 = 0; c0 < D; c0 += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = c0 + i < D ? w[(int64_t)(c0 + i) * K + k] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (c0 + i < D) { out[(int64_t)k * D + c0 + i] = v[i]; s += v[i] * v[i]; }   // sequential over d (pack_codebook_kernel)
      }
      e2[k] = s;
    }
    return;
  }
  int Kpad;
  int64_t total;
  if (kind <= 1) { Kpad = (KH * KW * d1 + 31) / 32 * 32; total = (int64_t)d0 * Kpad; }          // (kBK = 32)
  else if (kind == 2) { Kpad = (KH * KW * d0 + 31) / 32 * 32; total = (int64_t)d1 * Kpad; }
  else if (kind <= 4) { Kpad = (4 * d0 + 31) / 32 * 32; total = (int64_t)4 * d1 * Kpad; }
  else { Kpad = d0; total = (int64_t)16 * d0 * d1; }
  const bool pairs = kind == 1 || kind == 4 || kind == 5;
  const int64_t groups = total / 8;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = pack_multi_elem(kind, w, gi * 8 + j, d0, d1, KH, KW, Kpad);
    const float4 a = make_float4(v[0], v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
    reinterpret_cast<float4 *>(out)[2 * gi] = a;
    reinterpret_cast<float4 *>(out)[2 * gi + 1] = b;
    if (pairs) {
      uint4 hi, lo;
      f16s::weight8_encode(a, b, hi, lo);
      uint4 *pr = reinterpret_cast<uint4 *>(out + total);
      pr[2 * gi] = hi;
      pr[2 * gi + 1] = lo;
    }
  }
}

int pack_multi(const void *table_dev, int n, int blocks_per_entry, hipStream_t stream) {
  if (!table_dev || n <= 0 || n > 65535 || blocks_per_entry <= 0) return invalid("pack_multi: bad argument");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks_per_entry, n), dim3(256), 0, stream, static_cast<const long long *>(table_dev));
  return check_launch("pack_multi");
}

int pack_linear_wT_bf16(const float *w, float *out, int N, int K, hipStream_t stream) {
  if (!w || !out || N <= 0 || K <= 0 || (N & 31) || (K & 31)) return invalid("pack_linear_wT_bf16: N, K multiples of 32");
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15)
    return invalid("pack_linear_wT_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(linear_wT_bf16_kernel, dim3(N / 32, K / 32), dim3(256), 0, stream, w, out, out + (size_t)N * K, N, K);
  return check_launch("pack_linear_wT_bf16");
}

int split_conv_weight_f16(const float *packed, float *out, int64_t n_floats, hipStream_t stream) {
  if (!packed || !out || n_floats <= 0 || (n_floats & 7)) return invalid("split_conv_weight_f16: bad argument");
  if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15)
    return invalid("split_conv_weight_f16: pointers must be 16-byte aligned");
  const int64_t ng = n_floats / 8;
  hipLaunchKernelGGL(split_weight_f16_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float4 *>(packed), reinterpret_cast<uint4 *>(out), ng);
  return check_launch("split_conv_weight_f16");
}

// fp32 <-> activation pair format (ISI_CONV_OUT_PAIR / IN*_PAIR, split_f16.h): groups of 8 consecutive elements
__global__ void pair_encode_kernel(const float4 *__restrict__ x, uint4 *__restrict__ out, int64_t ng) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += stride) {
    uint4 hi, lo;
    f16s::pair8_encode(x[2 * i], x[2 * i + 1], hi, lo);
    out[2 * i] = hi;
    out[2 * i + 1] = lo;
  }
}
__global__ void pair_decode_kernel(const uint4 *__restrict__ in, float4 *__restrict__ x, int64_t ng) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += stride) {
    float4 a, b;
    f16s::pair8_decode(in[2 * i], in[2 * i + 1], a, b);
    x[2 * i] = a;
    x[2 * i + 1] = b;
  }
}
int pair_encode_f32(const float *x, float *out, int64_t n, hipStream_t stream) {
  if (!x || !out || n < 0 || (n & 7)) return invalid("pair_encode: need a multiple of 8 elements");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return invalid("pair_encode: pointers must be 16-byte aligned");
  if (n == 0) return ISI_OK;
  const int64_t ng = n / 8, blocks = (ng + 255) / 256;
  hipLaunchKernelGGL(pair_encode_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                     reinterpret_cast<const float4 *>(x), reinterpret_cast<uint4 *>(out), ng);
  return check_launch("pair_encode_f32");
}
int pair_decode_f32(const float *in, float *x, int64_t n, hipStream_t stream) {
  if (!x || !in || n < 0 || (n & 7)) return invalid("pair_decode: need a multiple of 8 elements");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(in)) & 15) return invalid("pair_decode: pointers must be 16-byte aligned");
  if (n == 0) return ISI_OK;
  const int64_t ng = n / 8, blocks = (ng + 255) / 256;
  hipLaunchKernelGGL(pair_decode_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                     reinterpret_cast<const uint4 *>(in), reinterpret_cast<float4 *>(x), ng);
  return check_launch("pair_decode_f32");
}

__global__ void relu_inplace_kernel(float *__restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    x[i] = x[i] < 0.f ? 0.f : x[i];   // NaN stays NaN
}

int relu_inplace_f32(float *x, int64_t n, hipStream_t stream) {
  if (!x || n < 0) return invalid("relu_inplace: bad argument");
  if (n == 0) return ISI_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     stream, x, n);
  return check_launch("relu_inplace_f32");
}

size_t packed_conv_weight_floats(int Cout, int Cin, int KH, int KW) {
  return (size_t)Cout * round_up((size_t)KH * KW * Cin, kBK);
}
size_t packed_convT_k4s2_weight_floats(int Cin, int Cout) {
  if (convT_small_applicable(Cin, Cout)) return (size_t)16 * Cin * Cout;  // wn[(ky*4+kx)*Cout+co][ci]
  return (size_t)4 * Cout * round_up((size_t)4 * Cin, kBK);
}

int pack_conv_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW,
                         hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0)
    return invalid("pack_conv_weight: bad argument");
  const int Kpad = (int)round_up((size_t)KH * KW * Cin, kBK);
  const int64_t total = (int64_t)Cout * Kpad;
  hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     w, packed, Cout, Cin, KH, KW, Kpad);
  return check_launch("pack_conv_weight_f32");
}

int pack_conv_weight_w16_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW, hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return invalid("pack_conv_weight_w16: bad argument");
  if (reinterpret_cast<uintptr_t>(packed) & 15) return invalid("pack_conv_weight_w16: output must be 16-byte aligned");
  const int Kpad = (int)round_up((size_t)KH * KW * Cin, kBK);
  const int64_t groups = (int64_t)Cout * Kpad / 8;
  hipLaunchKernelGGL(pack_conv_w16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, w, packed, Cout, Cin,
                     KH, KW, Kpad);
  return check_launch("pack_conv_weight_w16_f32");
}

int pack_conv_dgrad_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW,
                               hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0)
    return invalid("pack_conv_dgrad_weight: bad argument");
  const int Kpad = (int)round_up((size_t)KH * KW * Cout, kBK);
  const int64_t total = (int64_t)Cin * Kpad;
  hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     w, packed, Cout, Cin, KH, KW, Kpad);
  return check_launch("pack_conv_dgrad_weight_f32");
}

int pack_convT_k4s2_weight_f32(const float *w, float *packed, int Cin, int Cout,
                               hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0) return invalid("pack_convT_weight: bad argument");
  if (convT_small_applicable(Cin, Cout)) return pack_convT_small_weight_f32(w, packed, Cin, Cout, stream);
  const int Kpad = (int)round_up((size_t)4 * Cin, kBK);
  const int64_t total = (int64_t)4 * Cout * Kpad;
  hipLaunchKernelGGL(pack_convT_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     w, packed, Cin, Cout, Kpad);
  return check_launch("pack_convT_k4s2_weight_f32");
}

int pack_codebook_f32(const float *embed, float *codes_kd, float *e2, int D, int K,
                      hipStream_t stream) {
  if (!embed || !codes_kd || !e2 || D <= 0 || K <= 0) return invalid("pack_codebook: bad argument");
  hipLaunchKernelGGL(pack_codebook_kernel, dim3((K + 63) / 64), dim3(64), 0, stream, embed,
                     codes_kd, e2, D, K);
  return check_launch("pack_codebook_f32");
}

}  // namespace isi
