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
  for (int d = 0; d < D; ++d) {
    const float v = embed[(int64_t)d * K + k];
    codes[(int64_t)k * D + d] = v;
    s += v * v;  // sequential over d like a dim-0 reduction
  }
  e2[k] = s;
}

// fp32 packed weight -> split-f16 pair format of ISI_CONV_W16: every quad of four consecutive floats becomes
// 16 bytes {hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3}, the f16 pieces of 1024 w exactly as the kernels of ISI_CONV_F16X3
// compute them while staging (conv_igemm_f32.hip: split_f16x4) -- a staged quad is then one 16-byte load and two
// 8-byte LDS stores, no conversion
__global__ void split_weight_f16_kernel(const float4 *__restrict__ in, uint4 *__restrict__ out, int64_t nq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) out[i] = f16s::weight_encode(in[i]);
}

int split_conv_weight_f16(const float *packed, float *out, int64_t n_floats, hipStream_t stream) {
  if (!packed || !out || n_floats <= 0 || (n_floats & 3)) return invalid("split_conv_weight_f16: bad argument");
  if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15)
    return invalid("split_conv_weight_f16: pointers must be 16-byte aligned");
  const int64_t nq = n_floats / 4;
  hipLaunchKernelGGL(split_weight_f16_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float4 *>(packed), reinterpret_cast<uint4 *>(out), nq);
  return check_launch("split_conv_weight_f16");
}

// fp32 <-> activation pair format (ISI_CONV_OUT_PAIR / IN*_PAIR): hi = f16(4 x) | lo = f16(4 x - hi) << 16
__global__ void pair_encode_kernel(const float *__restrict__ x, unsigned *__restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = f16s::pair_encode(x[i]);
}
__global__ void pair_decode_kernel(const unsigned *__restrict__ in, float *__restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = f16s::pair_decode(in[i]);
}
int pair_encode_f32(const float *x, float *out, int64_t n, hipStream_t stream) {
  if (!x || !out || n < 0) return invalid("pair_encode: bad argument");
  if (n == 0) return ISI_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(pair_encode_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, x,
                     reinterpret_cast<unsigned *>(out), n);
  return check_launch("pair_encode_f32");
}
int pair_decode_f32(const float *in, float *x, int64_t n, hipStream_t stream) {
  if (!x || !in || n < 0) return invalid("pair_decode: bad argument");
  if (n == 0) return ISI_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(pair_decode_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                     reinterpret_cast<const unsigned *>(in), x, n);
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
  hipLaunchKernelGGL(pack_codebook_kernel, dim3((K + 127) / 128), dim3(128), 0, stream, embed,
                     codes_kd, e2, D, K);
  return check_launch("pack_codebook_f32");
}

}  // namespace isi
