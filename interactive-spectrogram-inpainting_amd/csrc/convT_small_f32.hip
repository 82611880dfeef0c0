// ConvTranspose2d(kernel 4, stride 2, padding 1) with very few output channels
// (the decoder's last layer, C/2 -> in_channel = 2; reference
// vqvae/encoder_decoder.py:204-207).  With N = Cout <= 4 a matrix-core GEMM would
// waste > 90 % of every MFMA, and the layer is HBM-bound anyway (it reads the
// largest activation of the network and writes the spectrogram), so this is a
// direct fp32 VALU kernel:
//
//   * a workgroup owns a 4 x 64 patch of the input grid (-> 8 x 128 x Cout outputs);
//   * the (4+2) x (64+2) input halo is staged through LDS in 32-channel slices with
//     full-line coalesced buffer loads (zero padding = out-of-range buffer offset);
//   * each thread owns one input-grid pixel = a 2 x 2 output block and walks its
//     3 x 3 neighbourhood; the tap (ky,kx) an input neighbour feeds into output
//     phase (py,px) is fixed at compile time (dy = -1: ky 3 | dy = 0: ky 1 (py 0),
//     ky 2 (py 1) | dy = +1: ky 0);
//   * weights are wave-uniform and come in through scalar loads (SGPR operands of
//     v_fma_f32), so LDS bandwidth is spent on activations only.
//
// Weight layout (isi_pack_convT_k4s2_weight_f32 when Cout <= 4 and Cin % 32 == 0):
//   wk[ky][kx][ci][co].
#include "isi_common.h"
#include "prof.h"

namespace isi {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvTSmallArgs {
  const float *in, *wk, *bias;
  float *out;
  unsigned in_bytes;
  int H, W, Cin, relu;
  int on, oc, oh, ow;  // output element strides
};

namespace {
constexpr int TH = 4, TW = 64, HW_ = TW + 2, HH_ = TH + 2, LDP = 36;
constexpr unsigned OOB = 0xFFFFFFF0u;
}  // namespace

template <int CO>
__global__ __launch_bounds__(256) void convT_k4s2_small_kernel(const ConvTSmallArgs p) {
  __shared__ __attribute__((aligned(16))) float xs[HH_ * HW_ * LDP];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
  const float *__restrict__ wk = p.wk;
  const __amdgpu_buffer_rsrc_t rsi =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);

  float acc[2][2][CO];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[py][px][co] = p.bias ? p.bias[co] : 0.f;

  const int Cin = p.Cin;
  for (int c0 = 0; c0 < Cin; c0 += 32) {
    __syncthreads();
    for (int i = tid; i < HH_ * HW_ * 8; i += 256) {
      const int pix = i >> 3, q = i & 7;
      const int hy = pix / HW_, hx = pix - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * Cin + c0 + q * 4) * 4u : OOB;
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsi, off, 0, 0);
      *reinterpret_cast<i32x4 *>(xs + pix * LDP + q * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const float *xp = xs + ((ty + d) * HW_ + tx + f) * LDP;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 xv = *reinterpret_cast<const float4 *>(xp + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = e == 0 ? xv.x : e == 1 ? xv.y : e == 2 ? xv.z : xv.w;
            const int ci = c0 + q * 4 + e;
            // phases fed by neighbour offset d-1 (rows) / f-1 (cols)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              if ((d == 0 && py == 1) || (d == 2 && py == 0)) continue;
              const int ky = d == 0 ? 3 : d == 2 ? 0 : (py == 0 ? 1 : 2);
#pragma unroll
              for (int px = 0; px < 2; ++px) {
                if ((f == 0 && px == 1) || (f == 2 && px == 0)) continue;
                const int kx = f == 0 ? 3 : f == 2 ? 0 : (px == 0 ? 1 : 2);
#pragma unroll
                for (int co = 0; co < CO; ++co)
                  acc[py][px][co] = fmaf(x, wk[((ky * 4 + kx) * Cin + ci) * CO + co], acc[py][px][co]);
              }
            }
          }
        }
      }
    }
  }

  const int m = y0 + ty, n = x0 + tx;
  if (m < p.H && n < p.W) {
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float v = acc[py][px][co];
          if (p.relu) v = fmaxf(v, 0.f);
          p.out[b * p.on + co * p.oc + (2 * m + py) * p.oh + (2 * n + px) * p.ow] = v;
        }
  }
}

__global__ void pack_convT_small_kernel(const float *__restrict__ w, float *__restrict__ out, int Cin,
                                        int Cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * Cin * Cout) return;
  const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
  const int ky = t >> 2, kx = t & 3;
  out[i] = w[((ci * Cout + co) * 4 + ky) * 4 + kx];
}

bool convT_small_applicable(int Cin, int Cout) { return Cout >= 1 && Cout <= 4 && Cin % 32 == 0; }

int pack_convT_small_weight_f32(const float *w, float *packed, int Cin, int Cout, hipStream_t stream) {
  const int total = 16 * Cin * Cout;
  hipLaunchKernelGGL(pack_convT_small_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, packed,
                     Cin, Cout);
  return check_launch("pack_convT_small_weight_f32");
}

template <int CO>
static int launch_small(const ConvTSmallArgs &a, int B, hipStream_t stream) {
  const double M = (double)B * a.H * a.W;
  prof::Scope scope(prof::K_CONVT_SMALL, 2.0 * M * 16 * a.Cin * CO,
                    4.0 * (M * a.Cin + 4.0 * M * CO + 16.0 * a.Cin * CO), stream);
  hipLaunchKernelGGL(convT_k4s2_small_kernel<CO>, dim3((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, B), dim3(256),
                     0, stream, a);
  return check_launch("convT_k4s2_small_f32");
}

// src must be dense channels-last [B,H,W,Cin]; dst strides arbitrary (32-bit range checked by caller).
int convT_k4s2_small_f32(const float *in, const float *wk, const float *bias, float *out, int B, int H,
                         int W, int Cin, int Cout, int on, int oc, int oh, int ow, int relu,
                         hipStream_t stream) {
  if (!convT_small_applicable(Cin, Cout)) return unsupported("convT_small: need Cout <= 4 and Cin % 32 == 0");
  if (B > 65535 || (H + TH - 1) / TH > 65535) return unsupported("convT_small: grid too large");
  ConvTSmallArgs a;
  a.in = in; a.wk = wk; a.bias = bias; a.out = out;
  a.in_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
  a.H = H; a.W = W; a.Cin = Cin; a.relu = relu;
  a.on = on; a.oc = oc; a.oh = oh; a.ow = ow;
  switch (Cout) {
    case 1: return launch_small<1>(a, B, stream);
    case 2: return launch_small<2>(a, B, stream);
    case 3: return launch_small<3>(a, B, stream);
    default: return launch_small<4>(a, B, stream);
  }
}

}  // namespace isi
