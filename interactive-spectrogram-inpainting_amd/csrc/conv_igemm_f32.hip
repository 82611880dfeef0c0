// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: every product is one fp32 FMA, bitwise an fmaf chain).
//
//   M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin (taps outer, channels inner)
//   A[m][k] = input pixel gathered on the fly (zero padded), channels-last
//   B[n][k] = packed weight row (isi_pack_conv_weight_f32)
//
// One workgroup = 256 threads = 4 wavefronts computes a BM x BN tile; each
// wave owns a (BM/WM) x (BN/WN) sub-tile built from 32x32 MFMA tiles.  A and B
// K-chunks of 32 are staged global -> registers -> LDS (rows padded to 36
// floats: conflict-free ds_read_b128 fragments), double buffered, the loads of
// chunk k+1 being issued before the MFMAs of chunk k.
//
// The same kernel runs the four 2x2 phase convolutions of
// ConvTranspose2d(k4,s2,p1) with blockIdx.z = phase.
//
// Replaces (reference, torch.nn): nn.Conv2d / nn.ConvTranspose2d / nn.ReLU /
// residual add / torch.cat at vqvae/encoder_decoder.py:22-35,95-112,138,199-215
// and vqvae/vqvae.py:193-201,260,270-272,282.
#include "isi_common.h"
#include "prof.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKArgs {
  const float *in0, *in1, *w, *bias, *res;
  float *out;
  int C0, C1, Cin;
  int64_t s0n, s0c, s0h, s0w;  // source 0 element strides
  int64_t s1n, s1h, s1w;       // source 1 (channel stride 1)
  int64_t rn, rc, rh, rw;      // residual strides (logical output coordinates)
  int64_t on, oc, oh, ow;      // output strides, in units of GEMM-grid pixels
  int H, W, OH, OW, Cout, K, Kpad, KH, KW, stride, relu, M;
  int nphase;
  int pad_y[4], pad_x[4];
  int64_t w_off[4], out_off[4], res_off[4];
};

constexpr int LDK = 36;  // padded LDS row (floats): 144 B, 16-B aligned, bank-conflict free

template <int BM, int BN, int WM, int WN, bool SCALAR_A>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvKArgs p) {
  constexpr int TM = BM / WM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int RA = BM / 32;  // A rows staged per thread
  constexpr int RB = BN / 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TM >= 1 && TN >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;                      // [2][BM*LDK]
  float *Bs = smem + 2 * BM * LDK;       // [2][BN*LDK]
  int *row_b = reinterpret_cast<int *>(Bs + 2 * BN * LDK);  // [BM]
  int *row_y = row_b + BM;
  int *row_x = row_y + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM);
  const int wn0 = (wave % WN) * (BN / WN);
  const int phase = blockIdx.z;

  // XCD-aware tile order: consecutive M tiles (neighbouring pixels, shared
  // halo rows) are given to one XCD so their re-reads hit that XCD's L2.
  const int ntile = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntile / 8, r = ntile % 8, xcd = bid % 8, idx = bid / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = tile * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-row output coordinates -> LDS
  if (tid < BM) {
    const int m = m0 + tid;
    int b = -1, oy = 0, ox = 0;
    if (m < p.M) {
      b = m / (p.OH * p.OW);
      const int rem = m - b * (p.OH * p.OW);
      oy = rem / p.OW;
      ox = rem - oy * p.OW;
    }
    row_b[tid] = b;
    row_y[tid] = oy;
    row_x[tid] = ox;
  }
  __syncthreads();

  const int lrow = tid >> 3;  // 0..31
  const int lq = tid & 7;     // quad inside the 32-wide K chunk
  const int pad_y = p.pad_y[phase], pad_x = p.pad_x[phase];
  const float *wbase = p.w + p.w_off[phase];

  int a_b[RA], a_y0[RA], a_x0[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int r = lrow + 32 * j;
    a_b[j] = row_b[r];
    a_y0[j] = row_y[r] * p.stride - pad_y;
    a_x0[j] = row_x[r] * p.stride - pad_x;
  }

  float4 ra[RA], rb[RB];

  auto load_chunk = [&](int kc) {
    const int kk = kc * kBK + lq * 4;
    if constexpr (!SCALAR_A) {
      const bool kvalid = kk < p.K;
      const int tap = kk / p.Cin;
      int c = kk - tap * p.Cin;
      const int kh = tap / p.KW;
      const int kw = tap - kh * p.KW;
      const float *src = p.in0;
      int64_t sn = p.s0n, sh = p.s0h, sw = p.s0w;
      if (c >= p.C0) {
        c -= p.C0;
        src = p.in1;
        sn = p.s1n; sh = p.s1h; sw = p.s1w;
      }
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int iy = a_y0[j] + kh, ix = a_x0[j] + kw;
        const bool ok = kvalid && a_b[j] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4 *>(src + a_b[j] * sn + iy * sh + ix * sw + c);
        ra[j] = v;
      }
    } else {
      // element-wise gather with arbitrary strides (NCHW sources, Cin % 4 != 0)
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k1 = kk + e;
          const int tap = k1 / p.Cin;
          const int c = k1 - tap * p.Cin;
          const int kh = tap / p.KW;
          const int kw = tap - kh * p.KW;
          const int iy = a_y0[j] + kh, ix = a_x0[j] + kw;
          const bool ok = k1 < p.K && a_b[j] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          float t = 0.f;
          if (ok) {
            if (c < p.C0) t = p.in0[a_b[j] * p.s0n + c * p.s0c + iy * p.s0h + ix * p.s0w];
            else t = p.in1[a_b[j] * p.s1n + (c - p.C0) + iy * p.s1h + ix * p.s1w];
          }
          v[e] = t;
        }
        ra[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int n = n0 + lrow + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < p.Cout) v = *reinterpret_cast<const float4 *>(wbase + (int64_t)n * p.Kpad + kk);
      rb[j] = v;
    }
  };

  auto store_chunk = [&](int buf) {
    float *a = As + buf * BM * LDK;
    float *b = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < RA; ++j)
      *reinterpret_cast<float4 *>(a + (lrow + 32 * j) * LDK + lq * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<float4 *>(b + (lrow + 32 * j) * LDK + lq * 4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / kBK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fq = lane >> 5;

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) load_chunk(kc + 1);  // global loads in flight under the MFMAs

    const float *a = As + buf * BM * LDK + (wm0 + frow) * LDK + fq * 4;
    const float *b = Bs + buf * BN * LDK + (wn0 + frow) * LDK + fq * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK + s * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK + s * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
            const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
      }
    }
    if (kc + 1 < nk) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, residual, ReLU, store.  C layout of the 32x32 tile:
  // col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  const int64_t out_off = p.out_off[phase], res_off = p.res_off[phase];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + frow;
    if (n >= p.Cout) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
        const int b = row_b[row];
        if (b < 0) continue;
        const int oy = row_y[row], ox = row_x[row];
        float v = acc[i][j][r] + bias;
        if (p.res) v += p.res[res_off + b * p.rn + n * p.rc + oy * p.rh + ox * p.rw];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[out_off + b * p.on + n * p.oc + oy * p.oh + ox * p.ow] = v;
      }
    }
  }
}

template <int BM, int BN>
constexpr size_t conv_smem_bytes() {
  return (size_t)(2 * BM * LDK + 2 * BN * LDK) * sizeof(float) + 3 * BM * sizeof(int);
}

template <int BM, int BN, int WM, int WN, bool SCALAR_A>
static int launch_cfg(const ConvKArgs &a, hipStream_t stream) {
  auto kern = conv_igemm_f32_kernel<BM, BN, WM, WN, SCALAR_A>;
  constexpr size_t smem = conv_smem_bytes<BM, BN>();
  static bool attr_set = false;  // idempotent; racing threads set the same value
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_igemm)");
    attr_set = true;
  }
  dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.nphase);
  {
    // algorithmic work: every MAC of the convolution once; input read once,
    // output written once, weights once (DESIGN.md "roofline accounting")
    const double np = a.nphase;
    const double flops = 2.0 * a.M * np * a.Cout * a.K;
    const double in_px = a.nphase == 1 ? (double)a.M / (a.OH * a.OW) * a.H * a.W : (double)a.M;
    const double bytes = 4.0 * (in_px * a.Cin + (double)a.M * np * a.Cout * (a.res ? 2 : 1) +
                                np * a.Cout * a.K);
    const int kid = SCALAR_A ? prof::K_CONV_GATHER
                             : (BN == 128 ? prof::K_CONV_128x128 : BN == 64 ? prof::K_CONV_128x64 : prof::K_CONV_128x32);
    prof::Scope scope(kid, flops, bytes, stream);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  }
  return check_launch("conv_igemm_f32");
}

static int launch_conv(const ConvKArgs &a, bool scalar_a, hipStream_t stream) {
  if (scalar_a) {
    if (a.Cout <= 32) return launch_cfg<128, 32, 4, 1, true>(a, stream);
    if (a.Cout <= 64) return launch_cfg<128, 64, 2, 2, true>(a, stream);
    return launch_cfg<128, 128, 2, 2, true>(a, stream);
  }
  if (a.Cout <= 32) return launch_cfg<128, 32, 4, 1, false>(a, stream);
  if (a.Cout <= 64) return launch_cfg<128, 64, 2, 2, false>(a, stream);
  return launch_cfg<128, 128, 2, 2, false>(a, stream);
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int conv2d_f32(const isi_src *s0, const isi_src *s1, const float *packed_w, const float *bias,
               const isi_src *res, const isi_dst *dst, int B, int H, int W, int Cout, int KH,
               int KW, int stride, int pad, int relu, hipStream_t stream) {
  if (!s0 || !s0->ptr || !packed_w || !dst || !dst->ptr) return invalid("conv2d: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return invalid("conv2d: bad shape");
  const int OH = (H + 2 * pad - KH) / stride + 1;
  const int OW = (W + 2 * pad - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return invalid("conv2d: empty output");
  if ((int64_t)B * OH * OW > INT32_MAX) return unsupported("conv2d: more than 2^31 output pixels");
  ConvKArgs a;
  memset(&a, 0, sizeof a);
  a.in0 = s0->ptr; a.C0 = s0->C;
  a.s0n = s0->sn; a.s0c = s0->sc; a.s0h = s0->sh; a.s0w = s0->sw;
  const bool two = s1 && s1->ptr;
  a.in1 = two ? s1->ptr : s0->ptr;
  a.C1 = two ? s1->C : 0;
  if (two) { a.s1n = s1->sn; a.s1h = s1->sh; a.s1w = s1->sw; }
  a.Cin = a.C0 + a.C1;
  a.w = packed_w; a.bias = bias;
  a.res = (res && res->ptr) ? res->ptr : nullptr;
  if (a.res) { a.rn = res->sn; a.rc = res->sc; a.rh = res->sh; a.rw = res->sw; }
  a.out = dst->ptr; a.on = dst->sn; a.oc = dst->sc; a.oh = dst->sh; a.ow = dst->sw;
  a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.Cout = Cout;
  a.K = KH * KW * a.Cin; a.Kpad = (int)round_up(a.K, kBK);
  a.KH = KH; a.KW = KW; a.stride = stride; a.relu = relu; a.M = B * OH * OW;
  a.nphase = 1; a.pad_y[0] = pad; a.pad_x[0] = pad;
  if (two && s1->sc != 1) return unsupported("conv2d: second source must be channels-last");
  bool vec = s0->sc == 1 && (a.C0 % 4 == 0) && (a.C1 % 4 == 0) && aligned16(s0->ptr) &&
             (s0->sn % 4 == 0) && (s0->sh % 4 == 0) && (s0->sw % 4 == 0);
  if (two) vec = vec && aligned16(s1->ptr) && (s1->sn % 4 == 0) && (s1->sh % 4 == 0) && (s1->sw % 4 == 0);
  if (!aligned16(packed_w)) return invalid("conv2d: packed weight must be 16-byte aligned");
  return launch_conv(a, !vec, stream);
}

int conv_transpose2d_k4s2_f32(const isi_src *s, const float *packed_w, const float *bias,
                              const isi_dst *dst, int B, int H, int W, int Cout, int relu,
                              hipStream_t stream) {
  if (!s || !s->ptr || !packed_w || !dst || !dst->ptr) return invalid("convT: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return invalid("convT: bad shape");
  if ((int64_t)B * H * W > INT32_MAX) return unsupported("convT: more than 2^31 pixels per phase");
  ConvKArgs a;
  memset(&a, 0, sizeof a);
  a.in0 = s->ptr; a.in1 = s->ptr; a.C0 = s->C; a.C1 = 0; a.Cin = s->C;
  a.s0n = s->sn; a.s0c = s->sc; a.s0h = s->sh; a.s0w = s->sw;
  a.w = packed_w; a.bias = bias; a.res = nullptr;
  a.out = dst->ptr;
  // GEMM-grid pixel (m_y, m_x) of phase (py,px) is output pixel (2 m_y + py, 2 m_x + px)
  a.on = dst->sn; a.oc = dst->sc; a.oh = 2 * dst->sh; a.ow = 2 * dst->sw;
  a.H = H; a.W = W; a.OH = H; a.OW = W; a.Cout = Cout;
  a.K = 4 * a.Cin; a.Kpad = (int)round_up(a.K, kBK);
  a.KH = 2; a.KW = 2; a.stride = 1; a.relu = relu; a.M = B * H * W;
  a.nphase = 4;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int ph = py * 2 + px;
      a.pad_y[ph] = 1 - py;
      a.pad_x[ph] = 1 - px;
      a.w_off[ph] = (int64_t)ph * Cout * a.Kpad;
      a.out_off[ph] = py * dst->sh + px * dst->sw;
    }
  const bool vec = s->sc == 1 && (a.C0 % 4 == 0) && aligned16(s->ptr) && (s->sn % 4 == 0) &&
                   (s->sh % 4 == 0) && (s->sw % 4 == 0);
  if (!aligned16(packed_w)) return invalid("convT: packed weight must be 16-byte aligned");
  return launch_conv(a, !vec, stream);
}

}  // namespace isi
