// Element-wise / scan stages of the audio <-> spectrogram front-end (gfx950).
//
// The reference delegates the front-end to the absent package GANsynth_pytorch
// (`SpectrogramsHelper.to_spectrogram / to_audio`; call sites utils/misc.py:10-29,
// train_vqvae.py:392-400, sample.py:599, flask_server.py:596,1016); the arithmetic built
// here is the published GANSynth representation, specification: oracle/spectrogram_oracle.py.
//
// The two contractions of each direction run on the exact-fp32 matrix pipe through the
// existing GEMM kernel (isi_conv2d_f32):
//   STFT    = convolution of the audio, viewed as [B, 1, L/hop, hop] channels-last, with the
//             windowed DFT basis as a 1 x (n_fft/hop) kernel  ->  X[b, t, re(F) | im(F)]
//   mel     = 1x1 convolution with the mel matrix (power and unwrapped phase rows)
//   inverse = the transposed matrices; frames are overlap-added by overlap_add_kernel.
// The kernels below are the HBM-bound stages in between: one pass each, threads along the
// frequency axis (unit stride in the channels-last intermediates), a sequential scan over the
// (short) time axis where the phase is unwrapped / integrated, and LDS tile transposes to and
// from the [B, 2, F, T] layout the VQ-VAE consumes.
#include <cmath>

#include "isi_common.h"
#include "isi_internal.h"

namespace isi {

namespace {
constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;
constexpr float SPEC_EPS = 1e-6f;
// principal value in [-pi, pi], numpy.unwrap's convention at the ends
__device__ __forceinline__ float wrap_pi(float d) {
  float w = d + PI_F;
  w = w - TWO_PI_F * floorf(w / TWO_PI_F) - PI_F;
  if (w == -PI_F && d > 0.f) w = PI_F;
  return w;
}
}  // namespace

// stft [B,T,2F] (re block | im block) -> a, ph [B,T,F]
//   mel == 0: a = log(|X| + eps), ph = angle(X)
//   mel == 1: a = |X|^2,          ph = unwrapped angle (running sum of wrapped differences)
__global__ __launch_bounds__(256) void spec_polar_kernel(const float *__restrict__ x, float *__restrict__ a,
                                                         float *__restrict__ ph, int T, int F, int mel) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (f >= F) return;
  const float *xr = x + (size_t)b * T * 2 * F + f;
  float *ar = a + (size_t)b * T * F + f, *pr = ph + (size_t)b * T * F + f;
  float prev = 0.f, run = 0.f;
  for (int t = 0; t < T; ++t) {
    const float re = xr[(size_t)t * 2 * F], im = xr[(size_t)t * 2 * F + F];
    const float ang = atan2f(im, re);
    if (mel) {
      run = t == 0 ? ang : run + wrap_pi(ang - prev);
      prev = ang;
      ar[(size_t)t * F] = re * re + im * im;
      pr[(size_t)t * F] = run;
    } else {
      ar[(size_t)t * F] = logf(sqrtf(re * re + im * im) + SPEC_EPS);
      pr[(size_t)t * F] = ang;
    }
  }
}

// a, ph [B,T,F] -> out [B,2,F,T]:  out0 = mel ? log(a + eps) : a ;  out1 = IF(ph) (wrapped difference / pi)
// 32(t) x 32(f) tiles transposed through LDS: reads unit-stride in f, writes unit-stride in t.
__global__ __launch_bounds__(256) void spec_finish_kernel(const float *__restrict__ a, const float *__restrict__ ph,
                                                          float *__restrict__ out, int T, int F, int mel) {
  __shared__ float ta[32][33], tp[32][33];
  const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, f = f0 + tx;
    float va = 0.f, vp = 0.f;
    if (t < T && f < F) {
      const size_t o = ((size_t)b * T + t) * F + f;
      va = mel ? logf(a[o] + SPEC_EPS) : a[o];
      const float cur = ph[o];
      vp = t == 0 ? cur : wrap_pi(cur - ph[o - F]);
      vp *= (1.f / PI_F);
    }
    ta[i][tx] = va;
    tp[i][tx] = vp;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int f = f0 + i, t = t0 + tx;
    if (f < F && t < T) {
      out[(((size_t)b * 2 + 0) * F + f) * T + t] = ta[tx][i];
      out[(((size_t)b * 2 + 1) * F + f) * T + t] = tp[tx][i];
    }
  }
}

// spec [B,2,F,T] -> a = exp(ch0), ph = running sum of ch1 * pi, both [B,T,F].
// A workgroup owns 32 frequencies; 32 x 32 tiles are transposed through LDS and the running
// phase of each frequency is carried from tile to tile.
__global__ __launch_bounds__(256) void spec_inverse_prepare_kernel(const float *__restrict__ spec, float *__restrict__ a,
                                                                   float *__restrict__ ph, int T, int F) {
  __shared__ float ta[32][33], tp[32][33];
  __shared__ float carry[32];
  const int f0 = blockIdx.x * 32, b = blockIdx.y;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (threadIdx.x < 32) carry[threadIdx.x] = 0.f;
  for (int t0 = 0; t0 < T; t0 += 32) {
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {  // rows = frequencies, unit stride along t
      const int f = f0 + i, t = t0 + tx;
      float va = 0.f, vp = 0.f;
      if (f < F && t < T) {
        va = expf(spec[(((size_t)b * 2 + 0) * F + f) * T + t]);
        vp = spec[(((size_t)b * 2 + 1) * F + f) * T + t] * PI_F;
      }
      ta[i][tx] = va;
      tp[i][tx] = vp;
    }
    __syncthreads();
    if (threadIdx.x < 32) {  // sequential scan of one frequency's 32 frames
      float run = carry[threadIdx.x];
      for (int j = 0; j < 32; ++j) {
        run += tp[threadIdx.x][j];
        tp[threadIdx.x][j] = run;
      }
      carry[threadIdx.x] = run;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {  // rows = frames, unit stride along f
      const int t = t0 + i, f = f0 + tx;
      if (t < T && f < F) {
        const size_t o = ((size_t)b * T + t) * F + f;
        a[o] = ta[tx][i];
        ph[o] = tp[tx][i];
      }
    }
  }
}

// a, ph [B,T,F] -> stft [B,T,2F]: mag = mel ? sqrt(max(a,0) + eps) : a ; (re, im) = mag (cos ph, sin ph)
__global__ void spec_to_stft_kernel(const float *__restrict__ a, const float *__restrict__ ph, float *__restrict__ x,
                                    int64_t rows, int F, int mel) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t r = i / F;
  const int f = (int)(i - r * F);
  float mag = a[i];
  if (mel) mag = expf(0.5f * logf(fmaxf(mag, 0.f) + SPEC_EPS));
  float s, c;
  sincosf(ph[i], &s, &c);
  x[r * 2 * F + f] = mag * c;
  x[r * 2 * F + F + f] = mag * s;
}

// audio[b, n] = sum over frames t of frames[b, t, left + n - t hop]   (frames already carry the synthesis window)
__global__ void overlap_add_kernel(const float *__restrict__ frames, float *__restrict__ audio, int T, int n_fft,
                                   int hop, int left, int64_t L) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= L) return;
  const int64_t pos = n + left;                 // position in the padded signal
  int t_hi = (int)(pos / hop);
  if (t_hi > T - 1) t_hi = T - 1;
  float s = 0.f;
  for (int t = t_hi; t >= 0; --t) {
    const int64_t k = pos - (int64_t)t * hop;
    if (k >= n_fft) break;
    s += frames[((size_t)b * T + t) * n_fft + k];
  }
  audio[(size_t)b * L + n] = s;
}

// Per-channel affine map of a [B,2,H,W] spectrogram with the masked-phase rule, one pass:
//   y0 = a0 x0 + b0 ;  y1 = a1 x1 + b1, forced to 0 where the log-magnitude (ref channel 0 when `ref` is given,
//   else y0) is <= thr.  Serves DataNormalizer.normalize / denormalize (vqvae.py:254-255,297-300), the
//   masked-phase transform (vqvae.py:238-241,301-302) and, with b = 0 and ref = forward output, their backward.
__global__ __launch_bounds__(256) void spec_affine_mask_kernel(const float *__restrict__ x, const float *__restrict__ ref,
                                                               float *__restrict__ y, int64_t HW4, float a0, float b0,
                                                               float a1, float b1, float thr, int use_mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index inside one channel plane
  const int64_t b = blockIdx.y;
  if (i >= HW4) return;
  const size_t o0 = ((size_t)b * 2) * HW4 + i, o1 = o0 + HW4;
  const float4 m = reinterpret_cast<const float4 *>(x)[o0], p = reinterpret_cast<const float4 *>(x)[o1];
  float4 ym = make_float4(a0 * m.x + b0, a0 * m.y + b0, a0 * m.z + b0, a0 * m.w + b0);
  float4 yp = make_float4(a1 * p.x + b1, a1 * p.y + b1, a1 * p.z + b1, a1 * p.w + b1);
  if (use_mask) {
    const float4 r = ref ? reinterpret_cast<const float4 *>(ref)[o0] : ym;
    yp.x = r.x <= thr ? 0.f : yp.x; yp.y = r.y <= thr ? 0.f : yp.y;
    yp.z = r.z <= thr ? 0.f : yp.z; yp.w = r.w <= thr ? 0.f : yp.w;
  }
  reinterpret_cast<float4 *>(y)[o0] = ym;
  reinterpret_cast<float4 *>(y)[o1] = yp;
}

int spec_affine_mask_f32(const float *x, const float *ref, float *y, int64_t B, int64_t HW, float a0, float b0, float a1,
                         float b1, float thr, int use_mask, hipStream_t st) {
  if (!x || !y || B <= 0 || HW <= 0 || B > 65535) return invalid("spec_affine_mask: bad argument");
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ref)) & 15))
    return invalid("spec_affine_mask: H*W must be a multiple of 4 and pointers 16-byte aligned");
  const int64_t HW4 = HW / 4;
  hipLaunchKernelGGL(spec_affine_mask_kernel, dim3((unsigned)((HW4 + 255) / 256), (unsigned)B), dim3(256), 0, st, x, ref, y,
                     HW4, a0, b0, a1, b1, thr, use_mask);
  return check_launch("spec_affine_mask");
}

// ---------------------------------------------------------------- multi-scale spectral loss (utils/losses/spectral.py:10-118)
// One scale: magnitudes |X| of the predicted and target STFTs (rows [B*T][RS], re block | im block of F bins each),
// the linear distance on |X| and the logarithmic one on log(|X| + eps).  The forward leaves, per (sample, row
// chunk), the four sums  sum |dm|, sum dm^2, sum |dl|, sum dl^2  (dm = |Xp| - |Xt|, dl = log difference): every
// reduction the reference's criteria need (L1 / MSE means, per-sample L2 norms) is a sum of these; partials are
// written per workgroup, no atomics.
__global__ __launch_bounds__(256) void spec_distance_fwd_kernel(const float *__restrict__ xp, const float *__restrict__ xt,
                                                                float *__restrict__ partial, int T, int F, int RS,
                                                                float eps, int rows_per_block) {
  __shared__ float red[4][4];
  const int b = blockIdx.y, t0 = blockIdx.x * rows_per_block, t1 = min(T, t0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int n = (t1 - t0) * F;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int t = t0 + i / F, f = i - (i / F) * F;
    const size_t o = ((size_t)b * T + t) * RS + f;
    const float pr = xp[o], pi = xp[o + F], tr = xt[o], ti = xt[o + F];
    const float mp = sqrtf(pr * pr + pi * pi), mt = sqrtf(tr * tr + ti * ti);
    const float dm = mp - mt, dl = logf(mp + eps) - logf(mt + eps);
    s0 += fabsf(dm); s1 += dm * dm; s2 += fabsf(dl); s3 += dl * dl;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
  }
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[((size_t)b * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// d loss / d Xp for  loss = sum_b [ clin[b] * g(dm) + clog[b] * g(dl) ],  g = |.| (kind 0) or (.)^2 / 2 ... precisely:
// kind 0: d/dm = sign(dm);  kind 1: d/dm = dm  (the caller folds every constant factor into clin / clog).
__global__ __launch_bounds__(256) void spec_distance_bwd_kernel(const float *__restrict__ xp, const float *__restrict__ xt,
                                                                float *__restrict__ dx, const float *__restrict__ clin,
                                                                const float *__restrict__ clog, int T, int F, int RS,
                                                                float eps, int kind) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= (int64_t)T * RS) return;
  const int t = (int)(i / RS), c = (int)(i - (int64_t)t * RS);
  const size_t row = ((size_t)b * T + t) * RS;
  float out = 0.f;
  if (c < 2 * F) {
    const int f = c < F ? c : c - F;
    const float pr = xp[row + f], pi = xp[row + F + f], tr = xt[row + f], ti = xt[row + F + f];
    const float mp = sqrtf(pr * pr + pi * pi), mt = sqrtf(tr * tr + ti * ti);
    const float dm = mp - mt, dl = logf(mp + eps) - logf(mt + eps);
    const float gm = kind == 0 ? (dm > 0.f ? 1.f : dm < 0.f ? -1.f : 0.f) : dm;
    const float gl = kind == 0 ? (dl > 0.f ? 1.f : dl < 0.f ? -1.f : 0.f) : dl;
    const float dmag = clin[b] * gm + clog[b] * gl / (mp + eps);
    out = mp > 0.f ? dmag * (c < F ? pr : pi) / mp : 0.f;
  }
  dx[row + c] = out;
}

int spec_distance_fwd_f32(const float *xp, const float *xt, float *partial, int B, int T, int F, int RS, float eps,
                          int rows_per_block, hipStream_t st) {
  if (!xp || !xt || !partial || B <= 0 || T <= 0 || F <= 0 || RS < 2 * F || rows_per_block <= 0 || B > 65535)
    return invalid("spec_distance_fwd: bad argument");
  hipLaunchKernelGGL(spec_distance_fwd_kernel, dim3((T + rows_per_block - 1) / rows_per_block, B), dim3(256), 0, st, xp, xt,
                     partial, T, F, RS, eps, rows_per_block);
  return check_launch("spec_distance_fwd");
}

int spec_distance_bwd_f32(const float *xp, const float *xt, float *dx, const float *clin, const float *clog, int B, int T,
                          int F, int RS, float eps, int kind, hipStream_t st) {
  if (!xp || !xt || !dx || !clin || !clog || B <= 0 || T <= 0 || F <= 0 || RS < 2 * F || B > 65535 || kind < 0 || kind > 1)
    return invalid("spec_distance_bwd: bad argument");
  const int64_t n = (int64_t)T * RS;
  hipLaunchKernelGGL(spec_distance_bwd_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, xp, xt, dx, clin, clog,
                     T, F, RS, eps, kind);
  return check_launch("spec_distance_bwd");
}

// ---------------------------------------------------------------- backward of the inverse front-end (to_audio)
// Adjoint of spec_to_stft_kernel: (d_a, d_ph) from d stft; a = mel ? power : magnitude (the forward's input).
__global__ void spec_to_stft_bwd_kernel(const float *__restrict__ a, const float *__restrict__ ph,
                                        const float *__restrict__ dx, float *__restrict__ da, float *__restrict__ dph,
                                        int64_t rows, int F, int mel) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t r = i / F;
  const int f = (int)(i - r * F);
  const float av = a[i];
  float mag = av, dmag_da = 1.f;
  if (mel) {
    const float pw = fmaxf(av, 0.f) + SPEC_EPS;
    mag = expf(0.5f * logf(pw));
    dmag_da = av > 0.f ? 0.5f * mag / pw : 0.f;
  }
  float sn, cs;
  sincosf(ph[i], &sn, &cs);
  const float gr = dx[r * 2 * F + f], gi = dx[r * 2 * F + F + f];
  da[i] = (gr * cs + gi * sn) * dmag_da;
  dph[i] = mag * (gi * cs - gr * sn);
}

// Adjoint of spec_inverse_prepare_kernel: d spec [B,2,F,T] from d_a, d_ph [B,T,F]:
//   d ch0 = d_a * exp(ch0) ;  d ch1 = pi * (sum over later frames of d_ph)   (reverse running sum, tile by tile)
__global__ __launch_bounds__(256) void spec_inverse_prepare_bwd_kernel(const float *__restrict__ spec,
                                                                       const float *__restrict__ da,
                                                                       const float *__restrict__ dph,
                                                                       float *__restrict__ dspec, int T, int F) {
  __shared__ float ta[32][33], tp[32][33];
  __shared__ float carry[32];
  const int f0 = blockIdx.x * 32, b = blockIdx.y;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (threadIdx.x < 32) carry[threadIdx.x] = 0.f;
  for (int t0 = ((T - 1) / 32) * 32; t0 >= 0; t0 -= 32) {
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {  // rows = frames, unit stride along f
      const int t = t0 + i, f = f0 + tx;
      float va = 0.f, vp = 0.f;
      if (t < T && f < F) {
        const size_t o = ((size_t)b * T + t) * F + f;
        va = da[o];
        vp = dph[o];
      }
      ta[tx][i] = va;   // [f][t]
      tp[tx][i] = vp;
    }
    __syncthreads();
    if (threadIdx.x < 32) {  // reverse scan of one frequency's 32 frames
      float run = carry[threadIdx.x];
      for (int j = 31; j >= 0; --j) {
        run += tp[threadIdx.x][j];
        tp[threadIdx.x][j] = run;
      }
      carry[threadIdx.x] = run;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {  // rows = frequencies, unit stride along t
      const int f = f0 + i, t = t0 + tx;
      if (f < F && t < T) {
        const size_t o0 = (((size_t)b * 2 + 0) * F + f) * T + t, o1 = (((size_t)b * 2 + 1) * F + f) * T + t;
        dspec[o0] = ta[i][tx] * expf(spec[o0]);
        dspec[o1] = tp[i][tx] * PI_F;
      }
    }
  }
}

int spec_to_stft_bwd_f32(const float *a, const float *ph, const float *dx, float *da, float *dph, int64_t rows, int F,
                         int mel, hipStream_t st) {
  if (!a || !ph || !dx || !da || !dph || rows <= 0 || F <= 0) return invalid("spec_to_stft_bwd: bad argument");
  const int64_t n = rows * F;
  hipLaunchKernelGGL(spec_to_stft_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, ph, dx, da, dph, rows,
                     F, mel);
  return check_launch("spec_to_stft_bwd");
}

int spec_inverse_prepare_bwd_f32(const float *spec, const float *da, const float *dph, float *dspec, int B, int T, int F,
                                 hipStream_t st) {
  if (!spec || !da || !dph || !dspec || B <= 0 || T <= 0 || F <= 0 || B > 65535)
    return invalid("spec_inverse_prepare_bwd: bad argument");
  hipLaunchKernelGGL(spec_inverse_prepare_bwd_kernel, dim3((F + 31) / 32, B), dim3(256), 0, st, spec, da, dph, dspec, T, F);
  return check_launch("spec_inverse_prepare_bwd");
}

int spec_polar_f32(const float *stft, float *a, float *ph, int B, int T, int F, int mel, hipStream_t st) {
  if (!stft || !a || !ph || B <= 0 || T <= 0 || F <= 0 || B > 65535) return invalid("spec_polar: bad argument");
  hipLaunchKernelGGL(spec_polar_kernel, dim3((F + 255) / 256, B), dim3(256), 0, st, stft, a, ph, T, F, mel);
  return check_launch("spec_polar");
}

int spec_finish_f32(const float *a, const float *ph, float *out, int B, int T, int F, int mel, hipStream_t st) {
  if (!a || !ph || !out || B <= 0 || T <= 0 || F <= 0 || B > 65535 || (T + 31) / 32 > 65535)
    return invalid("spec_finish: bad argument");
  hipLaunchKernelGGL(spec_finish_kernel, dim3((F + 31) / 32, (T + 31) / 32, B), dim3(256), 0, st, a, ph, out, T, F,
                     mel);
  return check_launch("spec_finish");
}

int spec_inverse_prepare_f32(const float *spec, float *a, float *ph, int B, int T, int F, hipStream_t st) {
  if (!spec || !a || !ph || B <= 0 || T <= 0 || F <= 0 || B > 65535) return invalid("spec_inverse_prepare: bad argument");
  hipLaunchKernelGGL(spec_inverse_prepare_kernel, dim3((F + 31) / 32, B), dim3(256), 0, st, spec, a, ph, T, F);
  return check_launch("spec_inverse_prepare");
}

int spec_to_stft_f32(const float *a, const float *ph, float *stft, int64_t rows, int F, int mel, hipStream_t st) {
  if (!a || !ph || !stft || rows <= 0 || F <= 0) return invalid("spec_to_stft: bad argument");
  const int64_t n = rows * F;
  hipLaunchKernelGGL(spec_to_stft_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, ph, stft, rows, F,
                     mel);
  return check_launch("spec_to_stft");
}

int overlap_add_f32(const float *frames, float *audio, int B, int T, int n_fft, int hop, int left, int64_t L,
                    hipStream_t st) {
  if (!frames || !audio || B <= 0 || T <= 0 || n_fft <= 0 || hop <= 0 || left < 0 || L <= 0 || B > 65535)
    return invalid("overlap_add: bad argument");
  hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((L + 255) / 256), B), dim3(256), 0, st, frames, audio, T,
                     n_fft, hop, left, L);
  return check_launch("overlap_add");
}

}  // namespace isi
