// Fused RosinalityResBlock for gfx950 (exact-fp32 matrix pipe).
//
//   out = [relu]( r + W2 * relu(W1 (*) r + b1) + b2 )        r = rectified input
//
// i.e. reference vqvae/encoder_decoder.py:22-35 with its in-place first ReLU
// already applied by the producer of r.  One launch replaces the 3x3 conv
// (C -> R), the ReLU, the 1x1 conv (R -> C) and the residual add; the R-channel
// intermediate never leaves the CU:
//
//   GEMM1  [128 px x 9C] x [9C x 32]   implicit GEMM over the 3x3 taps, K chunks
//          of 32 channels staged global -> registers -> LDS (double buffered);
//          each of the 4 waves owns 32 pixels x 32 hidden channels
//   h      = relu(acc1 + b1) written to the wave's own LDS rows
//   GEMM2  [32 px x 32] x [32 x C] per wave, W2 resident in LDS, then
//          + b2 + r (centre pixel, re-read from L2), ReLU, store.
//
// Requirements (else the caller uses two isi_conv2d_f32 launches): channels-last
// dense input/output, C % 32 == 0, C <= 128, R <= 32.
#include "isi_common.h"
#include "prof.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ResKArgs {
  const float *in, *w1, *b1, *w2, *b2;
  float *out;
  unsigned in_bytes, w1_bytes, w2_bytes;
  int C, R, H, W, M, K1pad, relu;
};

namespace {
constexpr int LDK = 36;
constexpr int BM = 128;
constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}
__device__ __forceinline__ float elem(const float4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}
}  // namespace

template <int TC>  // TC = C / 32 output-channel tiles of GEMM2
__global__ __launch_bounds__(256) void resblock_f32_kernel(const ResKArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;                       // [2][BM*LDK]   (reused as h after GEMM1)
  float *Bs = As + 2 * BM * LDK;          // [2][32*LDK]   W1 chunk
  float *W2s = Bs + 2 * 32 * LDK;         // [C][LDK]      W2, k = hidden channel
  int *row_b = reinterpret_cast<int *>(W2s + TC * 32 * LDK);  // [BM]
  int *row_y = row_b + BM;
  int *row_x = row_y + BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = TC * 32;

  const int ntile = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntile / 8, r = ntile % 8, xcd = bid % 8, idx = bid / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = tile * BM;

  if (tid < BM) {
    const int m = m0 + tid;
    int b = -1, oy = 0, ox = 0;
    if (m < p.M) {
      b = m / (p.H * p.W);
      const int rem = m - b * (p.H * p.W);
      oy = rem / p.W;
      ox = rem - oy * p.W;
    }
    row_b[tid] = b; row_y[tid] = oy; row_x[tid] = ox;
  }

  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w2), 0, p.w2_bytes, 0x00020000);

  const int lrow = tid >> 3, lq = tid & 7;
  // W2 -> LDS: packed [C][32] (hidden channels beyond R are zero-padded by the packer)
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int n = lrow + 32 * j;
    *reinterpret_cast<float4 *>(W2s + n * LDK + lq * 4) = buf_load4(rs2, (unsigned)(n * 32 + lq * 4) * 4u);
  }
  __syncthreads();

  int a_y[4], a_x[4], a_n[4];
  bool a_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = lrow + 32 * j;
    const int b = row_b[r];
    a_ok[j] = b >= 0;
    a_y[j] = row_y[r] - 1;
    a_x[j] = row_x[r] - 1;
    a_n[j] = b * p.H * p.W * C;
  }
  const unsigned b_off = lrow < p.R ? (unsigned)(lrow * p.K1pad + lq * 4) * 4u : OOB;

  float4 ra[4], rb;
  const int nk = 9 * TC;  // K chunks: taps outer, 32-channel groups inner
  auto load_chunk = [&](int kc) {
    const int tap = kc / TC, c0 = (kc - tap * TC) * 32;
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = a_y[j] + kh, ix = a_x[j] + kw;
      const bool ok = a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      ra[j] = buf_load4(rsi, ok ? (unsigned)(a_n[j] + (iy * p.W + ix) * C + c0 + lq * 4) * 4u : OOB);
    }
    rb = buf_load4(rs1, b_off == OOB ? OOB : b_off + (unsigned)kc * 128u);
  };
  auto store_chunk = [&](int buf) {
    float *a = As + buf * BM * LDK;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4 *>(a + (lrow + 32 * j) * LDK + lq * 4) = ra[j];
    *reinterpret_cast<float4 *>(Bs + buf * 32 * LDK + lrow * LDK + lq * 4) = rb;
  };

  f32x16 acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 31, fq = lane >> 5;
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) load_chunk(kc + 1);
    const float *a = As + buf * BM * LDK + (wave * 32 + frow) * LDK + fq * 4;
    const float *b = Bs + buf * 32 * LDK + frow * LDK + fq * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 af = *reinterpret_cast<const float4 *>(a + s * 8);
      const float4 bf = *reinterpret_cast<const float4 *>(b + s * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(af, e), elem(bf, e), acc1, 0, 0, 0);
    }
    if (kc + 1 < nk) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- h = relu(acc1 + b1) -> this wave's rows of hs (aliases the dead A buffers)
  float *hs = As + wave * 32 * LDK;
  {
    const float b1 = frow < p.R ? p.b1[frow] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fq;
      hs[row * LDK + frow] = frow < p.R ? fmaxf(acc1[r] + b1, 0.f) : 0.f;
    }
  }
  __syncthreads();

  // ---- GEMM2: [32 px x 32] x [32 x C]
  f32x16 acc2[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 af = *reinterpret_cast<const float4 *>(hs + frow * LDK + fq * 4 + s * 8);
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const float4 bf = *reinterpret_cast<const float4 *>(W2s + (j * 32 + frow) * LDK + fq * 4 + s * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(elem(af, e), elem(bf, e), acc2[j], 0, 0, 0);
    }
  }

  // ---- epilogue: + b2 + r, ReLU, store (dense channels-last: offset = m*C + n)
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int n = j * 32 + frow;
    const float b2 = p.b2[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
      const int m = m0 + row;
      if (m >= p.M) continue;
      float v = acc2[j][r] + b2 + p.in[(size_t)m * C + n];
      if (p.relu) v = fmaxf(v, 0.f);
      p.out[(size_t)m * C + n] = v;
    }
  }
}

template <int TC>
static int launch_res(const ResKArgs &a, hipStream_t stream) {
  auto kern = resblock_f32_kernel<TC>;
  constexpr size_t smem = (size_t)(2 * BM * LDK + 2 * 32 * LDK + TC * 32 * LDK) * sizeof(float) + 3 * BM * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(resblock)");
    attr_set = true;
  }
  const double C = a.C, R = a.R, M = a.M;
  prof::Scope scope(prof::K_RESBLOCK, 2.0 * M * R * 9 * C + 2.0 * M * C * R,
                    4.0 * (3.0 * M * C + 10.0 * C * R), stream);
  hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(256), smem, stream, a);
  return check_launch("resblock_f32");
}

bool resblock_fusable(int C, int R) { return C % 32 == 0 && C >= 32 && C <= 128 && R >= 1 && R <= 32; }

// in/out: dense channels-last [B,H,W,C].  w1: packed 3x3 weight [R][9C]; w2: packed 1x1 weight [C][32].
int resblock_f32(const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                 float *out, int B, int H, int W, int C, int R, int relu, hipStream_t stream) {
  if (!in || !w1 || !b1 || !w2 || !b2 || !out) return invalid("resblock: null pointer");
  if (B <= 0 || H <= 0 || W <= 0) return invalid("resblock: bad shape");
  if (!resblock_fusable(C, R)) return unsupported("resblock: need C % 32 == 0, C <= 128, R <= 32");
  const int64_t elems = (int64_t)B * H * W * C;
  if (elems > ((int64_t)1 << 30)) return unsupported("resblock: tensor spans 4 GiB or more");
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15)
    return invalid("resblock: pointers must be 16-byte aligned");
  ResKArgs a;
  a.in = in; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.out = out;
  a.in_bytes = (unsigned)(elems * 4);
  a.K1pad = 9 * C;  // multiple of 32 because C is
  a.w1_bytes = (unsigned)((size_t)R * a.K1pad * 4);
  a.w2_bytes = (unsigned)((size_t)C * 32 * 4);
  a.C = C; a.R = R; a.H = H; a.W = W; a.M = B * H * W; a.relu = relu;
  switch (C / 32) {
    case 1: return launch_res<1>(a, stream);
    case 2: return launch_res<2>(a, stream);
    case 3: return launch_res<3>(a, stream);
    default: return launch_res<4>(a, stream);
  }
}

}  // namespace isi
