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
#include <cstdlib>

#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"
#include "prof.h"
#include "split_f16.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

// All fields are scalars so the whole struct stays in SGPRs (s_load from the
// kernarg segment); element strides / offsets are 32-bit (launchers reject
// tensors spanning 4 GiB or more, which also bounds the buffer descriptors).
struct ConvKArgs {
  const float *in0, *in1, *w, *bias, *res;
  float *out;
  unsigned in0_bytes, in1_bytes, w_bytes, out_bytes, res_bytes;
  int C0, Cin, src_uniform;  // src_uniform: a 32-wide K chunk never straddles the two sources
  int s0n, s0c, s0h, s0w;    // source 0 element strides
  int s1n, s1h, s1w;         // source 1 (channel stride 1)
  int rn, rc, rh, rw;        // residual strides (logical output coordinates)
  int on, oc, oh, ow;        // output strides, in units of GEMM-grid pixels
  int H, W, OH, OW, Cout, K, Kpad, KW, stride, relu, M;
  int bf16x3;                // opt-in split-bf16 products (flags & ISI_CONV_BF16X3)
  int pad;                   // plain convolution: symmetric zero padding
  int convT;                 // 1: blockIdx.z = phase (py,px) of ConvTranspose2d(k4,s2,p1)
  int w_phase_stride;        // convT: floats between two phase weight matrices
  int dst_sh, dst_sw;        // convT: strides of the full output tensor (phase offset)
  int in0_pair, in1_pair;    // ISI_CONV_IN0_PAIR / IN1_PAIR: the source holds split-f16 pairs (hi | lo << 16 per element)
  int out_pair;              // ISI_CONV_OUT_PAIR: write the output as split-f16 pairs
  const float *w16;          // ISI_CONV_W16: split-f16 pair copy of the packed weight (behind the fp32 one), or null
  int KH;                    // kernel height (K = KH * KW * Cin)
  int chunk_major;           // 1: K is walked slice-major -- for each 32-channel slice all KH x KW taps -- instead of
                             // tap-major (the packed weight keeps k = tap * Cin + c; only the visiting order changes)
  int nz;                    // plain conv, nz > 1: blockIdx.z = one of nz independent operand sets of the same shape
  int zs_in0, zs_w, zs_res, zs_out;   // element strides between two sets (source 0, packed weight, residual, output)
  const float *gate;         // optional, laid out exactly like the output: out = gate > 0 ? value : 0 (the ReLU mask of
                             // the layer's input applied in the epilogue of its input-gradient convolution)
  float *twin;               // optional fp32 copy of a pair-format output (training tape): the LDS-DMA kernel only
  int gate_pair;             // the gate tensor is in the pair format (ISI_CONV_GATE_PAIR): its hi piece decides
};

constexpr int LDK = 36;  // padded LDS row (floats): 144 B, 16-B aligned, bank-conflict free
// LDS stages: 2 = one barrier per K chunk; 1 = two barriers but half the LDS, i.e. more
// workgroups per CU.  Measured (profiles/): 1 stage is +14 % on the 128x64 tile (3 -> 5
// workgroups per CU), neutral-to-worse on 128x128.
template <int BN> constexpr int nbuf_for() { return BN == 64 ? 1 : 2; }
constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer_load beyond num_records returns 0: free zero padding

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return *reinterpret_cast<float4 *>(&v);
}

// ---- split-bf16 ("bf16x3") products: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
// a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// The dropped a_lo.b_lo term and the rounding of lo bound the relative error of every product
// by ~2^-16 (fp32: 2^-24); opt-in (ISI_CONV_BF16X3), never the default.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = 32;  // LDS row of a bf16 plane (elements): 64 B, unpadded; the four 16-B slots of a row are
                         // XOR-swizzled with (row >> 2) & 3: the 16-lane groups of ds_read_b128 then hit 16 distinct
                         // slots, and the contiguous 16-lane groups of ds_write_b64 cover 32 distinct banks
__device__ __forceinline__ int bf_slot(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 8; }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// hi = bf16(x) (round to nearest even: v_cvt_pk_bf16_f32), lo = bf16(x - hi): ~10 VALU per float4
__device__ __forceinline__ void split_bf16x4(const float4 v, uint2 &hi, uint2 &lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const bf16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}

// x = hi + mid + lo exactly (3 x 8 significand bits): the six products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi,
// mid.mid carry every term above 2^-24 of the result ("bf16x6": fp32-grade products at 6/16 of the fp32 pipe's time)
__device__ __forceinline__ void split3_bf16x4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
  const bf16x2 la = __builtin_convertvector(ra - __builtin_convertvector(ma, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(rb - __builtin_convertvector(mb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  mid = make_uint2(__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb));
  lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}

// ---- split-f16 ("f16x3", ISI_CONV_F16X3), pack-time weight pieces (ISI_CONV_W16) and activation pairs
// (ISI_CONV_*_PAIR): definitions and error / range analysis in split_f16.h
using f16s::f16x8;
constexpr float kF16ScaleA = f16s::kScaleA, kF16ScaleB = f16s::kScaleB, kF16Unscale = f16s::kUnscale;
__device__ __forceinline__ void split_f16x4(const float4 v, const float s, uint2 &hi, uint2 &lo) { f16s::split4(v, s, hi, lo); }

template <int BM, int BN, int WM, int WN, int MODE, int PREC = 0, bool OUTP = false>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvKArgs p) {
  constexpr bool BF = PREC >= 1;
  constexpr bool BF6 = PREC == 2;
  constexpr bool F16 = PREC >= 3;
  constexpr bool WPRE = PREC == 4;   // weights arrive as split-f16 pairs (ISI_CONV_W16): no conversion
  constexpr int TM = BM / WM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int RA = BM / 32;  // A rows staged per thread
  constexpr int RB = BN / 32;
  constexpr bool SCALAR_A = MODE == 2;  // element-wise gather loader
  constexpr bool DUAL = MODE == 1;      // quads of one chunk may come from either source
  constexpr int NBUF = BF ? 1 : nbuf_for<BN>();
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TM >= 1 && TN >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;                         // [NBUF][BM*LDK]
  float *Bs = smem + NBUF * BM * LDK;       // [NBUF][BN*LDK]
  // bf16x3: four bf16 planes [rows][LDB] (A hi, A lo, B hi, B lo), single stage
  unsigned short *Ahi = reinterpret_cast<unsigned short *>(smem);
  unsigned short *Alo = Ahi + BM * LDB;
  unsigned short *Bhi = Alo + BM * LDB;
  unsigned short *Blo = Bhi + BN * LDB;
  unsigned short *Ami = Blo + BN * LDB;     // bf16x6 only: the middle pieces
  unsigned short *Bmi = Ami + BM * LDB;
  int *row_b = BF ? reinterpret_cast<int *>(BF6 ? Bmi + BN * LDB : Blo + BN * LDB)
                  : reinterpret_cast<int *>(Bs + NBUF * BN * LDK);  // [BM] batch index or -1
  int *row_y = row_b + BM;
  int *row_x = row_y + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM);
  const int wn0 = (wave % WN) * (BN / WN);

  // phase of the transposed convolution (uniform)
  const int py = p.convT ? (int)(blockIdx.z >> 1) : 0;
  const int px = p.convT ? (int)(blockIdx.z & 1) : 0;
  const int pad_y = p.convT ? 1 - py : p.pad;
  const int pad_x = p.convT ? 1 - px : p.pad;
  const int w_off = p.convT ? (int)blockIdx.z * p.w_phase_stride : 0;
  const int out_off = p.convT ? py * p.dst_sh + px * p.dst_sw : 0;
  // batched plain convolution: operand set blockIdx.z
  const int zb = p.convT ? 0 : (int)blockIdx.z;
  const float *in0 = p.in0 + (size_t)zb * p.zs_in0;
  const float *wz = p.w + (size_t)zb * p.zs_w;

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

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in0), 0, p.in0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1), 0, p.in1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wz), 0, p.w_bytes, 0x00020000);

  // per staged A row: validity, top-left input coordinate, batch offsets (elements)
  int a_y0[RA], a_x0[RA], a_n0[RA], a_n1[RA];
  bool a_ok[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int r = lrow + 32 * j;
    const int b = row_b[r];
    a_ok[j] = b >= 0;
    a_y0[j] = row_y[r] * p.stride - pad_y;
    a_x0[j] = row_x[r] * p.stride - pad_x;
    a_n0[j] = b * p.s0n;
    a_n1[j] = b * p.s1n;
  }
  // per staged B row: byte offset of the packed weight row (OOB past Cout: zeros)
  unsigned b_off[RB], b_row[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + lrow + 32 * j;
    b_off[j] = n < p.Cout ? (unsigned)(w_off + n * p.Kpad + lq * 4) * 4u : OOB;
    b_row[j] = (unsigned)(w_off + n * p.Kpad) * 4u;
  }

  // this thread's position in K: tap (kh,kw) and channel c of its quad, advanced by 32 per chunk
  int kc_c, kc_kh, kc_kw;
  {
    const int kk = lq * 4;
    const int tap = kk / p.Cin;
    kc_c = kk - tap * p.Cin;
    kc_kh = tap / p.KW;
    kc_kw = tap - kc_kh * p.KW;
  }
  int kk_next = lq * 4;  // K index of the quad the next load_chunk() fetches

  float4 ra[RA], rb[RB];
  float4 ra1[DUAL ? RA : 1];  // second-source candidates (DUAL only)
  bool sel1 = false;
  bool chunk_pair = false;   // uniform: the chunk in flight comes from a pair-format source (split-f16 kernels)

  auto load_chunk = [&]() {
    if constexpr (!SCALAR_A) {
      const bool kvalid = kk_next < p.K;
      const bool second = kc_c >= p.C0;
      const int c = second ? kc_c - p.C0 : kc_c;
      if constexpr (!DUAL) {
        // the whole workgroup reads one source in this chunk: uniform descriptor
        const bool sec_u = __builtin_amdgcn_readfirstlane((int)second) != 0;
        chunk_pair = (sec_u ? p.in1_pair : p.in0_pair) != 0;
        const __amdgpu_buffer_rsrc_t rs = sec_u ? rs1 : rs0;
        const int sh = sec_u ? p.s1h : p.s0h, sw = sec_u ? p.s1w : p.s0w;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          const int iy = a_y0[j] + kc_kh, ix = a_x0[j] + kc_kw;
          const bool ok = kvalid && a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const int nb = sec_u ? a_n1[j] : a_n0[j];
          ra[j] = buf_load4(rs, ok ? (unsigned)(nb + iy * sh + ix * sw + c) * 4u : OOB);
        }
      } else {
        // one masked load per source; the pick happens when the chunk is written to LDS
        sel1 = second;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          const int iy = a_y0[j] + kc_kh, ix = a_x0[j] + kc_kw;
          const bool ok = kvalid && a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const unsigned o0 = (ok && !second) ? (unsigned)(a_n0[j] + iy * p.s0h + ix * p.s0w + c) * 4u : OOB;
          const unsigned o1 = (ok && second) ? (unsigned)(a_n1[j] + iy * p.s1h + ix * p.s1w + c) * 4u : OOB;
          ra[j] = buf_load4(rs0, o0);
          ra1[j] = buf_load4(rs1, o1);
        }
      }
    } else {
      // element-wise gather with arbitrary strides (NCHW sources, Cin % 4 != 0)
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k1 = kk_next + e;
          const int tap = k1 / p.Cin;
          const int c = k1 - tap * p.Cin;
          const int kh = tap / p.KW;
          const int kw = tap - kh * p.KW;
          const int iy = a_y0[j] + kh, ix = a_x0[j] + kw;
          const bool ok = k1 < p.K && a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          float t = 0.f;
          if (ok) {
            if (c < p.C0) t = in0[a_n0[j] + c * p.s0c + iy * p.s0h + ix * p.s0w];
            else t = p.in1[a_n1[j] + (c - p.C0) + iy * p.s1h + ix * p.s1w];
          }
          v[e] = t;
        }
        ra[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (p.chunk_major) {
      // weight quad of this thread: k = tap * Cin + c
      const unsigned koff = (unsigned)((kc_kh * p.KW + kc_kw) * p.Cin + kc_c) * 4u;
#pragma unroll
      for (int j = 0; j < RB; ++j) rb[j] = buf_load4(rsw, b_off[j] == OOB ? OOB : b_row[j] + koff);
      kk_next += kBK;
      // next tap of the same 32-channel slice; after the last tap, the next slice.  The nine shifted reads of a
      // slice are then adjacent in time on every workgroup of the XCD: their re-reads hit L2 instead of going to
      // the fabric (tap-major order re-reads a pixel's slice four chunks -- 6 MB of other traffic -- later)
      if (++kc_kw == p.KW) {
        kc_kw = 0;
        if (++kc_kh == p.KH) { kc_kh = 0; kc_c += kBK; }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      rb[j] = buf_load4(rsw, b_off[j]);
      if (b_off[j] != OOB) b_off[j] += kBK * 4u;
    }
    // advance this thread's K position by one chunk
    kk_next += kBK;
    kc_c += kBK;
    while (kc_c >= p.Cin) {
      kc_c -= p.Cin;
      if (++kc_kw == p.KW) { kc_kw = 0; ++kc_kh; }
    }
  };

  auto store_chunk = [&](int buf) {
    float *a = As + buf * BM * LDK;
    float *b = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      float4 v = ra[j];
      if constexpr (DUAL) v = sel1 ? ra1[j] : v;
      if constexpr (BF) {
        uint2 hi, mid, lo;
        if constexpr (WPRE) {
          if (chunk_pair) {
            // pair8 source: the 16 bytes this thread loaded ARE an operand fragment -- piece lq of the row's 128-byte
            // chunk = plane (lq & 1) of channel group (lq >> 1): one 16-byte LDS store, no arithmetic
            const int w16 = (lrow + 32 * j) * LDB + bf_slot(lrow + 32 * j, lq >> 1);
            *reinterpret_cast<float4 *>(((lq & 1) ? Alo : Ahi) + w16) = v;
            continue;
          }
        }
        if constexpr (BF6) split3_bf16x4(v, hi, mid, lo);
        else if constexpr (F16) split_f16x4(v, kF16ScaleA, hi, lo);
        else split_bf16x4(v, hi, lo);
        const int wo = (lrow + 32 * j) * LDB + bf_slot(lrow + 32 * j, lq >> 1) + (lq & 1) * 4;
        *reinterpret_cast<uint2 *>(Ahi + wo) = hi;
        *reinterpret_cast<uint2 *>(Alo + wo) = lo;
        if constexpr (BF6) *reinterpret_cast<uint2 *>(Ami + wo) = mid;
      } else {
        *reinterpret_cast<float4 *>(a + (lrow + 32 * j) * LDK + lq * 4) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if constexpr (BF) {
        uint2 hi, mid, lo;
        if constexpr (WPRE) {   // weights in the blocked pair format (ISI_CONV_W16): a 16-byte piece = one fragment
          const int w16 = (lrow + 32 * j) * LDB + bf_slot(lrow + 32 * j, lq >> 1);
          *reinterpret_cast<float4 *>(((lq & 1) ? Blo : Bhi) + w16) = rb[j];
          continue;
        }
        if constexpr (BF6) split3_bf16x4(rb[j], hi, mid, lo);
        else if constexpr (F16) split_f16x4(rb[j], kF16ScaleB, hi, lo);
        else split_bf16x4(rb[j], hi, lo);
        const int wo = (lrow + 32 * j) * LDB + bf_slot(lrow + 32 * j, lq >> 1) + (lq & 1) * 4;
        *reinterpret_cast<uint2 *>(Bhi + wo) = hi;
        *reinterpret_cast<uint2 *>(Blo + wo) = lo;
        if constexpr (BF6) *reinterpret_cast<uint2 *>(Bmi + wo) = mid;
      } else {
        *reinterpret_cast<float4 *>(b + (lrow + 32 * j) * LDK + lq * 4) = rb[j];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / kBK;
  load_chunk();
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fq = lane >> 5;

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = NBUF == 2 ? (kc & 1) : 0;
    if (kc + 1 < nk) load_chunk();  // global loads in flight under the MFMAs

    if constexpr (BF) {
      // 32x32x16 bf16 MFMA: lane (row = lane & 31, k-block = lane >> 5) holds 8 consecutive k
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        s16x8 ah[TM], al[TM], bh[TN], bl[TN], am[BF6 ? TM : 1], bm[BF6 ? TN : 1];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int off = (wm0 + i * 32 + frow) * LDB + bf_slot(wm0 + i * 32 + frow, s * 2 + fq);
          ah[i] = *reinterpret_cast<const s16x8 *>(Ahi + off);
          al[i] = *reinterpret_cast<const s16x8 *>(Alo + off);
          if constexpr (BF6) am[i] = *reinterpret_cast<const s16x8 *>(Ami + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int off = (wn0 + j * 32 + frow) * LDB + bf_slot(wn0 + j * 32 + frow, s * 2 + fq);
          bh[j] = *reinterpret_cast<const s16x8 *>(Bhi + off);
          bl[j] = *reinterpret_cast<const s16x8 *>(Blo + off);
          if constexpr (BF6) bm[j] = *reinterpret_cast<const s16x8 *>(Bmi + off);
        }
        if constexpr (BF6) {
          // smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
#pragma unroll
          for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const s16x8 av = t == 0 ? al[i] : (t == 2 || t == 3) ? am[i] : ah[i];
                const s16x8 bv = t == 1 ? bl[j] : (t == 2 || t == 4) ? bm[j] : bh[j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                    __builtin_bit_cast(bf16x8, bv), acc[i][j], 0, 0, 0);
              }
          }
        } else {
#pragma unroll
          for (int t = 0; t < 3; ++t) {  // lo terms first, the dominant hi.hi last
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const s16x8 av = t == 0 ? al[i] : ah[i];
                const s16x8 bv = t == 1 ? bl[j] : bh[j];
                if constexpr (F16)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av),
                                                                     __builtin_bit_cast(f16x8, bv), acc[i][j], 0, 0, 0);
                else
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                      __builtin_bit_cast(bf16x8, bv), acc[i][j], 0, 0, 0);
              }
          }
        }
      }
    }
    const float *a = As + buf * BM * LDK + (wm0 + frow) * LDK + fq * 4;
    const float *b = Bs + buf * BN * LDK + (wn0 + frow) * LDK + fq * 4;
#pragma unroll
    for (int s = 0; s < (BF ? 0 : 4); ++s) {
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
    if (NBUF == 1) __syncthreads();  // everyone has read the single stage before it is overwritten
    if (kc + 1 < nk) store_chunk(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: bias, residual, ReLU, store.  C layout of the 32x32 tile:
  // col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // Row offsets (elements) are prepared once in LDS (the A/B tiles are dead now);
  // invalid rows / columns get an out-of-range buffer offset, so loads and stores
  // are unconditional and all residual loads of a tile are in flight together.
  int *row_oo = reinterpret_cast<int *>(smem);  // [BM] output offset or -1
  int *row_ro = row_oo + BM;                    // [BM] residual offset
  if (tid < BM) {
    const int b = row_b[tid], oy = row_y[tid], ox = row_x[tid];
    row_oo[tid] = b < 0 ? -1 : out_off + b * p.on + oy * p.oh + ox * p.ow;
    row_ro[tid] = b < 0 ? 0 : b * p.rn + oy * p.rh + ox * p.rw;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)zb * p.zs_out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res + (size_t)zb * p.zs_res : in0), 0, p.res_bytes, 0x00020000);
  const bool has_res = p.res != nullptr;
  const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gate ? p.gate : in0), 0, p.gate ? p.out_bytes : 4u, 0x00020000);
  const bool has_gate = p.gate != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + frow;
    const bool nok = n < p.Cout;
    const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned oo[16];
      float res[16], gate[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
        const int o = row_oo[row];
        oo[r] = (nok && o >= 0) ? (unsigned)(o + n * p.oc) * 4u : OOB;
        res[r] = 0.f;
        gate[r] = 1.f;
        if (has_gate) {
          // ONE load form for both gate formats (a branch per element broke the batch of 16 loads in flight: the gated
          // kernels ran 45 % slower).  Pair8 storage: the hi piece of channel n sits 2 (n & 7) bytes into its 32-byte group;
          // the dword that holds it is fetched and the half selected below.
          const unsigned po = (oo[r] == OOB || !p.gate_pair) ? oo[r] : ((oo[r] - (unsigned)(n & 7) * 2u) & ~3u);
          gate[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsg, po, 0, 0));
        }
        if (has_res)
          res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                 rsr, oo[r] == OOB ? OOB : (unsigned)(row_ro[row] + n * p.rc) * 4u, 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = (F16 ? acc[i][j][r] * kF16Unscale : acc[i][j][r]) + bias + res[r];
        // fmaxf alone would turn a NaN (an operand beyond the range of ISI_CONV_F16X3, an fp32 overflow) into 0;
        // v - v is 0 for finite v and NaN otherwise.  (The select form `v < 0 ? 0 : v` makes this compiler allocate
        // 244 VGPRs for the split-f16 variant, 1 wave per SIMD.)
        if (p.relu) v = fmaxf(v, 0.f) + (v - v);
        if (has_gate) {
          // fp32 gate: positive value; pair gate: a rectified activation is positive <=> its f16 hi piece is a positive number
          const unsigned gw = __builtin_bit_cast(unsigned, gate[r]);
          const bool pass = p.gate_pair ? (short)((n & 1) ? (gw >> 16) : (gw & 0xffffu)) > 0 : gate[r] > 0.f;
          v = pass ? v : 0.f;
        }
        if constexpr (PREC == 4 && OUTP) {
          // pair-format output (split_f16.h: {hi[8] | lo[8]} per group of 8 channels).  A lane of this layout owns
          // ONE channel of 16 pixels, so its two pieces go out as 2-byte stores (conv_pair_f16.hip transposes
          // through LDS and writes whole groups; this path serves the launches that kernel does not take)
          const float t4 = v * kF16ScaleA;
          const _Float16 h = (_Float16)t4;
          const _Float16 l = (_Float16)(t4 - (float)h);
          const unsigned po = oo[r] == OOB ? OOB : oo[r] - (unsigned)(n & 7) * 4u + (unsigned)(n & 7) * 2u;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, h), rso, po, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, l), rso, po == OOB ? OOB : po + 16u, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rso, oo[r], 0, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int PREC>
constexpr size_t conv_smem_bytes() {
  if (PREC >= 1) return (size_t)((PREC == 2 ? 3 : 2) * (BM + BN) * LDB) * sizeof(unsigned short) + 3 * BM * sizeof(int);
  return (size_t)(nbuf_for<BN>() * BM * LDK + nbuf_for<BN>() * BN * LDK) * sizeof(float) + 3 * BM * sizeof(int);
}

template <int BM, int BN, int WM, int WN, int MODE, int PREC = 0, bool OUTP = false>
static int launch_cfg(const ConvKArgs &a, int nphase, hipStream_t stream) {
  auto kern = conv_igemm_f32_kernel<BM, BN, WM, WN, MODE, PREC, OUTP>;
  constexpr size_t smem = conv_smem_bytes<BM, BN, PREC>();
  static DeviceOnce attr_set;  // idempotent; racing threads set the same value
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return check_launch("hipFuncSetAttribute(conv_igemm)");
    attr_set.mark();
  }
  dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN, nphase > 1 ? nphase : std::max(a.nz, 1));
  {
    // algorithmic work: every MAC of the convolution once; input read once,
    // output written once, weights once (DESIGN.md "roofline accounting")
    const double np = nphase * (double)std::max(a.nz, 1);
    const double flops = 2.0 * a.M * np * a.Cout * a.K;
    const double in_px = nphase == 1 ? (double)a.M / (a.OH * a.OW) * a.H * a.W : (double)a.M;
    const double bytes = 4.0 * (in_px * a.Cin + (double)a.M * np * a.Cout * (a.res ? 2 : 1) +
                                np * a.Cout * a.K);
    const int kid = PREC >= 3 ? prof::K_CONV_F16X3 : PREC == 2 ? prof::K_CONV_BF16X6 : PREC == 1 ? prof::K_CONV_BF16X3 : MODE == 2 ? prof::K_CONV_GATHER
                             : (BN == 128 ? prof::K_CONV_128x128 : BN == 64 ? prof::K_CONV_128x64 : prof::K_CONV_128x32);
    prof::Scope scope(kid, flops, bytes, stream);
    ISI_PROF_LAUNCH(scope, kern, grid, dim3(256), smem, stream, a);
  }
  return check_launch("conv_igemm_f32");
}

static int launch_conv(const ConvKArgs &a_in, bool scalar_a, int nphase, hipStream_t stream) {
  ConvKArgs a = a_in;
  const int mode = scalar_a ? 2 : (a.src_uniform ? 0 : 1);
  const bool tap_major = knobs().conv_tap_major != 0;   // measurements
  a.chunk_major = (mode == 0 && a.KH * a.KW > 1 && a.C0 % kBK == 0 && a.Cin % kBK == 0 && !tap_major) ? 1 : 0;
  const bool two = a.Cin > a.C0;
  // the LDS-DMA kernel of the pair pipeline (conv_pair_f16.hip): pair8 sources, blocked weight pieces, whole
  // 64-column output tiles, channels-last output, no residual
  // It wins on the plain convolutions (1.13 - 1.2x, tools/bench_conv_pair.py); the transposed convolutions' four
  // phases (K = 4 Cin: 16 or fewer chunks per tile, its per-tile set-up and epilogue are not amortised) and K < 256
  // stay on this file's register-staged kernel.
  const bool dma_all = knobs().conv_pair_all != 0;
  const bool dma_shape = dma_all || (!a.convT && a.K >= 256);
  if (dma_shape && a.bf16x3 == 3 && a.w16 && mode == 0 && a.in0_pair && (!two || a.in1_pair) && !a.res && a.nz <= 1 && a.oc == 1 &&
      conv_pair_kernel_ok(a.C0, a.Cin - a.C0, a.Cout, a.KH * a.KW) && a.KH <= 4 && a.KW <= 4 && a.M < (1 << 24) &&
      a.in0_bytes < 0x70000000u && a.in1_bytes < 0x70000000u) {
    PairConvArgs c;
    memset(&c, 0, sizeof c);
    c.in0 = a.in0; c.in1 = two ? a.in1 : nullptr; c.w16 = a.w16; c.bias = a.bias; c.out = a.out;
    c.in0_bytes = a.in0_bytes; c.in1_bytes = a.in1_bytes; c.w_bytes = a.w_bytes; c.out_bytes = a.out_bytes;
    c.C0 = a.C0; c.C1 = a.Cin - a.C0;
    c.s0n = a.s0n; c.s0h = a.s0h; c.s0w = a.s0w; c.s1n = a.s1n; c.s1h = a.s1h; c.s1w = a.s1w;
    c.on = a.on; c.oh = a.oh; c.ow = a.ow;
    c.H = a.H; c.W = a.W; c.OH = a.OH; c.OW = a.OW; c.Cout = a.Cout; c.Kpad = a.Kpad; c.KH = a.KH; c.KW = a.KW;
    c.stride = a.stride; c.pad = a.pad; c.relu = a.relu; c.M = a.M;
    c.convT = a.convT; c.w_phase_stride = a.w_phase_stride; c.dst_sh = a.dst_sh; c.dst_sw = a.dst_sw;
    c.out_pair = a.out_pair; c.twin = a.twin;
    return conv_pair_f16(c, stream);
  }
  if (a.twin) return unsupported("conv: the fp32 twin of a pair-format output is written by the LDS-DMA kernel only "
                                 "(pair sources of 32-channel multiples, Cout % 64 == 0, K >= 256, dense channels-last output)");
  if ((a.in0_pair || a.in1_pair || a.out_pair) && !(a.bf16x3 == 3 && a.w16 && mode == 0 && a.Cout > 32 && a.K >= 128))
    return unsupported("conv: pair-format tensors need the split-f16 kernels (ISI_CONV_F16X3 | ISI_CONV_W16, "
                       "channels-last sources of 32-channel multiples, Cout > 32, K >= 128)");
  if (a.bf16x3 && mode == 0 && a.Cout <= 32 && a.K >= 128 && !a.out_pair && !a.in0_pair && !a.in1_pair) {
    // the residual blocks' 32-channel side (3x3 C -> 32 forward, 1x1 C -> 32 input gradient): 128 x 32 tiles, one
    // 32 x 32 accumulator tile per wave.  On the fp32 matrix pipe these layers ran AT its peak (148 of 157 TFLOP/s
    // for the 3x3 at B = 64) -- the split products have 5x that ceiling
    if (a.bf16x3 == 2) return launch_cfg<128, 32, 4, 1, 0, 2>(a, nphase, stream);
    if (a.bf16x3 == 3) return launch_cfg<128, 32, 4, 1, 0, 3>(a, nphase, stream);   // (pieces prepared at pack time: not built at this width)
    return launch_cfg<128, 32, 4, 1, 0, 1>(a, nphase, stream);
  }
  // (round 5: the three-term split-bf16 mode -- the training step's input gradients -- also takes K = 64..127: the 1x1
  // quantiser convolutions' input gradients (K = embed_dim = 64, 192 output channels at the bottom resolution) ran on the
  // fp32 matrix pipe, 142 us of a step)
  if (a.bf16x3 && mode == 0 && a.Cout > 32 && (a.K >= 128 || (a.bf16x3 == 1 && a.K >= 64 && !a.in0_pair && !a.in1_pair && !a.out_pair))) {
    // 128x64 tiles are ~20 % slower per FLOP than 128x128, but a GEMM that fills less than the chip's
    // 3 workgroups per CU with 128x128 tiles (the prior's d x d linears at 8 k rows: 260 tiles) finishes
    // sooner with twice as many, half as long, workgroups
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128) * nphase * (a.nz > 1 ? a.nz : 1);
    const bool narrow = a.Cout <= 64 || tiles128 < 640;
    if (a.bf16x3 == 2) {   // six-term split: fp32-grade products
      if (narrow) return launch_cfg<128, 64, 2, 2, 0, 2>(a, nphase, stream);
      return launch_cfg<128, 128, 2, 2, 0, 2>(a, nphase, stream);
    }
    if (a.bf16x3 == 3 && a.w16) {   // split-f16 with the weights' pieces prepared at pack time
      a.w = a.w16;
      if (a.out_pair) {
        if (a.oc != 1 || (a.Cout & 7)) return unsupported("conv: a pair-format output is channels-last with Cout % 8 == 0");
        if (narrow) return launch_cfg<128, 64, 2, 2, 0, 4, true>(a, nphase, stream);
        return launch_cfg<128, 128, 2, 2, 0, 4, true>(a, nphase, stream);
      }
      if (narrow) return launch_cfg<128, 64, 2, 2, 0, 4>(a, nphase, stream);
      return launch_cfg<128, 128, 2, 2, 0, 4>(a, nphase, stream);
    }
    if (a.bf16x3 == 3) {   // split-f16: fp32-grade products from three terms, f16 range
      if (narrow) return launch_cfg<128, 64, 2, 2, 0, 3>(a, nphase, stream);
      return launch_cfg<128, 128, 2, 2, 0, 3>(a, nphase, stream);
    }
    if (narrow) return launch_cfg<128, 64, 2, 2, 0, 1>(a, nphase, stream);
    return launch_cfg<128, 128, 2, 2, 0, 1>(a, nphase, stream);
  }
#define ISI_CONV_DISPATCH(MODE)                                                        \
  do {                                                                                  \
    if (a.Cout <= 32) return launch_cfg<128, 32, 4, 1, MODE>(a, nphase, stream);        \
    if (a.Cout <= 64) return launch_cfg<128, 64, 2, 2, MODE>(a, nphase, stream);        \
    return launch_cfg<128, 128, 2, 2, MODE>(a, nphase, stream);                         \
  } while (0)
  if (mode == 2) ISI_CONV_DISPATCH(2);
  if (mode == 1) ISI_CONV_DISPATCH(1);
  ISI_CONV_DISPATCH(0);
#undef ISI_CONV_DISPATCH
}

// flags -> 0 exact fp32 | 1 bf16x3 | 2 bf16x6 | 3 f16x3 (the most precise requested mode wins)
static int split_mode(int flags) {
  return (flags & ISI_CONV_BF16X6) ? 2 : (flags & ISI_CONV_F16X3) ? 3 : (flags & ISI_CONV_BF16X3) ? 1 : 0;
}

// Would a convolution of dense channels-last sources (C0 [+ C1] channels) run the split-f16 kernel that accepts
// pair-format sources?  (launch_conv's own conditions, for callers that plan tensor formats ahead: vqvae_run.cpp)
bool conv_pair_sources_ok(int C0, int C1, int Cout, int taps) {
  // the split-f16 kernels that read (and write) pair tensors: this file's (32-channel slices per source, Cout > 32,
  // K >= 128) and conv_pair_f16.hip (a subset of these shapes)
  return C0 > 0 && C0 % kBK == 0 && C1 % kBK == 0 && Cout > 32 && Cout % 8 == 0 && taps * (C0 + C1) >= 128;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Largest element offset reachable through a strided 4-D view, +1.
static int64_t extent(int64_t n, int64_t sn, int64_t c, int64_t sc, int64_t h, int64_t sh, int64_t w,
                      int64_t sw) {
  return (n - 1) * sn + (c - 1) * sc + (h - 1) * sh + (w - 1) * sw + 1;
}
constexpr int64_t kMaxElems = (int64_t)1 << 30;  // 4 GiB of fp32: 32-bit byte offsets

int conv2d_f32(const isi_src *s0, const isi_src *s1, const float *packed_w, const float *bias,
               const isi_src *res, const isi_dst *dst, int B, int H, int W, int Cout, int KH,
               int KW, int stride, int pad, int relu, hipStream_t stream, const float *gate, float *twin) {
  return conv2d_batched_f32(s0, s1, packed_w, bias, res, dst, B, H, W, Cout, KH, KW, stride, pad, relu, 1, 0, 0, 0, 0,
                            stream, gate, twin);
}

// nz independent convolutions of one shape in a single launch (grid z): set z reads source 0 at
// ptr + z * zs_in0, the packed weight at packed_w + z * zs_w, the residual at + z * zs_res and writes
// at dst + z * zs_out (element strides; the views describe one set).  One source only when nz > 1.
int conv2d_batched_f32(const isi_src *s0, const isi_src *s1, const float *packed_w, const float *bias,
                       const isi_src *res, const isi_dst *dst, int B, int H, int W, int Cout, int KH,
                       int KW, int stride, int pad, int relu, int nz, int64_t zs_in0, int64_t zs_w,
                       int64_t zs_res, int64_t zs_out, hipStream_t stream, const float *gate, float *twin) {
  if (nz < 1 || nz > 65535) return invalid("conv2d: bad batch count");
  if (twin && (gate || nz > 1 || !(relu & ISI_CONV_OUT_PAIR))) return unsupported("conv2d: an fp32 twin accompanies a single pair-format output");
  if (gate && (nz > 1 || (relu & (ISI_CONV_IN0_PAIR | ISI_CONV_IN1_PAIR | ISI_CONV_OUT_PAIR))))
    return unsupported("conv2d: the gated epilogue is for single fp32 launches");
  if (nz > 1 && s1 && s1->ptr) return unsupported("conv2d: batched launches take one source");
  if (nz > 1 && ((zs_in0 | zs_w | zs_res | zs_out) & 3)) return invalid("conv2d: batch strides must be multiples of 4 floats");
  if (!s0 || !s0->ptr || !packed_w || !dst || !dst->ptr) return invalid("conv2d: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return invalid("conv2d: bad shape");
  const int OH = (H + 2 * pad - KH) / stride + 1;
  const int OW = (W + 2 * pad - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return invalid("conv2d: empty output");
  if ((int64_t)B * OH * OW > INT32_MAX) return unsupported("conv2d: more than 2^31 output pixels");
  const bool two = s1 && s1->ptr;
  if (two && s1->sc != 1) return unsupported("conv2d: second source must be channels-last");
  const int64_t e0 = extent(B, s0->sn, s0->C, s0->sc, H, s0->sh, W, s0->sw);
  const int64_t e1 = two ? extent(B, s1->sn, s1->C, 1, H, s1->sh, W, s1->sw) : 1;
  const int64_t eo = extent(B, dst->sn, Cout, dst->sc, OH, dst->sh, OW, dst->sw);
  const int64_t er = (res && res->ptr) ? extent(B, res->sn, Cout, res->sc, OH, res->sh, OW, res->sw) : 1;
  if (e0 > kMaxElems || e1 > kMaxElems || eo > kMaxElems || er > kMaxElems)
    return unsupported("conv2d: a tensor spans 4 GiB or more");
  if ((!gate || (!(relu & (ISI_CONV_OUT_PAIR | ISI_CONV_GATE_PAIR)) && aligned16(gate))) && e0 <= kMaxElems && eo <= kMaxElems &&
      aligned16(packed_w) && conv_first_applicable(s0, s1, res, dst, Cout, KH, KW, stride, pad, OH, OW, nz)) {
    // the 2-channel first layer has its own HBM-oriented kernel (conv_first_f32.hip), bit-identical results
    if (relu & (ISI_CONV_IN0_PAIR | ISI_CONV_IN1_PAIR)) return unsupported("conv2d: pair-format source on the 2-channel layer");
    return conv_first_f32(s0, packed_w, bias, dst, B, H, W, Cout, OH, OW, e0, relu, stream, twin, gate);
  }
  // the prior's linear layers: rows of a dense matrix, three-term products -> the GEMM kernel (gemm_split_f32.hip)
  if (nz == 1 && !(gate && (relu & ISI_CONV_GATE_PAIR)) && !two && KH == 1 && KW == 1 && stride == 1 && pad == 0 && B == 1 && H == 1 && s0->sc == 1 &&
      dst->sc == 1 && (!res || !res->ptr || res->sc == 1) && !(relu & (ISI_CONV_IN0_PAIR | ISI_CONV_IN1_PAIR | ISI_CONV_OUT_PAIR)) &&
      gemm_split_applicable(W, Cout, s0->C, split_mode(relu)) && aligned16(s0->ptr) && aligned16(packed_w) && s0->sw % 4 == 0 &&
      !knobs().no_gemm_kernel) {
    GemmExtra gx;
    memset(&gx, 0, sizeof gx);
    gx.nz = 1; gx.gate = gate; gx.ldg = dst->sw;          // (a gate is laid out like dst)
    return gemm_split_f32(s0->ptr, s0->sw, packed_w, bias, (res && res->ptr) ? res->ptr : nullptr, (res && res->ptr) ? res->sw : 0,
                          dst->ptr, dst->sw, W, Cout, s0->C, relu & 1, split_mode(relu), stream,
                          ((split_mode(relu) == 3 && (relu & ISI_CONV_W16)) || (split_mode(relu) == 1 && (relu & ISI_CONV_W16_BF16)))
                              ? packed_w + (size_t)Cout * round_up((size_t)s0->C, kBK) : nullptr,
                          gate ? &gx : nullptr);
  }
  const int64_t zmax = std::max(std::max(zs_in0, zs_w), std::max(zs_res, zs_out));
  if (zmax < 0 || zmax >= ((int64_t)1 << 31)) return unsupported("conv2d: batch stride out of range");
  ConvKArgs a;
  memset(&a, 0, sizeof a);
  a.nz = nz; a.zs_in0 = (int)zs_in0; a.zs_w = (int)zs_w; a.zs_res = (int)zs_res; a.zs_out = (int)zs_out;
  a.in0 = s0->ptr; a.C0 = s0->C; a.in0_bytes = (unsigned)(e0 * 4);
  a.s0n = (int)s0->sn; a.s0c = (int)s0->sc; a.s0h = (int)s0->sh; a.s0w = (int)s0->sw;
  a.in1 = two ? s1->ptr : s0->ptr;
  a.in1_bytes = two ? (unsigned)(e1 * 4) : a.in0_bytes;
  const int C1 = two ? s1->C : 0;
  if (two) { a.s1n = (int)s1->sn; a.s1h = (int)s1->sh; a.s1w = (int)s1->sw; }
  a.Cin = a.C0 + C1;
  a.src_uniform = (!two || (a.C0 % kBK == 0 && C1 % kBK == 0)) ? 1 : 0;
  a.w = packed_w; a.bias = bias; a.gate = gate; a.twin = twin; a.gate_pair = (gate && (relu & ISI_CONV_GATE_PAIR)) ? 1 : 0;
  a.res = (res && res->ptr) ? res->ptr : nullptr;
  if (a.res) { a.rn = (int)res->sn; a.rc = (int)res->sc; a.rh = (int)res->sh; a.rw = (int)res->sw; }
  a.out = dst->ptr; a.on = (int)dst->sn; a.oc = (int)dst->sc; a.oh = (int)dst->sh; a.ow = (int)dst->sw;
  a.out_bytes = (unsigned)(eo * 4); a.res_bytes = (unsigned)(er * 4);
  a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.Cout = Cout;
  a.K = KH * KW * a.Cin; a.Kpad = (int)round_up(a.K, kBK);
  a.w_bytes = (unsigned)((size_t)Cout * a.Kpad * 4);
  a.KW = KW; a.stride = stride; a.relu = relu & 1; a.bf16x3 = split_mode(relu); a.M = B * OH * OW;
  a.KH = KH;
  a.in0_pair = (relu & ISI_CONV_IN0_PAIR) ? 1 : 0; a.in1_pair = (two && (relu & ISI_CONV_IN1_PAIR)) ? 1 : 0;
  a.out_pair = (relu & ISI_CONV_OUT_PAIR) ? 1 : 0;
  if (a.res && (a.in0_pair || a.in1_pair || a.out_pair)) return unsupported("conv2d: pair formats with a residual input");
  a.w16 = ((relu & ISI_CONV_W16) && nz == 1) ? packed_w + (size_t)Cout * a.Kpad : nullptr;
  a.pad = pad; a.convT = 0;
  bool vec = s0->sc == 1 && (a.C0 % 4 == 0) && (C1 % 4 == 0) && aligned16(s0->ptr) &&
             (s0->sn % 4 == 0) && (s0->sh % 4 == 0) && (s0->sw % 4 == 0);
  if (two) vec = vec && aligned16(s1->ptr) && (s1->sn % 4 == 0) && (s1->sh % 4 == 0) && (s1->sw % 4 == 0);
  if (!aligned16(packed_w)) return invalid("conv2d: packed weight must be 16-byte aligned");
  if (twin && !(dst->sc == 1 && dst->sw == Cout && dst->sh == (int64_t)OW * Cout && (B == 1 || dst->sn == (int64_t)OH * OW * Cout) &&
                aligned16(twin)))
    return unsupported("conv2d: the fp32 twin needs a dense channels-last output");
  return launch_conv(a, !vec, 1, stream);
}

int conv_transpose2d_k4s2_f32(const isi_src *s, const float *packed_w, const float *bias,
                              const isi_dst *dst, int B, int H, int W, int Cout, int relu,
                              hipStream_t stream, const float *gate, float *twin) {
  if (!s || !s->ptr || !packed_w || !dst || !dst->ptr) return invalid("convT: null pointer");
  if (twin && (gate || !(relu & ISI_CONV_OUT_PAIR) || !(relu & ISI_CONV_IN0_PAIR)))
    return unsupported("convT: an fp32 twin accompanies a pair-format output of the pair kernel");
  if (gate && (convT_small_applicable(s->C, Cout) || (relu & (ISI_CONV_IN0_PAIR | ISI_CONV_OUT_PAIR))))
    return unsupported("convT: the gated epilogue is for the fp32 implicit-GEMM launches");
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return invalid("convT: bad shape");
  if ((int64_t)B * H * W > INT32_MAX) return unsupported("convT: more than 2^31 pixels per phase");
  const int64_t e0 = extent(B, s->sn, s->C, s->sc, H, s->sh, W, s->sw);
  const int64_t eo = extent(B, dst->sn, Cout, dst->sc, 2 * H, dst->sh, 2 * W, dst->sw);
  if (e0 > kMaxElems || eo > kMaxElems) return unsupported("convT: a tensor spans 4 GiB or more");
  if (convT_small_applicable(s->C, Cout)) {
    if (relu & ISI_CONV_IN0_PAIR) {
      // pair-format input: the LDS-DMA form of the few-channel kernel (dense channels-last source, fp32 output)
      const int Cin = s->C;
      const bool dense_in = s->sc == 1 && (W == 1 || s->sw == Cin) && (H == 1 || s->sh == (int64_t)W * Cin) &&
                            (B == 1 || s->sn == (int64_t)H * W * Cin);
      if ((relu & ISI_CONV_OUT_PAIR) || !(relu & ISI_CONV_W16) || split_mode(relu) != 3 || !dense_in ||
          !convT_small_pair_ok(Cin, Cout) || !aligned16(s->ptr) || !aligned16(packed_w))
        return unsupported("convT: pair-format input on the few-channel kernel needs ISI_CONV_F16X3 | ISI_CONV_W16, "
                           "64 dense channels-last input channels, Cout <= 2 and an fp32 output");
      return convT_k4s2_small_pair_f16(s->ptr, packed_w, bias, dst->ptr, B, H, W, Cin, Cout, (int)dst->sn, (int)dst->sc,
                                       (int)dst->sh, (int)dst->sw, relu & ISI_CONV_RELU, stream);
    }
    if (relu & ISI_CONV_OUT_PAIR) return unsupported("convT: pair-format output on the few-channel kernel");
    // few output channels: GEMM + col2im gather kernel (weights were packed in its layout)
    return convT_k4s2_small_f32(s->ptr, packed_w, bias, dst->ptr, B, H, W, s->C, Cout, e0, (int)s->sn,
                                (int)s->sc, (int)s->sh, (int)s->sw, (int)dst->sn, (int)dst->sc,
                                (int)dst->sh, (int)dst->sw, relu & ISI_CONV_RELU, stream);
  }
  {
    // pair pipeline: the fused-phase LDS-DMA kernel (convT_pair_f16.hip) on dense channels-last tensors
    const int Cin = s->C;
    // (the stride of a dimension of extent 1 is never used: torch leaves arbitrary values there)
    const bool dense_in = s->sc == 1 && (W == 1 || s->sw == Cin) && (H == 1 || s->sh == (int64_t)W * Cin) &&
                          (B == 1 || s->sn == (int64_t)H * W * Cin);
    const bool dense_out = dst->sc == 1 && dst->sw == Cout && dst->sh == (int64_t)2 * W * Cout &&
                           (B == 1 || dst->sn == (int64_t)4 * H * W * Cout);
    if ((relu & ISI_CONV_IN0_PAIR) && (relu & ISI_CONV_W16) && split_mode(relu) == 3 && dense_in && dense_out &&
        convT_pair_ok(Cin, Cout) && aligned16(s->ptr) && aligned16(dst->ptr) && aligned16(packed_w)) {
      const size_t Kpad = round_up((size_t)4 * Cin, kBK);
      return convT_pair_f16(s->ptr, packed_w + (size_t)4 * Cout * Kpad, bias, dst->ptr, B, H, W, Cin, Cout, relu & 1,
                            (relu & ISI_CONV_OUT_PAIR) ? 1 : 0, stream, nullptr, 0, twin);
    }
  }
  if (twin) return unsupported("convT: the fp32 twin is written by the fused pair kernel only (dense tensors, Cin % 16 == 0, Cout % 64 == 0)");
  ConvKArgs a;
  memset(&a, 0, sizeof a);
  a.in0 = s->ptr; a.in1 = s->ptr; a.C0 = s->C; a.Cin = s->C; a.src_uniform = 1;
  a.in0_bytes = a.in1_bytes = (unsigned)(e0 * 4);
  a.s0n = (int)s->sn; a.s0c = (int)s->sc; a.s0h = (int)s->sh; a.s0w = (int)s->sw;
  a.w = packed_w; a.bias = bias; a.res = nullptr; a.gate = gate; a.gate_pair = (gate && (relu & ISI_CONV_GATE_PAIR)) ? 1 : 0;
  a.out = dst->ptr;
  // GEMM-grid pixel (m_y, m_x) of phase (py,px) is output pixel (2 m_y + py, 2 m_x + px)
  a.on = (int)dst->sn; a.oc = (int)dst->sc; a.oh = (int)(2 * dst->sh); a.ow = (int)(2 * dst->sw);
  a.dst_sh = (int)dst->sh; a.dst_sw = (int)dst->sw;
  a.out_bytes = (unsigned)(eo * 4); a.res_bytes = 4;
  a.H = H; a.W = W; a.OH = H; a.OW = W; a.Cout = Cout;
  a.K = 4 * a.Cin; a.Kpad = (int)round_up(a.K, kBK);
  a.w_phase_stride = Cout * a.Kpad;
  a.w_bytes = (unsigned)((size_t)4 * Cout * a.Kpad * 4);
  a.KW = 2; a.stride = 1; a.relu = relu & 1; a.bf16x3 = split_mode(relu); a.M = B * H * W;
  a.KH = 2;
  a.in0_pair = (relu & ISI_CONV_IN0_PAIR) ? 1 : 0; a.out_pair = (relu & ISI_CONV_OUT_PAIR) ? 1 : 0;
  a.w16 = (relu & ISI_CONV_W16) ? packed_w + (size_t)4 * Cout * a.Kpad : nullptr;
  a.convT = 1;
  const bool vec = s->sc == 1 && (a.C0 % 4 == 0) && aligned16(s->ptr) && (s->sn % 4 == 0) &&
                   (s->sh % 4 == 0) && (s->sw % 4 == 0);
  if (!aligned16(packed_w)) return invalid("convT: packed weight must be 16-byte aligned");
  return launch_conv(a, !vec, 4, stream);
}

}  // namespace isi
