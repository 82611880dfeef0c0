// L2 nearest-neighbour vector quantiser for gfx950 (eval mode).
//
// Restates QuantizedBottleneck.forward (reference vqvae/bottleneck.py:53-61,
// 75-77,94-101) without materialising the [N,K] distance and one-hot matrices:
//
//   d[n,k]  = (|z_n|^2 - 2 z_n.e_k) + |e_k|^2          (fp32, same association)
//   idx[n]  = first k attaining min_k d[n,k]            ((-dist).max(1), ties -> lowest)
//   q[n]    = z_n + (e_idx - z_n)                        (straight-through value)
//   sse    += sum (e_idx - z_n)^2 ;  counts[idx[n]] += 1
//   (a z_n holding NaN / Inf: idx = -1, q = NaN, sse = NaN, not counted)
//
// The whole codebook ([K][D+4] fp32, 136 KiB at K=512, D=64) stays in LDS for
// the lifetime of a persistent workgroup.  Each wave handles 32 vectors at a
// time: z.e_k runs on the exact-fp32 matrix pipe with the operands swapped
// (codes = MFMA rows, vectors = MFMA columns) so that the 16 accumulators of a
// lane all belong to ONE vector and the running arg-min is lane-local; the two
// half-waves (lanes l and l+32 hold interleaved code rows of the same vector)
// are merged with a single cross-lane exchange at the end.
#include "isi_common.h"
#include "knobs.h"
#include "prof.h"
#include "split_f16.h"

namespace isi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8v __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

#ifndef ISI_VQ_WAVES
#define ISI_VQ_WAVES 8
#endif
constexpr int VQ_BLOCK = 64 * ISI_VQ_WAVES;
constexpr int VQ_VEC_PER_BLOCK_ITER = 32 * ISI_VQ_WAVES;  // each wave owns 32 vectors per pass

template <int D>
__global__ __launch_bounds__(VQ_BLOCK) void vq_nearest_kernel(
    const float *__restrict__ z, const float *__restrict__ codes, const float *__restrict__ e2g,
    int64_t *__restrict__ idx_out, float *__restrict__ q_out, int32_t *__restrict__ counts,
    float *__restrict__ sse_part, int64_t N, int K) {
  constexpr int LDD = D + 4;  // padded row: conflict-free b128 fragment reads
  constexpr int NQ = D / 8;   // k-quads held per lane (half-waves interleave quads)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // K need not be a multiple of the 32-code MFMA tile: the LDS image is padded to Kp rows of zeros whose |e|^2 is
  // +inf, so a padding row's distance is +inf and never wins (`d < best` with best starting at +inf)
  const int Kp = (K + 31) & ~31;
  float *cb = smem;            // [Kp][LDD]
  float *e2 = smem + (size_t)Kp * LDD;  // [Kp]
  float *red = e2 + Kp;        // [waves] per-wave sse
  int *hist = reinterpret_cast<int *>(red + ISI_VQ_WAVES);  // [Kp] workgroup histogram (flushed once at the end)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;   // vector (MFMA column) handled by this lane
  const int half = lane >> 5;

  // ---- codebook -> LDS (once per persistent workgroup)
  for (int i = tid; i < Kp * (D / 4); i += VQ_BLOCK) {
    const int k = i / (D / 4), qd = i - k * (D / 4);
    *reinterpret_cast<float4 *>(cb + (size_t)k * LDD + qd * 4) =
        k < K ? *reinterpret_cast<const float4 *>(codes + (size_t)k * D + qd * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < Kp; i += VQ_BLOCK) { e2[i] = i < K ? e2g[i] : INFINITY; hist[i] = 0; }
  __syncthreads();

  float sse = 0.f;
  const int64_t n_iter = (N + VQ_VEC_PER_BLOCK_ITER - 1) / VQ_VEC_PER_BLOCK_ITER;
  for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const int64_t n = it * VQ_VEC_PER_BLOCK_ITER + wave * 32 + col;
    const bool valid = n < N;
    // this lane's quads of z_n: quad index 2*j + half
    float4 zq[NQ];
    float x2p = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) v = *reinterpret_cast<const float4 *>(z + n * D + (2 * j + half) * 4);
      zq[j] = v;
      x2p += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    const float x2 = x2p + __shfl_xor(x2p, 32);

    float best = INFINITY;
    int besti = 0;
    for (int kt = 0; kt < Kp; kt += 32) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float *arow = cb + (size_t)(kt + col) * LDD + half * 4;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + j * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, zq[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, zq[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, zq[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, zq[j].w, acc, 0, 0, 0);
      }
      // rows (codes) of this lane, ascending: (r&3) + 8*(r>>2) + 4*half
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int code = kt + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float d = (x2 - 2.f * acc[r]) + e2[code];
        if (d < best) { best = d; besti = code; }
      }
    }
    // merge the two half-waves (same vector, disjoint code rows)
    {
      const float ob = __shfl_xor(best, 32);
      const int oi = __shfl_xor(besti, 32);
      if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    // a vector with a NaN / Inf component has no nearest code (every comparison above fails): index -1, so that a
    // range violation upstream (ISI_CONV_F16X3 operands beyond f16, or an overflow in fp32) is never mistaken for
    // code 0; its quantised value and the squared error are NaN
    const bool lost = !(best < INFINITY);
    if (valid && lost) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        *reinterpret_cast<float4 *>(q_out + n * D + (2 * j + half) * 4) = make_float4(NAN, NAN, NAN, NAN);
      sse = NAN;
      if (half == 0) idx_out[n] = -1;
    } else if (valid) {
      const float *crow = cb + (size_t)besti * LDD + half * 4;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float4 e = *reinterpret_cast<const float4 *>(crow + j * 8);
        const float4 v = zq[j];
        float4 dq = make_float4(e.x - v.x, e.y - v.y, e.z - v.z, e.w - v.w);
        sse += dq.x * dq.x + dq.y * dq.y + dq.z * dq.z + dq.w * dq.w;
        *reinterpret_cast<float4 *>(q_out + n * D + (2 * j + half) * 4) =
            make_float4(v.x + dq.x, v.y + dq.y, v.z + dq.z, v.w + dq.w);
      }
      if (half == 0) {
        idx_out[n] = besti;
        atomicAdd(&hist[besti], 1);
      }
    }
  }

  __syncthreads();
  for (int i = tid; i < K; i += VQ_BLOCK)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);

  // ---- deterministic per-workgroup partial of the squared error
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if (lane == 0) red[wave] = sse;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < ISI_VQ_WAVES; ++w) t += red[w];
    sse_part[blockIdx.x] = t;
  }
}


// ---- search on the f16 matrix pipe, decision in exact fp32 (ISI_CONV_F16X3; D = 64, both f16 planes of the
// codebook in LDS: K <= 512).  The exact kernel above is bound by the fp32 matrix pipe (2 N K D flops at 157 TFLOP/s:
// 32 MFMAs of 16 passes per 32 codes).  Here the K distances of a vector are computed with split-f16 products
// (hi.hi + hi.lo + lo.hi of two 11-bit f16 pieces of 4 z and 1024 e on v_mfma_f32_32x32x16_f16: 12 MFMAs of 8
// passes per 32 codes, error ~2^-22 |z||e|) ONLY to find the two best candidates; the distances of those two are
// then recomputed in plain fp32 from the fp32 code vectors (re-read from memory, L2) -- |z|^2 - 2 z.e + |e|^2 with
// the reference's association (bottleneck.py:54-58), z.e as a fixed-order fp32 sum -- and the smaller one wins
// (ties: lower index).  The decision is therefore an fp32 one, like the reference's: index disagreement with the
// CPU reference stays at the level of the reference's own rounding (tools/index_flips.py), not that of the f16
// split.  (A third code within 2^-22 |z||e| of the best two would escape; not observed.)
// Codebook planes in LDS: [Kp][64] f16 each, a row's eight 16-byte slots XOR-swizzled with (row >> 1) & 7
// (conflict-free ds_read_b128); slot 2 s + h holds the 8 channels lane half h feeds to k-step s:
// quads 4 s + h and 4 s + 2 + h -- the channels a lane of the fused kernel below owns after its 1x1 convolution.
// A lane (col = vector, h = lane >> 5) holds the vector's quads zq[j] = quad 2 j + h.  Ranges as for the
// convolutions (|z| < 16384, |e| < 64).  Beyond: a vector out of range has NaN distances to every code -> its index
// is -1.  A finite CODE beyond the range ("far" code: the EMA update of bottleneck.py:86-92 divides an unused code by a
// vanishing cluster size -- 1e5 and more in every trained codebook) takes no part in the f16 search; vq_decide_f32 then
// proves per vector that no far code can win -- (min |e_far| - |z|)^2 exceeds the winner's distance -- or, failing that,
// compares the winner with every far code in exact fp32: the result is the reference's either way.  A non-finite code
// makes every index -1 (loud).
typedef f16s::f16x8 vq_f16x8;
constexpr float kVqNone = 0x1p115f, kVqPad = 0x1p100f;   // finite sentinels (low mantissa bits zero; -kVqNone / (2 u) finite: see vq_candidates_f16)
constexpr float kVqScaleZ = f16s::kScaleA, kVqScaleE = f16s::kScaleB, kVqUnscale = f16s::kUnscale;
__device__ __forceinline__ void vq_split4(const float4 v, const float s, uint2 &hi, uint2 &lo) { f16s::split4(v, s, hi, lo); }

// fills the two planes (and |e|^2, histogram, far-code bookkeeping) of a workgroup.  A code with a component beyond
// the f16 pieces' range (|e| * 2^10 rounds to inf) would have NaN distances: it gets zero pieces and the padding rows'
// |e|^2 = +2^100 (never a candidate), its bit is set in the far mask and far[0] keeps the smallest |e|^2 of such codes
// (vq_decide_f32).  A code with a NON-FINITE component or norm gets |e|^2 = -2^100 instead, which makes it the "best"
// candidate of EVERY vector -- vq_decide_f32 turns that into index -1 (and a NaN diff): loud.
// far: [0] bits of min |e|^2 over far codes (+inf: none), [1 + k / 32] mask.
// (Contains barriers: call from uniform control flow.)
__device__ __forceinline__ int *vq_far_area(int *hist, int Kp) { return hist + Kp; }
__device__ __forceinline__ void vq_fill_planes(unsigned short *cbh, unsigned short *cbl, float *e2, int *hist,
                                               const float *__restrict__ codes, const float *__restrict__ e2g, int K,
                                               int Kp, int tid) {
  constexpr int D = 64;
  int *far = vq_far_area(hist, Kp);
  for (int i = tid; i < Kp; i += VQ_BLOCK) { e2[i] = i < K ? e2g[i] : kVqPad; hist[i] = 0; }   // (finite: see vq_candidates_f16)
  for (int i = tid; i < 1 + Kp / 32; i += VQ_BLOCK) far[i] = i == 0 ? 0x7f800000 : 0;
  __syncthreads();
  for (int i = tid; i < Kp * (D / 4); i += VQ_BLOCK) {
    const int k = i >> 4, qd = i & 15;
    const float4 cv = k < K ? *reinterpret_cast<const float4 *>(codes + (size_t)k * D + qd * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 hi, lo;
    vq_split4(cv, kVqScaleE, hi, lo);
    const bool bad = (hi.x & 0x7C00u) == 0x7C00u || (hi.x & 0x7C000000u) == 0x7C000000u ||
                     (hi.y & 0x7C00u) == 0x7C00u || (hi.y & 0x7C000000u) == 0x7C000000u;
    if (bad) {
      const float n2 = e2g[k];
      const float s4 = (cv.x + cv.y) + (cv.z + cv.w);            // NaN / Inf if any component is not finite
      const bool finite = (s4 - s4 == 0.f) && (n2 - n2 == 0.f) && n2 > 0.f;
      hi = make_uint2(0u, 0u);
      lo = make_uint2(0u, 0u);
      if (finite) {
        // (a concurrent writer of the same row writes +-2^100 as well: -2^100 must win, so only upgrade from the norm)
        if (e2[k] == n2) e2[k] = kVqPad;
        atomicOr(&far[1 + (k >> 5)], 1 << (k & 31));
        atomicMin(&far[0], __builtin_bit_cast(int, n2));         // positive floats order like their bits
      } else {
        e2[k] = -kVqPad;
      }
    }
    const int slot = 2 * (qd >> 2) + (qd & 1), second = (qd >> 1) & 1;
    const int wo = k * D + ((slot ^ ((k >> 1) & 7)) * 8) + second * 4;
    *reinterpret_cast<uint2 *>(cbh + wo) = hi;
    *reinterpret_cast<uint2 *>(cbl + wo) = lo;
  }
  __syncthreads();
  // a row may hold a far quad AND a non-finite one: the poison must survive whatever the write order was
  for (int k = tid; k < K; k += VQ_BLOCK) {
    const float n2 = e2g[k];
    if (!(n2 - n2 == 0.f)) e2[k] = -kVqPad;
  }
}

__device__ __forceinline__ float vq_quad_dot(const float4 a, const float4 b) {
  return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}

// The two best candidates of this lane pair's vector by split-f16 distance.
//
// Round 3.  The search was bound by its vector instructions (8 per distance: fma, two compares, min, med3, three
// selects; 128 per 32-code tile and wave against 12 MFMAs -- tools/stamps_vq.py: 23-28 k cycles per 32-vector tile
// where the matrix work is 6 k).  Since then, per 32-code tile, the lane's 16 distances are ranked with their register
// number r PACKED into the four low mantissa bits -- v_and_or_b32, then v_med3 / v_max keep the tile's two best keys:
// 3 instructions per distance -- and only those two enter the exact (value, code number) update.  The distance carries
// a margin above its own rounding,
//     D = |z|^2 (1 + 2^-20) - 2 z.e + |e|^2,
// and the four truncated bits are 2^-19 of D, below the 2e-6 at which two codes count as a near-tie of the reference's
// own fp32 formula; the winner is still decided in exact fp32 between the two candidates (vq_decide_f32).  Equal
// distances (duplicate codes) keep the lower code number: lower r = lower code within a lane, `<` across tiles,
// (value, index) order between the two lane halves.  (Round 6: what is ranked is -D / (2 u), see below.)
// Sentinels (finite, so that packing never makes a NaN; powers of two, so that stripping the packed bits leaves them
// unchanged): padding rows |e|^2 = +2^100, a code beyond the f16 range
// -2^100 (vq_fill_planes: it wins everywhere and vq_decide_f32 answers -1), "no candidate" kVqNone = 2^115.
struct VqCand { float b1, b2; int i1, i2; float x2; };
__device__ __forceinline__ VqCand vq_candidates_f16(const unsigned short *cbh, const unsigned short *cbl, const float *e2,
                                                    int Kp, const float4 (&zq)[8], int col, int half) {
  constexpr int D = 64;
  s16x8v zh[4], zl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint2 h0, l0, h1, l1;
    vq_split4(zq[2 * s], kVqScaleZ, h0, l0);
    vq_split4(zq[2 * s + 1], kVqScaleZ, h1, l1);
    zh[s] = __builtin_bit_cast(s16x8v, make_uint4(h0.x, h0.y, h1.x, h1.y));
    zl[s] = __builtin_bit_cast(s16x8v, make_uint4(l0.x, l0.y, l1.x, l1.y));
  }
  float x2p = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) x2p += zq[j].x * zq[j].x + zq[j].y * zq[j].y + zq[j].z * zq[j].z + zq[j].w * zq[j].w;
  const float x2 = x2p + __shfl_xor(x2p, 32);
  // Round 6.  The accumulators hold a = -D / (2 u) -- they start from -(|z|^2 (1 + 2^-20) + |e_row|^2) / (2 u), one fma per
  // register where the zero-|e|^2 start cost a move, and the tile needs no fma behind its matrix instructions -- and the
  // ranking keeps the two LARGEST keys of a tile (a < 0 for a finite distance; the packed register number makes a key
  // more negative, so equal distances still keep the lower code).  Only a tile's two winners are scaled back to D
  // (1 / (2 u) = 2^11: exact).  Sentinels in this space: "none" -2^126 <-> D = kVqNone = 2^115.
  // The tiles are software-pipelined IN the wave (two accumulator sets, as before): tile t + 1's twelve matrix
  // instructions are issued between the pieces of tile t's ranking, and the code fragments are read from LDS one k-step
  // ahead, across tiles too.  With "read, wait, multiply, rank" per tile a wave alone on its SIMD (its partner waiting for
  // the next tile's activations) spent ~330 cycles per k-step -- LDS latency + three dependent matrix instructions + the
  // ranking's share -- 20-25 k cycles per 32-vector tile where its matrix work is 6 k (tools/stamps_vq.py); 15-21 k now.
  // (Measured and not kept: two accumulation chains per tile, |e|^2 read a tile ahead -- no change.)
  constexpr float kHalfInv = 0.5f / kVqUnscale, kANone = -kVqNone * kHalfInv;
  static_assert(kHalfInv == 2048.f, "the mapping a <-> D must be an exact power of two");
  const float ainit = -x2 * ((1.f + 0x1p-20f) * kHalfInv);
  float b1 = kVqNone, b2 = kVqNone;
  int i1 = 0, i2 = 0;
  auto init_tile = [&](f32x16 &acc, const int kt) {
    const float *e2t = e2 + kt + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(e2t[(r & 3) + 8 * (r >> 2)], -kHalfInv, ainit);
  };
  s16x8v fh[2], fl[2];                          // fragment slots: slot s & 1 holds k-step s of the tile being multiplied
  auto load_frag = [&](const int slot, const int kt, const int s) {
    const int row = min(kt + col, Kp - 1), sw = (row >> 1) & 7;     // (the prefetch behind the last tile re-reads its last rows)
    const int off = row * D + (((2 * s + half) ^ sw) * 8);
    fh[slot] = *reinterpret_cast<const s16x8v *>(cbh + off);
    fl[slot] = *reinterpret_cast<const s16x8v *>(cbl + off);
  };
  auto mfma_kstep = [&](f32x16 &acc, const int kt, const int s) {
    if (s < 3) load_frag((s + 1) & 1, kt, s + 1);
    else load_frag(0, kt + 32, 0);
    const s16x8v ah = fh[s & 1], al = fl[s & 1];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vq_f16x8, al), __builtin_bit_cast(vq_f16x8, zh[s]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vq_f16x8, ah), __builtin_bit_cast(vq_f16x8, zl[s]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vq_f16x8, ah), __builtin_bit_cast(vq_f16x8, zh[s]), acc, 0, 0, 0);
  };
  auto rank4 = [&](const f32x16 &acc, const int r0, float &t1, float &t2) {
#pragma unroll
    for (int r = r0; r < r0 + 4; ++r) {
      const float a = acc[r];                                  // (a scalar copy: __builtin_bit_cast of the vector ELEMENT read element 0)
      const float key = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) & 0xFFFFFFF0u) | (unsigned)r);
      t2 = __builtin_amdgcn_fmed3f(key, t1, t2);               // t1 >= t2 (a NaN key -- a vector out of range: every
      asm("v_max_f32 %0, %1, %2" : "=v"(t1) : "v"(key), "v"(t1));   // code, the vector ends with -1 -- leaves t1 alone); as asm:
                                                                  // fmaxf() puts a canonicalising v_max x, x in front of every key
    }
  };
  // the tile's two best into the running pair, exactly: value without the packed bits, uniform + decoded code number
  auto update = [&](const float t1, const float t2, const int kt) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const unsigned kb_ = __builtin_bit_cast(unsigned, w == 0 ? t1 : t2);
      const float v = __builtin_bit_cast(float, kb_ & 0xFFFFFFF0u) * (-2.f * kVqUnscale);
      const int cu = kt + (int)(kb_ & 3u) + 2 * (int)(kb_ & 12u);   // (r & 3) + 8 (r >> 2)
      const bool l1 = v < b1, l2 = v < b2;
      i2 = l1 ? i1 : (l2 ? cu : i2);
      b2 = l2 ? (l1 ? b1 : v) : b2;
      i1 = l1 ? cu : i1;
      b1 = l1 ? v : b1;
    }
  };
  // tile `kt_next` multiplied while tile `kt_prev` (complete) is ranked: 3 matrix instructions per 4 distances
  auto pipelined = [&](f32x16 &next, const int kt_next, const f32x16 &prev, const int kt_prev) {
    init_tile(next, kt_next);
    float t1 = kANone, t2 = kANone;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      mfma_kstep(next, kt_next, s);
      rank4(prev, 4 * s, t1, t2);
      __builtin_amdgcn_sched_barrier(0);
    }
    update(t1, t2, kt_prev);
  };
  auto rank_only = [&](const f32x16 &prev, const int kt_prev) {
    float t1 = kANone, t2 = kANone;
#pragma unroll
    for (int s = 0; s < 4; ++s) rank4(prev, 4 * s, t1, t2);
    update(t1, t2, kt_prev);
  };
  {
    f32x16 accA, accB;
    load_frag(0, 0, 0);
    init_tile(accA, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_kstep(accA, 0, s);
    int kt = 32;
    for (; kt + 32 < Kp; kt += 64) {
      pipelined(accB, kt, accA, kt - 32);
      pipelined(accA, kt + 32, accB, kt);
    }
    if (kt < Kp) {
      pipelined(accB, kt, accA, kt - 32);
      rank_only(accB, kt);
    } else {
      rank_only(accA, kt - 32);
    }
  }
  i1 += 4 * half;
  i2 += 4 * half;
  {  // merge the two halves' candidates (strict order on (value, index): both lanes end with the same pair)
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int j1 = __shfl_xor(i1, 32), j2 = __shfl_xor(i2, 32);
#define ISI_VQ_INSERT(DV, CV)                                                          \
    do {                                                                               \
      if ((DV) < b1 || ((DV) == b1 && (CV) < i1)) { b2 = b1; i2 = i1; b1 = (DV); i1 = (CV); } \
      else if ((DV) < b2 || ((DV) == b2 && (CV) < i2)) { b2 = (DV); i2 = (CV); }       \
    } while (0)
    ISI_VQ_INSERT(o1, j1);
    ISI_VQ_INSERT(o2, j2);
#undef ISI_VQ_INSERT
  }
  return VqCand{b1, b2, i1, i2, x2};
}

// The decision, in fp32, between the two candidates: returns the index of the nearest code (-1: no finite distance)
// and, in `ew`, this lane's quads of the winner's fp32 code vector.
__device__ __forceinline__ int vq_decide_f32(const VqCand c, const float *e2, int K, const float *__restrict__ codes,
                                             const float4 (&zq)[8], int half, float4 (&ew)[8],
                                             const int *far = nullptr, const float *__restrict__ e2g = nullptr) {
  constexpr int D = 64;
  if (!(c.b1 < kVqNone) || c.b1 < -0.5f * kVqPad) return -1;   // no finite distance / a non-finite code
  const int i1 = c.i1, i2 = c.i2;
  const bool has1 = c.b1 < 0.5f * kVqPad && i1 < K;           // (false: every code of the book is a far code)
  const bool has2 = has1 && c.b2 < 0.5f * kVqPad && i2 < K && i2 != i1;
  const float *r1 = codes + (size_t)(has1 ? i1 : 0) * D + half * 4, *r2 = codes + (size_t)(has2 ? i2 : (has1 ? i1 : 0)) * D + half * 4;
  float4 e2q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ew[j] = *reinterpret_cast<const float4 *>(r1 + j * 8);
    e2q[j] = *reinterpret_cast<const float4 *>(r2 + j * 8);
  }
  const float x2 = c.x2;
  float p1 = 0.f, p2 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { p1 += vq_quad_dot(ew[j], zq[j]); p2 += vq_quad_dot(e2q[j], zq[j]); }
  const float q1 = __shfl_xor(p1, 32), q2 = __shfl_xor(p2, 32);
  const float dot1 = half ? q1 + p1 : p1 + q1, dot2 = half ? q2 + p2 : p2 + q2;   // lower half first on both lanes
  const float d1 = has1 ? (x2 - 2.f * dot1) + e2[i1] : INFINITY;
  const float d2 = has2 ? (x2 - 2.f * dot2) + e2[i2] : INFINITY;
  int best = i1;
  float dbest = d1;
  if (d2 < d1 || (d2 == d1 && i2 < i1)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ew[j] = e2q[j];
    best = i2;
    dbest = d2;
  }
  // ---- far codes (beyond the f16 range, absent from the search above).  |z - e| >= |e| - |z|, so with the nearest far
  // norm n = min |e_far|: (n - |z|)^2 bounds every far code's distance from below; the fp32 formula's own rounding is
  // ~2^-22 of (|z| + |e|)^2, the 0.999 leaves three orders of magnitude more.  Certificate fails (or no near code at all):
  // every lane of the wave scans the far codes in exact fp32, (distance, index) order like the reference's first-index
  // argmax.  Wave-uniform by construction (the shuffles below need every lane).
  if (far != nullptr) {
    const float fmin2 = __builtin_bit_cast(float, far[0]);
    if (fmin2 < INFINITY) {
      const float gap = __builtin_sqrtf(fmin2) - __builtin_sqrtf(x2);
      const bool certified = gap > 0.f && gap * gap * 0.999f > dbest;
      if (__any(!certified)) {
        for (int w = 0; w < (K + 31) / 32; ++w) {
          unsigned m = (unsigned)far[1 + w];
          while (m) {
            const int k = 32 * w + __builtin_ctz(m);
            m &= m - 1;
            const float *rk = codes + (size_t)k * D + half * 4;
            float4 ek[8];
            float pk = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              ek[j] = *reinterpret_cast<const float4 *>(rk + j * 8);
              pk += vq_quad_dot(ek[j], zq[j]);
            }
            const float qk = __shfl_xor(pk, 32);
            const float dk = (x2 - 2.f * (half ? qk + pk : pk + qk)) + e2g[k];
            if (dk < dbest || (dk == dbest && k < best)) {     // (a NaN distance compares false)
              dbest = dk;
              best = k;
#pragma unroll
              for (int j = 0; j < 8; ++j) ew[j] = ek[j];
            }
          }
        }
      }
    }
  }
  if (!(dbest < INFINITY)) return -1;
  return best;
}

__device__ __forceinline__ int vq_search_f16(const unsigned short *cbh, const unsigned short *cbl, const float *e2, int Kp,
                                             int K, const float *__restrict__ codes, const float4 (&zq)[8], int col,
                                             int half, float4 (&ew)[8], const int *far, const float *__restrict__ e2g) {
  return vq_decide_f32(vq_candidates_f16(cbh, cbl, e2, Kp, zq, col, half), e2, K, codes, zq, half, ew, far, e2g);
}

__global__ __launch_bounds__(VQ_BLOCK) void vq_nearest_f16x3_kernel(
    const float *__restrict__ z, const float *__restrict__ codes, const float *__restrict__ e2g,
    int64_t *__restrict__ idx_out, float *__restrict__ q_out, int32_t *__restrict__ counts,
    float *__restrict__ sse_part, int64_t N, int K) {
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Kp = (K + 31) & ~31;                                     // padding rows: zeros, |e|^2 = +inf
  unsigned short *cbh = reinterpret_cast<unsigned short *>(smem);   // [Kp][64] hi pieces
  unsigned short *cbl = cbh + (size_t)Kp * D;                       // [Kp][64] lo pieces
  float *e2 = reinterpret_cast<float *>(cbl + (size_t)Kp * D);      // [Kp]
  float *red = e2 + Kp;
  int *hist = reinterpret_cast<int *>(red + ISI_VQ_WAVES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int half = lane >> 5;

  vq_fill_planes(cbh, cbl, e2, hist, codes, e2g, K, Kp, tid);
  __syncthreads();

  float sse = 0.f;
  const int64_t n_iter = (N + VQ_VEC_PER_BLOCK_ITER - 1) / VQ_VEC_PER_BLOCK_ITER;
  for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const int64_t n = it * VQ_VEC_PER_BLOCK_ITER + wave * 32 + col;
    const bool valid = n < N;
    float4 zq[8], ew[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      zq[j] = valid ? *reinterpret_cast<const float4 *>(z + n * D + (2 * j + half) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int besti = vq_search_f16(cbh, cbl, e2, Kp, K, codes, zq, col, half, ew, vq_far_area(hist, Kp), e2g);
    if (valid && besti < 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<float4 *>(q_out + n * D + (2 * j + half) * 4) = make_float4(NAN, NAN, NAN, NAN);
      sse = NAN;
      if (half == 0) idx_out[n] = -1;
    } else if (valid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 e = ew[j];
        const float4 v = zq[j];
        float4 dq = make_float4(e.x - v.x, e.y - v.y, e.z - v.z, e.w - v.w);
        sse += dq.x * dq.x + dq.y * dq.y + dq.z * dq.z + dq.w * dq.w;
        *reinterpret_cast<float4 *>(q_out + n * D + (2 * j + half) * 4) = make_float4(v.x + dq.x, v.y + dq.y, v.z + dq.z, v.w + dq.w);
      }
      if (half == 0) {
        idx_out[n] = besti;
        atomicAdd(&hist[besti], 1);
      }
    }
  }

  __syncthreads();
  for (int i = tid; i < K; i += VQ_BLOCK)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if (lane == 0) red[wave] = sse;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < ISI_VQ_WAVES; ++w) t += red[w];
    sse_part[blockIdx.x] = t;
  }
}


// ---- fused quantize_conv (1x1) + codebook search (SURVEY K4 + K5; reference vqvae/vqvae.py:260,272 followed by
// bottleneck.py:53-61): z never goes to memory.  Per wave and pass, 32 pixels:
//
//   z^T[d][pixel] = sum_c W[d][c] a[pixel][c] + bias[d]     split-f16 products (hi.hi + hi.lo + lo.hi), operands SWAPPED
//                                                            (weights = MFMA rows, pixels = MFMA columns)
//
// With the swap a lane's accumulators of the two 32-channel row tiles hold dims (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// (+ 32 per tile) of ONE pixel: exactly the quads `zq[j]` (quad 2 j + half) the distance MFMAs below take as their
// B operand -- the 1x1 convolution's result is consumed in the registers it is born in.  The activations come in the
// pair8 format (split_f16.h): a lane's operand fragment for a k-step is one 16-byte piece of its pixel's row, loaded
// straight from memory (two sources for cat(dec_t, enc_b), vqvae.py:270); the weight pieces (blocked ISI_CONV_W16
// copy) likewise -- 48 KiB per pass from L2, against 33 k matrix-pipe cycles of distance products per pass, and a
// wave waiting for them leaves the pipe to its partner (no barriers in this loop).  Products, k order and term
// order are those of the stand-alone convolution (conv_igemm_f32.hip): z, and hence every output, is bit-identical
// to the two-launch path.  Outputs as vq_nearest_kernel, plus an optional pair8 copy of q for the pair pipeline.
#ifdef ISI_MEASURE
#define ISI_VQ_DBGBIT(p, b) ((p).dbg & (b))
#else
#define ISI_VQ_DBGBIT(p, b) (0)     // the ablations are not compiled into the default build
#endif
// phase timestamps (-DISI_MEASURE builds; tools/stamps_vq.py): workgroup 8, waves 0 and 4, its second iteration
#ifdef ISI_MEASURE
__device__ long long g_vq_stamps[128];
#define ISI_VQ_STAMP(i_) do { if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && it == (int64_t)blockIdx.x + (int64_t)gridDim.x) \
    g_vq_stamps[(wave >> 2) * 64 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ISI_VQ_STAMP(i_) do { } while (0)
#endif
struct VqFusedArgs {
  const float *in0, *in1, *w16, *bias, *codes, *e2;
  const float *wfrag;                    // fragment-major copy of w16 (vq_weight_fragments_kernel)
  int64_t *idx;
  float *q, *q_pair;
  float *z_out;                          // nullable: z itself ([N][64] fp32), for the training step's backward and EMA sums
  int32_t *counts;
  float *sse_part;
  unsigned in0_bytes, in1_bytes, w_bytes;
  int C0, C1, Kpad;
  int s0n, s0h, s0w, s1n, s1h, s1w;     // element strides of the two sources (channel stride 1)
  int H, W;                             // pixel grid: vector n = (b H + y) W + x
  int64_t N;
  int K;
  int dbg;                              // measurements only (ISI_VQ_DBG): 1 no search, 2 no convolution, 4 no decision, 8 no stores
};

// The 1x1 weight re-laid out FRAGMENT-major for the kernel below: piece ((k-step * 2 + tile) * 2 + plane) holds, for
// lane l, the 16 bytes lane l feeds to the MFMA as its A operand (row 32 tile + (l & 31), channel group
// 2 k-step + (l >> 5)): one weight load instruction then reads 1 KiB contiguously instead of 32 rows 768 bytes apart.
// `counts` (nullable): the histogram the search accumulates into is zeroed here as well (one tiny launch instead of two)
__global__ void vq_weight_fragments_kernel(const uint4 *__restrict__ w16, uint4 *__restrict__ wf, int Kpad, int nstep,
                                           int32_t *__restrict__ counts, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (step, tile, plane, lane)
  if (counts && i < K) counts[i] = 0;
  if (i >= nstep * 2 * 2 * 64) return;
  const int lane = i & 63, plane = (i >> 6) & 1, t = (i >> 7) & 1, step = i >> 8;
  const int row = 32 * t + (lane & 31), grp = 2 * step + (lane >> 5);          // 8-channel group of the row
  wf[i] = w16[((size_t)row * Kpad + grp * 8) / 4 + plane];
}

template <int NCH>   // 32-channel chunks of C0 + C1, rounded up to an even number
__global__ __launch_bounds__(VQ_BLOCK) void vq_conv1x1_nearest_kernel(const VqFusedArgs p) {
  constexpr int D = 64, NQ = D / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = p.K;
  const int Kp = (K + 31) & ~31;
  unsigned short *cbh = reinterpret_cast<unsigned short *>(smem);   // [Kp][64] hi pieces
  unsigned short *cbl = cbh + (size_t)Kp * D;                       // [Kp][64] lo pieces
  float *e2 = reinterpret_cast<float *>(cbl + (size_t)Kp * D);      // [Kp]
  float *red = e2 + Kp;
  int *hist = reinterpret_cast<int *>(red + ISI_VQ_WAVES);

  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int wave = tid >> 6;

  vq_fill_planes(cbh, cbl, e2, hist, p.codes, p.e2, K, Kp, tid);
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in0), 0, p.in0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1), 0, p.in1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wfrag), 0, p.w_bytes, 0x00020000);
  constexpr unsigned OOBV = 0xFFFFFFF0u;
  // this lane's weight rows (MFMA A operand: row = output channel 32 t + col, k-block = half): byte offset of k = 0
  const int nchunk = (p.C0 + p.C1) / 32;
  const int hw = p.H * p.W;

  float sse = 0.f;
  const int64_t n_iter = (p.N + VQ_VEC_PER_BLOCK_ITER - 1) / VQ_VEC_PER_BLOCK_ITER;
  // byte offsets of vector n's rows in the two sources (OOBV: beyond N)
  auto pixel_offsets = [&](const int64_t n, unsigned &a0, unsigned &a1) {
    a0 = OOBV; a1 = OOBV;
    if (n < p.N) {
      const int b = (int)(n / hw);
      const int rem = (int)(n - (int64_t)b * hw);
      const int y = rem / p.W, x = rem - y * p.W;
      a0 = (unsigned)(b * p.s0n + y * p.s0h + x * p.s0w) * 4u;
      a1 = (unsigned)(b * p.s1n + y * p.s1h + x * p.s1w) * 4u;
    }
  };
  // ALL activation pieces of a tile (C0 + C1 <= 256 channels: up to 32 pieces of 16 bytes per lane) are requested
  // before anything is multiplied: they come from HBM and their round trip used to be paid once per two-chunk batch;
  // the weight fragments (L2-resident) follow batch by batch.  (Requesting them one tile AHEAD, between the previous
  // tile's decision and its stores, was measured: 0.209 against 0.195 ms for the two launches of a forward -- the
  // CU's vector-memory queue is what the tile waits for: ~105 memory instructions per wave and tile.  Round 6: HALF of
  // them a tile ahead, in front of the candidate search: the 1x1 phase 35 -> 23 k cycles, the search 15 -> 23-30 k
  // (17-35 spilled registers at 192 / 256 channels), 0.196 against 0.176 ms.)
  constexpr int NU = 2 * NCH;                                // k-steps of 16 channels
  i32x4v ahv[NU], alv[NU];
  auto request_tile = [&](const int64_t it_, const int lane_) {
    const int64_t n_ = it_ * VQ_VEC_PER_BLOCK_ITER + wave * 32 + (lane_ & 31);
    unsigned a0, a1;
    pixel_offsets(it_ < n_iter ? n_ : p.N, a0, a1);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int cc = u >> 1, s_ = u & 1;
      const bool live = cc < nchunk;                           // uniform
      const bool second = cc * 32 >= p.C0;                     // uniform
      const unsigned abase = second ? a1 : a0;
      const unsigned acol = (unsigned)(second ? cc * 32 - p.C0 : cc * 32) * 4u;
      // pieces of channel group 2 s + half of the chunk: hi at + 0, lo at + 16 (OOBV: zeros)
      const unsigned ao = (abase == OOBV || !live || ISI_VQ_DBGBIT(p, 32)) ? OOBV : abase + acol + (unsigned)(2 * s_ + (lane_ >> 5)) * 32u;
      ahv[u] = __builtin_amdgcn_raw_buffer_load_b128(second ? rs1 : rs0, ao, 0, 0);
      alv[u] = __builtin_amdgcn_raw_buffer_load_b128(second ? rs1 : rs0, ao == OOBV ? OOBV : ao + 16u, 0, 0);
    }
  };
  for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
    // the lane index is made opaque per iteration: the dozens of LDS / global offsets derived from it are invariant
    // across iterations and would otherwise be hoisted out of the loop into registers the tile's prefetched pieces need
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int col = lane & 31, half = lane >> 5;
    const int64_t n = it * VQ_VEC_PER_BLOCK_ITER + wave * 32 + col;
    const bool valid = n < p.N;
    ISI_VQ_STAMP(0);
    request_tile(it, lane);
    // ---- 1x1 convolution into this pixel's dims
    f32x16 zt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) zt[t][r] = 0.f;
#define ISI_VQ_MF(W_, A_, T_) zt[T_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vq_f16x8, W_), __builtin_bit_cast(vq_f16x8, A_), zt[T_], 0, 0, 0)
    if (ISI_VQ_DBGBIT(p, 2)) {
      zt[0][0] = (float)(n & 7); zt[1][5] = 1.f;
    } else {
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
      i32x4v w0h[4], w0l[4], w1h[4], w1l[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + (u >> 1), s_ = u & 1;
        const bool live = cc < nchunk;                          // uniform (odd chunk counts: zero operands)
        // weight pieces, fragment-major: ((step * 2 + tile) * 2 + plane) * 1 KiB + 16 lane
        const unsigned wo = (live && !ISI_VQ_DBGBIT(p, 16)) ? (unsigned)((cc * 2 + s_) * 4) * 1024u + (unsigned)lane * 16u : OOBV;
        w0h[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wo, 0, 0);
        w0l[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wo == OOBV ? OOBV : wo + 1024u, 0, 0);
        w1h[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wo == OOBV ? OOBV : wo + 2048u, 0, 0);
        w1l[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wo == OOBV ? OOBV : wo + 3072u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int uu = 2 * c + u;                              // k-step number (compile-time after unrolling)
        ISI_VQ_MF(w0h[u], alv[uu], 0); ISI_VQ_MF(w1h[u], alv[uu], 1);
        ISI_VQ_MF(w0l[u], ahv[uu], 0); ISI_VQ_MF(w1l[u], ahv[uu], 1);
        ISI_VQ_MF(w0h[u], ahv[uu], 0); ISI_VQ_MF(w1h[u], ahv[uu], 1);
      }
    }
    }
#undef ISI_VQ_MF
    ISI_VQ_STAMP(1);
    float4 zq[NQ], ew[NQ];
    // bias of this lane's dims: quad 2 j + half of tile t = j / 4 -> channels 8 j + 4 half + e.  Re-read per tile (L2):
    // kept across the loop the 32 registers pushed the prefetching variants into scratch (the offset is made opaque so
    // that the compiler does not hoist the loads back out)
    int opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
    float4 bq[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      bq[j] = p.bias ? *reinterpret_cast<const float4 *>(p.bias + 8 * j + 4 * half + opaque0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int t = j >> 2, r0 = (j & 3) * 4;
      float4 v = make_float4(zt[t][r0] * kVqUnscale + bq[j].x, zt[t][r0 + 1] * kVqUnscale + bq[j].y,
                             zt[t][r0 + 2] * kVqUnscale + bq[j].z, zt[t][r0 + 3] * kVqUnscale + bq[j].w);
      if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
      zq[j] = v;
    }
    if (p.z_out && valid) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) *reinterpret_cast<float4 *>(p.z_out + n * D + (2 * j + half) * 4) = zq[j];
    }
    ISI_VQ_STAMP(2);
    // ---- search (candidates on the f16 pipe, decision in fp32; shared with the stand-alone kernel)
    VqCand cand{0.f, 1.f, (int)(n & 255), (int)(n & 255) + 256};
    if (!ISI_VQ_DBGBIT(p, 1)) cand = vq_candidates_f16(cbh, cbl, e2, Kp, zq, col, half);
    int besti = cand.i1;
    ISI_VQ_STAMP(3);
    if (!ISI_VQ_DBGBIT(p, 4)) besti = vq_decide_f32(cand, e2, K, p.codes, zq, half, ew, vq_far_area(hist, Kp), p.e2);
    else {
#pragma unroll
      for (int j = 0; j < NQ; ++j) ew[j] = zq[j];
    }
    if (ISI_VQ_DBGBIT(p, 8)) { sse += zq[0].x + ew[3].y; continue; }
    ISI_VQ_STAMP(4);
    const bool lost = besti < 0;
    if (valid && lost) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        if (p.q) *reinterpret_cast<float4 *>(p.q + n * D + (2 * j + half) * 4) = make_float4(NAN, NAN, NAN, NAN);
        if (p.q_pair) {   // NaN pieces: the pair pipeline's consumers stay loud as well
          *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(p.q_pair) + (n * D + 8 * j) * 4 + half * 8) = make_uint2(0x7e007e00u, 0x7e007e00u);
          *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(p.q_pair) + (n * D + 8 * j) * 4 + 16 + half * 8) = make_uint2(0x7e007e00u, 0x7e007e00u);
        }
      }
      sse = NAN;
      if (half == 0) p.idx[n] = -1;
    } else if (valid) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float4 e = ew[j];
        const float4 v = zq[j];
        float4 dq = make_float4(e.x - v.x, e.y - v.y, e.z - v.z, e.w - v.w);
        sse += dq.x * dq.x + dq.y * dq.y + dq.z * dq.z + dq.w * dq.w;
        const float4 qv = make_float4(v.x + dq.x, v.y + dq.y, v.z + dq.z, v.w + dq.w);
        // (p.q null: a forward whose caller takes no quantised maps -- the decoders read q_pair: 8 scattered 16-byte
        // stores per lane and tile less in a kernel bound by its vector-memory instructions)
        if (p.q) *reinterpret_cast<float4 *>(p.q + n * D + (2 * j + half) * 4) = qv;
        if (p.q_pair) {
          // pair8 group j = channels 8 j .. 8 j + 7: this lane's quad is its half `half`: 8 bytes of hi pieces at
          // + 8 half, 8 bytes of lo pieces at + 16 + 8 half
          // v_permlane32_swap hands the lower-half lane both lanes' hi quads and the upper-half lane both lo quads
          // (tools/probes/permlane_probe.hip): one 16-byte store per lane instead of two 8-byte ones (the kernel is
          // bound by the CU's vector-memory instruction rate)
          uint2 hi, lo;
          f16s::split4(qv, f16s::kScaleA, hi, lo);
          typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
          const u32x2v sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
          const u32x2v sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
          char *g8 = reinterpret_cast<char *>(p.q_pair) + (n * D + 8 * j) * 4;
          *reinterpret_cast<uint4 *>(g8 + half * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
        }
      }
      if (half == 0) {
        p.idx[n] = besti;
        atomicAdd(&hist[besti], 1);
      }
    }
    ISI_VQ_STAMP(5);
  }

  const int lane = lane0;
  __syncthreads();
  for (int i = tid; i < Kp; i += VQ_BLOCK)
    if (hist[i]) atomicAdd(&p.counts[i], hist[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if (lane == 0) red[wave] = sse;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < ISI_VQ_WAVES; ++w) t += red[w];
    p.sse_part[blockIdx.x] = t;
  }
}

__global__ void vq_finalize_kernel(const float *__restrict__ sse_part, int n_part,
                                   const int32_t *__restrict__ counts, int K, int64_t N, int D,
                                   float *__restrict__ out2) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < n_part; i += 256) s += sse_part[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float sse = red[0];
  __syncthreads();
  float h = 0.f;
  for (int i = tid; i < K; i += 256) {
    const float pr = (float)counts[i] / (float)N;
    h += pr * logf(fmaxf(pr, 1e-7f));
  }
  red[tid] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    out2[0] = sse / ((float)N * (float)D);
    out2[1] = expf(-red[0]);
  }
}

// Both levels' scalars in ONE launch behind the bottom search (the forward's fused path): threads [0, 256) finish the top
// level, [256, 512) the bottom level; out5 = {diff_t, perplexity_t, diff_b, perplexity_b, diff_t + diff_b}.
__global__ __launch_bounds__(512) void vq_finalize2_kernel(const float *__restrict__ sse_t, int np_t,
                                                           const int32_t *__restrict__ counts_t, int K_t, int64_t N_t,
                                                           const float *__restrict__ sse_b, int np_b,
                                                           const int32_t *__restrict__ counts_b, int K_b, int64_t N_b,
                                                           int D, float *__restrict__ out5) {
  __shared__ float red[2][256];
  __shared__ float diff[2];
  const int lvl = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const float *sse_part = lvl ? sse_b : sse_t;
  const int32_t *counts = lvl ? counts_b : counts_t;
  const int n_part = lvl ? np_b : np_t, K = lvl ? K_b : K_t;
  const int64_t N = lvl ? N_b : N_t;
  float s = 0.f;
  for (int i = tid; i < n_part; i += 256) s += sse_part[i];
  red[lvl][tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[lvl][tid] += red[lvl][tid + o];
    __syncthreads();
  }
  const float sse = red[lvl][0];
  __syncthreads();
  float h = 0.f;
  for (int i = tid; i < K; i += 256) {
    const float pr = (float)counts[i] / (float)N;
    h += pr * logf(fmaxf(pr, 1e-7f));
  }
  red[lvl][tid] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[lvl][tid] += red[lvl][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float dd = sse / ((float)N * (float)D);
    out5[2 * lvl] = dd;
    out5[2 * lvl + 1] = expf(-red[lvl][0]);
    diff[lvl] = dd;
  }
  __syncthreads();
  if (threadIdx.x == 0) out5[4] = diff[0] + diff[1];
}
__global__ void vq_scalars_sum_kernel(float *__restrict__ out5) { out5[4] = out5[0] + out5[2]; }

__global__ void embed_code_kernel(const int64_t *__restrict__ idx, const float *__restrict__ codes,
                                  float *__restrict__ out, int64_t N, int D4, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D4) return;
  const int64_t n = i / D4;
  const int qd = (int)(i - n * D4);
  int64_t k = idx[n];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);  // callers validate; clamp keeps reads in bounds
  reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(codes)[k * D4 + qd];
}

static size_t vq_planes_lds_bytes(int Kp) {   // planes, |e|^2, per-wave sums, histogram, far-code mask (vq_fill_planes)
  return (size_t)Kp * 64 * 2 * sizeof(unsigned short) + ((size_t)2 * Kp + ISI_VQ_WAVES + 1 + Kp / 32 + 3) * sizeof(float);
}

static int vq_grid(int64_t N) {
  const int64_t n_iter = (N + VQ_VEC_PER_BLOCK_ITER - 1) / VQ_VEC_PER_BLOCK_ITER;
  return (int)(n_iter < 256 ? (n_iter < 1 ? 1 : n_iter) : 256);
}

int vq_num_partials(int64_t N) { return vq_grid(N); }

template <int D>
static int launch_vq(const float *z, const float *codes, const float *e2, int64_t *idx, float *q,
                     int32_t *counts, float *sse_part, int64_t N, int K, hipStream_t stream) {
  const int Kp = (K + 31) & ~31;
  const size_t smem = ((size_t)Kp * (D + 4) + 2 * Kp + ISI_VQ_WAVES) * sizeof(float);
  if (smem > 150 * 1024) return unsupported("vq: codebook does not fit in LDS");
  auto kern = vq_nearest_kernel<D>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return check_launch("hipFuncSetAttribute(vq)");
  {
    prof::Scope scope(prof::K_VQ_NEAREST, 2.0 * N * K * D, 4.0 * (2.0 * N * D + (double)K * D) + 8.0 * N,
                      stream);
    ISI_PROF_LAUNCH(scope, kern, dim3(vq_grid(N)), dim3(VQ_BLOCK), smem, stream, z, codes, e2, idx, q,
                       counts, sse_part, N, K);
  }
  return check_launch("vq_nearest_f32");
}

static int launch_vq_f16x3(const float *z, const float *codes, const float *e2, int64_t *idx, float *q,
                           int32_t *counts, float *sse_part, int64_t N, int K, hipStream_t stream) {
  const int Kp = (K + 31) & ~31;
  const size_t smem = vq_planes_lds_bytes(Kp);
  auto kern = vq_nearest_f16x3_kernel;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return check_launch("hipFuncSetAttribute(vq f16x3)");
  {
    prof::Scope scope(prof::K_VQ_NEAREST, 2.0 * N * K * 64, 4.0 * (2.0 * N * 64 + (double)K * 64) + 8.0 * N, stream);
    ISI_PROF_LAUNCH(scope, kern, dim3(vq_grid(N)), dim3(VQ_BLOCK), smem, stream, z, codes, e2, idx, q,
                       counts, sse_part, N, K);
  }
  return check_launch("vq_nearest_f16x3");
}

// flags: 0 exact fp32 products | ISI_CONV_F16X3 split-f16 products (D = 64 with the codebook's two f16 planes
// fitting in LDS; other shapes run the exact kernel)
int vq_nearest_f32(const float *z, const float *codes, const float *e2, int64_t *idx, float *q,
                   int32_t *counts, float *sse_part, int64_t N, int D, int K, int flags,
                   hipStream_t stream) {
  if (!z || !codes || !e2 || !idx || !q || !counts || !sse_part) return invalid("vq: null pointer");
  if (N <= 0 || K <= 0) return invalid("vq: need N > 0 and K > 0");
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(codes) |
       reinterpret_cast<uintptr_t>(q)) & 15)
    return invalid("vq: z, codes and q must be 16-byte aligned");
  if ((flags & ISI_CONV_F16X3) && D == 64 && vq_planes_lds_bytes((K + 31) & ~31) <= 150 * 1024)
    return launch_vq_f16x3(z, codes, e2, idx, q, counts, sse_part, N, K, stream);
  switch (D) {
    case 8: return launch_vq<8>(z, codes, e2, idx, q, counts, sse_part, N, K, stream);
    case 16: return launch_vq<16>(z, codes, e2, idx, q, counts, sse_part, N, K, stream);
    case 32: return launch_vq<32>(z, codes, e2, idx, q, counts, sse_part, N, K, stream);
    case 64: return launch_vq<64>(z, codes, e2, idx, q, counts, sse_part, N, K, stream);
    default: return unsupported("vq: embed_dim must be 8, 16, 32 or 64");
  }
}


int vq_zero_counts(int32_t *counts, int K, hipStream_t stream);
// Fused quantize_conv + search (see vq_conv1x1_nearest_kernel).  Sources: pair8, channels-last, 32-channel multiples;
// D = 64; the weight is the packed 1x1 weight's blocked pair copy (ISI_CONV_W16: packed_w + Cout * Kpad floats).
bool vq_conv1x1_fusable(int C0, int C1, int D, int K) {
  const int Kp = (K + 31) & ~31;
  return D == 64 && C0 > 0 && C0 % 32 == 0 && C1 % 32 == 0 && C0 + C1 <= 256 && vq_planes_lds_bytes(Kp) <= 150 * 1024;
}

size_t vq_conv1x1_workspace_floats(int C0, int C1, int D) { return (size_t)D * round_up((size_t)(C0 + C1), kBK); }
// the fragment-major copy of a packed 1x1 weight's pair copy, made ONCE at pack time (the third section of such a weight:
// isi_vqvae_w.w16 == 2) instead of by a pre-kernel in front of every search
int vq_pack_fragments_f32(const float *packed_w16, float *frag_out, int Kpad, hipStream_t stream) {
  if (!packed_w16 || !frag_out || Kpad <= 0 || (Kpad % 16) ||
      ((reinterpret_cast<uintptr_t>(packed_w16) | reinterpret_cast<uintptr_t>(frag_out)) & 15))
    return invalid("vq_pack_fragments: bad argument");
  const int nstep = Kpad / 16, total = nstep * 256;
  hipLaunchKernelGGL(vq_weight_fragments_kernel, dim3((total + 255) / 256), dim3(256), 0, stream,
                     reinterpret_cast<const uint4 *>(packed_w16), reinterpret_cast<uint4 *>(frag_out), Kpad, nstep,
                     (int32_t *)nullptr, 0);
  return check_launch("vq_pack_fragments");
}

int vq_debug_stamps(long long *host, int n) {
#ifdef ISI_MEASURE
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vq_stamps), sizeof(long long) * (size_t)(n < 128 ? n : 128)) == hipSuccess ? 0 : -2;
#else
  (void)host; (void)n;
  return unsupported("phase timestamps need a -DISI_MEASURE build");
#endif
}

int vq_conv1x1_nearest_f32(const isi_src *s0, const isi_src *s1, const float *w16, const float *bias, const float *codes,
                           const float *e2, int64_t *idx, float *q, float *q_pair, int32_t *counts, float *sse_part,
                           float *workspace, int B, int H, int W, int D, int K, hipStream_t stream, bool zero_counts,
                           float *z_out, const float *wfrag_packed) {
  if (!wfrag_packed && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15))) return invalid("vq_conv1x1: workspace (vq_conv1x1_workspace_floats, 16-byte aligned)");
  if (wfrag_packed && (reinterpret_cast<uintptr_t>(wfrag_packed) & 15)) return invalid("vq_conv1x1: the packed fragments must be 16-byte aligned");
  if (!s0 || !s0->ptr || !w16 || !codes || !e2 || !idx || (!q && !q_pair) || !counts || !sse_part) return invalid("vq_conv1x1: null pointer");
  const bool two = s1 && s1->ptr;
  const int C0 = s0->C, C1 = two ? s1->C : 0;
  if (B <= 0 || H <= 0 || W <= 0 || !vq_conv1x1_fusable(C0, C1, D, K)) return unsupported("vq_conv1x1: shape outside the fused kernel");
  if (s0->sc != 1 || (two && s1->sc != 1)) return unsupported("vq_conv1x1: sources must be channels-last");
  const int64_t N = (int64_t)B * H * W;
  const int64_t e0 = (int64_t)(B - 1) * s0->sn + (int64_t)(H - 1) * s0->sh + (int64_t)(W - 1) * s0->sw + C0;
  const int64_t e1 = two ? (int64_t)(B - 1) * s1->sn + (int64_t)(H - 1) * s1->sh + (int64_t)(W - 1) * s1->sw + C1 : 1;
  if (e0 > ((int64_t)1 << 30) || e1 > ((int64_t)1 << 30) || N * D > ((int64_t)1 << 30)) return unsupported("vq_conv1x1: a tensor spans 4 GiB or more");
  VqFusedArgs a;
  memset(&a, 0, sizeof a);
  a.in0 = s0->ptr; a.in1 = two ? s1->ptr : s0->ptr; a.w16 = w16; a.bias = bias; a.codes = codes; a.e2 = e2;
  a.idx = idx; a.q = q; a.q_pair = q_pair; a.z_out = z_out; a.counts = counts; a.sse_part = sse_part;
  a.in0_bytes = (unsigned)(e0 * 4); a.in1_bytes = two ? (unsigned)(e1 * 4) : a.in0_bytes;
  a.C0 = C0; a.C1 = C1; a.Kpad = (int)round_up((size_t)(C0 + C1), kBK);
  a.w_bytes = (unsigned)((size_t)D * a.Kpad * 4);
  a.s0n = (int)s0->sn; a.s0h = (int)s0->sh; a.s0w = (int)s0->sw;
  if (two) { a.s1n = (int)s1->sn; a.s1h = (int)s1->sh; a.s1w = (int)s1->sw; }
  a.H = H; a.W = W; a.N = N; a.K = K;
  const int Kp = (K + 31) & ~31;
  const size_t smem = vq_planes_lds_bytes(Kp);
  a.wfrag = wfrag_packed ? wfrag_packed : workspace;
  if (wfrag_packed) {          // fragments made at pack time (isi_vq_pack_fragments_f32): no pre-kernel
    if (zero_counts) {
      const int rc0 = vq_zero_counts(counts, K, stream);
      if (rc0) return rc0;
    }
  } else {
    const int nstep = a.Kpad / 16, total = nstep * 256 > K ? nstep * 256 : K;
    hipLaunchKernelGGL(vq_weight_fragments_kernel, dim3((total + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(w16), reinterpret_cast<uint4 *>(workspace), a.Kpad, nstep,
                       zero_counts ? counts : nullptr, K);
  }
  const int nch2 = ((C0 + C1) / 32 + 1) / 2;                 // pairs of 32-channel chunks
  auto kern = nch2 <= 1 ? vq_conv1x1_nearest_kernel<2> : nch2 == 2 ? vq_conv1x1_nearest_kernel<4>
              : nch2 == 3 ? vq_conv1x1_nearest_kernel<6> : vq_conv1x1_nearest_kernel<8>;
  a.dbg = knobs().vq_dbg;   // 0 outside -DISI_MEASURE builds
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return check_launch("hipFuncSetAttribute(vq_conv1x1)");
  {
    const double cin = C0 + C1;
    prof::Scope scope(prof::K_VQ_NEAREST, 2.0 * N * K * D + 2.0 * N * D * cin, 4.0 * (N * cin + 2.0 * N * D + (double)K * D) + 8.0 * N, stream);
    ISI_PROF_LAUNCH(scope, kern, dim3(vq_grid(N)), dim3(VQ_BLOCK), smem, stream, a);
  }
  return check_launch("vq_conv1x1_nearest_f32");
}

// (a kernel, not hipMemsetAsync: memset nodes inside a replayed HIP graph are not reliably ordered on ROCm 7.2 -- DESIGN.md §6)
__global__ void vq_zero_counts_kernel(int32_t *__restrict__ counts, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) counts[i] = 0;
}
__global__ void vq_unquantized_scalars_kernel(float *__restrict__ s2) { s2[0] = 0.f; s2[1] = __builtin_inff(); }
int vq_unquantized_scalars(float *scalars2, hipStream_t stream) {     // diff = 0, perplexity = inf (bottleneck.py:107-119)
  if (!scalars2) return invalid("vq_unquantized_scalars: null pointer");
  hipLaunchKernelGGL(vq_unquantized_scalars_kernel, dim3(1), dim3(1), 0, stream, scalars2);
  return check_launch("vq_unquantized_scalars");
}
int vq_zero_counts(int32_t *counts, int K, hipStream_t stream) {
  if (!counts || K <= 0) return invalid("vq_zero_counts: bad argument");
  hipLaunchKernelGGL(vq_zero_counts_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, counts, K);
  return check_launch("vq_zero_counts");
}

int vq_finalize_f32(const float *sse_part, int n_part, const int32_t *counts, int K, int64_t N,
                    int D, float *out2, hipStream_t stream) {
  if (!sse_part || !counts || !out2 || n_part <= 0 || K <= 0 || N <= 0 || D <= 0)
    return invalid("vq_finalize: bad argument");
  hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, stream, sse_part, n_part, counts, K,
                     N, D, out2);
  return check_launch("vq_finalize_f32");
}

int vq_finalize2_f32(const float *sse_t, int np_t, const int32_t *counts_t, int K_t, int64_t N_t, const float *sse_b, int np_b,
                     const int32_t *counts_b, int K_b, int64_t N_b, int D, float *out5, hipStream_t stream) {
  if (!sse_t || !counts_t || !sse_b || !counts_b || !out5 || np_t <= 0 || np_b <= 0 || K_t <= 0 || K_b <= 0 || N_t <= 0 ||
      N_b <= 0 || D <= 0)
    return invalid("vq_finalize2: bad argument");
  hipLaunchKernelGGL(vq_finalize2_kernel, dim3(1), dim3(512), 0, stream, sse_t, np_t, counts_t, K_t, N_t, sse_b, np_b, counts_b,
                     K_b, N_b, D, out5);
  return check_launch("vq_finalize2_f32");
}
int vq_scalars_sum_f32(float *out5, hipStream_t stream) {
  hipLaunchKernelGGL(vq_scalars_sum_kernel, dim3(1), dim3(1), 0, stream, out5);
  return check_launch("vq_scalars_sum_f32");
}

int embed_code_f32(const int64_t *idx, const float *codes, float *out, int64_t N, int D, int K,
                   hipStream_t stream) {
  if (!idx || !codes || !out || N <= 0 || D <= 0 || (D % 4) != 0 || K <= 0)
    return invalid("embed_code: bad argument");
  const int64_t total = N * (D / 4);
  hipLaunchKernelGGL(embed_code_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     idx, codes, out, N, D / 4, K);
  return check_launch("embed_code_f32");
}

}  // namespace isi
