// Small fp32 operators of the transformer prior (gfx950): LayerNorm with fused
// residual add, and a "few rows" linear layer for single-token decoding.
// They implement the non-attention parts of the layers specified in
// oracle/prior_oracle.py (the reference reaches them through the absent
// VQCPCB package, priors/transformer.py:370-417).
#include "isi_common.h"
#include "isi_internal.h"

namespace isi {

// out[m,:] = LayerNorm(drop(x[m,:]) + res[m,:]) * gamma + beta ; one wave per row.  drop = inverted dropout with the
// keep mask dropout_keep(seed, m D + c) (isi_internal.h: a hash, nothing stored); drop_thresh = 0: the identity.
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ res,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            float *__restrict__ out, int M, int D, float eps,
                                                            unsigned drop_thresh, float drop_scale, uint64_t seed0,
                                                            const uint64_t *seed_base) {
  const uint64_t seed = seed0 + (drop_thresh && seed_base ? *seed_base : 0);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float *xr = x + (size_t)row * D;
  const float *rr = res ? res + (size_t)row * D : nullptr;
  constexpr int MAXV = 8;  // D <= 64 * 4 * MAXV = 2048
  float4 v[MAXV];
  float sum = 0.f;
  const int nq = D >> 2;  // float4 per row
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int qd = lane + 64 * i;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qd < nq) {
      t = reinterpret_cast<const float4 *>(xr)[qd];
      if (drop_thresh) {
        const unsigned e0 = (unsigned)row * (unsigned)D + 4u * qd;
        t.x = dropout_keep(seed, e0, drop_thresh) ? t.x * drop_scale : 0.f;
        t.y = dropout_keep(seed, e0 + 1, drop_thresh) ? t.y * drop_scale : 0.f;
        t.z = dropout_keep(seed, e0 + 2, drop_thresh) ? t.z * drop_scale : 0.f;
        t.w = dropout_keep(seed, e0 + 3, drop_thresh) ? t.w * drop_scale : 0.f;
      }
      if (rr) {
        const float4 u = reinterpret_cast<const float4 *>(rr)[qd];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      sum += (t.x + t.y) + (t.z + t.w);
    }
    v[i] = t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)D;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      var += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = 1.0f / sqrtf(var / (float)D + eps);
  float *orow = out + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      const float4 g = reinterpret_cast<const float4 *>(gamma)[qd];
      const float4 bb = reinterpret_cast<const float4 *>(beta)[qd];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + bb.x;
      o.y = (v[i].y - mean) * rstd * g.y + bb.y;
      o.z = (v[i].z - mean) * rstd * g.z + bb.z;
      o.w = (v[i].w - mean) * rstd * g.w + bb.w;
      reinterpret_cast<float4 *>(orow)[qd] = o;
    }
  }
}

static bool dropout_args(float p, int64_t M, int D, unsigned *thresh, float *scale) {
  *thresh = 0; *scale = 1.f;
  if (p == 0.f) return true;
  if (!(p > 0.f && p < 1.f) || M * D > ((int64_t)1 << 32)) return false;
  const double t = (double)p * 4294967296.0;
  *thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (t < 1.0 ? 1u : (unsigned)t);
  *scale = 1.f / (1.f - p);
  return true;
}

int layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta, float *out,
                  int64_t M, int D, float eps, hipStream_t stream, float drop_p, uint64_t drop_seed) {
  if (!x || !gamma || !beta || !out || M <= 0 || D <= 0) return invalid("layernorm: bad argument");
  unsigned thresh; float scale;
  if (!dropout_args(drop_p, M, D, &thresh, &scale)) return invalid("layernorm: dropout needs 0 <= p < 1 and fewer than 2^32 elements");
  if ((D & 3) || D > 2048) return unsupported("layernorm: need D % 4 == 0 and D <= 2048");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(gamma) |
       reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(out)) & 15)
    return invalid("layernorm: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, res, gamma,
                     beta, out, (int)M, D, eps, thresh, scale, drop_seed, dropout_seed_base());
  return check_launch("layernorm_f32");
}

// out[m, n] = [relu]( x[m,:] . W[n,:] + bias[n] + res[m,n] ) for a handful of rows
// m < MR (single-token decoding): one wave per output feature n streams W[n,:]
// (torch layout [N,K], K contiguous) once with 16-B loads; the MR activations
// rows come from L2.  Weight-bandwidth bound by construction.
template <int MR>
__global__ __launch_bounds__(256) void linear_rows_f32_kernel(const float *__restrict__ x, int x_stride,
                                                              const float *__restrict__ W,
                                                              const float *__restrict__ bias,
                                                              const float *__restrict__ res, int res_stride,
                                                              float *__restrict__ out, int out_stride, int M,
                                                              int N, int K, int relu) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float4 *wr = reinterpret_cast<const float4 *>(W + (size_t)n * K);
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const int nq = K >> 2;
  for (int qd = lane; qd < nq; qd += 64) {
    const float4 w = wr[qd];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) {
        const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)m * x_stride)[qd];
        acc[m] += (w.x * xv.x + w.y * xv.y) + (w.z * xv.z + w.w * xv.w);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o);
  }
  if (lane == 0) {
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) {
        float v = acc[m] + b;
        if (res) v += res[(size_t)m * res_stride + n];
        if (relu) v = fmaxf(v, 0.f);
        out[(size_t)m * out_stride + n] = v;
      }
    }
  }
}

int linear_rows_f32(const float *x, int x_stride, const float *W, const float *bias, const float *res,
                    int res_stride, float *out, int out_stride, int M, int N, int K, int relu,
                    hipStream_t stream) {
  if (!x || !W || !out || M <= 0 || N <= 0 || K <= 0) return invalid("linear_rows: bad argument");
  if (M > 8) return unsupported("linear_rows: at most 8 rows (use isi_conv2d_f32 as a GEMM beyond)");
  if ((K & 3) || (x_stride & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15))
    return invalid("linear_rows: K and x_stride must be multiples of 4, pointers 16-byte aligned");
  dim3 grid((N + 3) / 4), block(256);
  if (M <= 1)
    hipLaunchKernelGGL(linear_rows_f32_kernel<1>, grid, block, 0, stream, x, x_stride, W, bias, res, res_stride,
                       out, out_stride, M, N, K, relu);
  else if (M <= 4)
    hipLaunchKernelGGL(linear_rows_f32_kernel<4>, grid, block, 0, stream, x, x_stride, W, bias, res, res_stride,
                       out, out_stride, M, N, K, relu);
  else
    hipLaunchKernelGGL(linear_rows_f32_kernel<8>, grid, block, 0, stream, x, x_stride, W, bias, res, res_stride,
                       out, out_stride, M, N, K, relu);
  return check_launch("linear_rows_f32");
}


// ------------------------------------------------------------------ decoding
// One query row per (batch, head) against Sk cached keys / values with the same
// relative-position logits as rel_attention_f32_kernel:
//   s_j = (q.k_j + q.e[h, floor(q_pos/Cq) - floor(j/Ck) + Ek - 1]) * scale ; out = softmax(s) V
// Bandwidth bound (each K, V and e row is read once): VALU dot products, no MFMA.
// NG = 2 (round 6): BOTH key splits of a two-split launch in one workgroup of 512 threads -- threads [256 g, 256 g + 256) are
// split g -- and the merge of rel_attention_combine_kernel (same operations, same order) at its end: the launch of that
// kernel (4.7 us + a dispatch gap, 16 times per decoding step of a batch of 17 .. 63 sequences) and the round trip of the
// partials through memory go away.  The two halves take the same path (by the split length, not by their own key count)
// and an empty half walks it with zero keys: every barrier is met by all 512 threads.
template <int HD, int NG = 1>
__global__ __launch_bounds__(256 * NG) void rel_attention_decode_f32_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ e, float *__restrict__ out, int Sk, int64_t q_sb, int64_t q_sh, int64_t k_ss,
    int64_t k_sb, int64_t k_sh, int64_t v_ss, int64_t v_sb, int64_t v_sh, int64_t o_sb, int64_t o_sh,
    int q_pos, int Cq, int Ck, int Ek, int R, float scale, int chunk, float *__restrict__ partial,
    const int *__restrict__ pos, int self_keys) {
  // (every argument "used" here: the compiler otherwise fetches them in four dependent scalar-memory round trips ahead of
  // the first key request -- prior_decode.hip: touch_args)
  asm volatile("" ::"s"(q), "s"(k), "s"(v), "s"(e), "s"(out), "s"(Sk), "s"(q_sb), "s"(q_sh), "s"(k_ss), "s"(k_sb), "s"(k_sh),
               "s"(v_ss), "s"(v_sb), "s"(v_sh));
  asm volatile("" ::"s"(o_sb), "s"(o_sh), "s"(q_pos), "s"(Cq), "s"(Ck), "s"(Ek), "s"(R), "s"(scale), "s"(chunk), "s"(partial),
               "s"(pos), "s"(self_keys));
  // Replayable form (hipGraph): the position comes from device memory; for self-attention the key
  // count is position + 1 and the (fixed) number of splits shares it evenly.
  const int nsplit = NG == 2 ? 2 : (int)gridDim.z;
  if (pos) {
    q_pos = *pos;
    if (self_keys) {
      Sk = q_pos + 1;
      chunk = (Sk + nsplit - 1) / nsplit;
    }
  }
  // blockIdx.z = key split: this workgroup handles keys [z*chunk, min(Sk, (z+1)*chunk)) and,
  // when there are several splits, writes an un-normalised partial (o, max, sum) that
  // rel_attention_combine_kernel merges.
  // A group of G = HD/4 lanes owns one key row (one coalesced 16-B load per lane);
  // 256/G rows are in flight per pass, 4 passes unrolled.
  constexpr int G = HD / 4;        // lanes per row: 4, 8 or 16
  constexpr int RPP = 256 / G;     // rows per pass
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int kg = NG == 2 ? (int)(threadIdx.x >> 8) : 0;          // key split of this half of the workgroup (NG = 2)
  const int split = NG == 2 ? kg : (int)blockIdx.z;
  const int sc_len = (chunk + 3) & ~3;
  float *red = sm + kg * (8 + RPP * HD + sc_len);                  // [8]
  float *part = red + 8;           // [RPP][HD] partial outputs
  float *sc = part + RPP * HD;     // [Sk]
  float *mg = sm + NG * (8 + RPP * HD + sc_len);                   // NG = 2: [2][HD + 2] the halves' (o, max, sum)
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G, gl = tid % G;
  const int h = blockIdx.x, b = blockIdx.y;
  const int kbeg = split * chunk;
  k += (size_t)kbeg * k_ss;
  v += (size_t)kbeg * v_ss;
  const int key0 = kbeg;              // absolute index of local key 0 (relative positions)
  Sk = max(min(Sk - kbeg, chunk), 0); // local key count
  if (NG == 1 && Sk <= 0) {           // empty split (only with a fixed split count): neutral partial
    if (gridDim.z > 1 && tid < HD) {
      float *pp = partial + (((size_t)b * gridDim.x + h) * gridDim.z + blockIdx.z) * (HD + 4);
      pp[tid] = 0.f;
      if (tid == 0) { pp[HD] = -1e30f; pp[HD + 1] = 0.f; }
    }
    return;
  }
  const float4 qq = *reinterpret_cast<const float4 *>(q + b * q_sb + h * q_sh + gl * 4);
  const int evq = q_pos / Cq;
  const float *kb = k + b * k_sb + h * k_sh + gl * 4;
  const float *vb = v + b * v_sb + h * v_sh + gl * 4;
  const float *eb = e ? e + (size_t)h * R * HD + gl * 4 : nullptr;
  float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f);
  float gmax, gsum;
  constexpr int UMAX = 9;          // rows per lane group held in registers: splits of up to 9 * RPP keys
  if ((NG == 2 ? chunk : Sk) <= UMAX * RPP) {
    // ---- short split (the decoding loop's 128-key splits): K, V and relative rows are all requested up front -- ONE
    // memory round trip instead of the score pass followed by the value pass (the kernel is a chain of dependent
    // latencies, not of bandwidth: 10 -> ~8 us per launch, 16 launches per position)
    float4 kk[UMAX], vv[UMAX], ee[UMAX];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = grp + u * RPP;
      const int jc = j < Sk ? j : 0;
      if (NG == 2 && Sk == 0) {        // (an empty half reads nothing: its rows may hold anything, 0 x NaN included)
        kk[u] = vv[u] = ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      kk[u] = *reinterpret_cast<const float4 *>(kb + (size_t)jc * k_ss);
      vv[u] = *reinterpret_cast<const float4 *>(vb + (size_t)jc * v_ss);
      ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (eb) {
        int r = evq - (key0 + jc) / Ck + Ek - 1;
        r = r < 0 ? 0 : (r >= R ? R - 1 : r);
        ee[u] = *reinterpret_cast<const float4 *>(eb + (size_t)r * HD);
      }
    }
    float sv[UMAX];
    float lmax = -1e30f;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const float4 kq = make_float4(kk[u].x + ee[u].x, kk[u].y + ee[u].y, kk[u].z + ee[u].z, kk[u].w + ee[u].w);
      float acc = (qq.x * kq.x + qq.y * kq.y) + (qq.z * kq.z + qq.w * kq.w);
      acc = G == 16 ? row16_sum(acc) : G == 8 ? group8_sum(acc) : group4_sum(acc);   // over the row's G lanes
      acc *= scale;
      sv[u] = acc;
      if (grp + u * RPP < Sk) lmax = fmaxf(lmax, acc);
    }
    lmax = wave64_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const float pj = grp + u * RPP < Sk ? __expf(sv[u] - gmax) : 0.f;
      if (gl == 0) lsum += pj;
      o0.x += pj * vv[u].x; o0.y += pj * vv[u].y; o0.z += pj * vv[u].z; o0.w += pj * vv[u].w;
    }
    lsum = wave64_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    gsum = (red[4] + red[5]) + (red[6] + red[7]);
  } else {
  // ---- long split (sequences beyond 8 x 144 keys): scores, then values, 16 rows of a lane group in flight per step
  // (with 4 the two passes of a 513-key split were 18 dependent round trips)
  constexpr int UB = 16;
  float lmax = -1e30f;
  for (int j0 = grp; j0 < Sk; j0 += UB * RPP) {
    float4 kk[UB], ee[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = j0 + u * RPP;
      const int jc = j < Sk ? j : j0;
      kk[u] = *reinterpret_cast<const float4 *>(kb + (size_t)jc * k_ss);
      ee[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (eb) {
        int r = evq - (key0 + jc) / Ck + Ek - 1;
        r = r < 0 ? 0 : (r >= R ? R - 1 : r);
        ee[u] = *reinterpret_cast<const float4 *>(eb + (size_t)r * HD);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = j0 + u * RPP;
      const float4 kq = make_float4(kk[u].x + ee[u].x, kk[u].y + ee[u].y, kk[u].z + ee[u].z, kk[u].w + ee[u].w);
      float acc = (qq.x * kq.x + qq.y * kq.y) + (qq.z * kq.z + qq.w * kq.w);
      acc = G == 16 ? row16_sum(acc) : G == 8 ? group8_sum(acc) : group4_sum(acc);
      if (j < Sk) {
        acc *= scale;
        if (gl == 0) sc[j] = acc;
        lmax = fmaxf(lmax, acc);
      }
    }
  }
  lmax = wave64_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int j = tid; j < Sk; j += 256) {
    const float pj = __expf(sc[j] - gmax);
    sc[j] = pj;
    lsum += pj;
  }
  lsum = wave64_sum(lsum);
  if (lane == 0) red[4 + wave] = lsum;
  __syncthreads();
  gsum = (red[4] + red[5]) + (red[6] + red[7]);
  // ---- out = sum_j p_j v_j
  for (int j0 = grp; j0 < Sk; j0 += UB * RPP) {
    float4 vv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = j0 + u * RPP;
      vv[u] = *reinterpret_cast<const float4 *>(vb + (size_t)(j < Sk ? j : j0) * v_ss);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = j0 + u * RPP;
      const float pj = j < Sk ? sc[j] : 0.f;
      o0.x += pj * vv[u].x; o0.y += pj * vv[u].y; o0.z += pj * vv[u].z; o0.w += pj * vv[u].w;
    }
  }
  }
  // the wave's 64 / G lane groups are added up in registers (lane l <-> l ^ o for o = G .. 32), then one row per wave
  // goes through LDS: four reads per output instead of a chain of 256 / G
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    o0.x += __shfl_xor(o0.x, o); o0.y += __shfl_xor(o0.y, o); o0.z += __shfl_xor(o0.z, o); o0.w += __shfl_xor(o0.w, o);
  }
  if (lane < G) *reinterpret_cast<float4 *>(part + wave * HD + lane * 4) = o0;
  __syncthreads();
  if (NG == 2) {                       // the two halves' partials meet in LDS: rel_attention_combine_kernel's merge
    if (tid < HD) {
      mg[kg * (HD + 2) + tid] = (part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid]);
      if (tid == 0) { mg[kg * (HD + 2) + HD] = gmax; mg[kg * (HD + 2) + HD + 1] = gsum; }
    }
    __syncthreads();
    if (kg == 0 && tid < HD) {
      float M = -1e30f;
      for (int s_ = 0; s_ < 2; ++s_) M = fmaxf(M, mg[s_ * (HD + 2) + HD]);
      float num = 0.f, den = 0.f;
      for (int s_ = 0; s_ < 2; ++s_) {
        const float w = __expf(mg[s_ * (HD + 2) + HD] - M);
        num += w * mg[s_ * (HD + 2) + tid];
        den += w * mg[s_ * (HD + 2) + HD + 1];
      }
      out[b * o_sb + h * o_sh + tid] = num * (1.0f / den);
    }
    return;
  }
  if (tid < HD) {
    const float acc = (part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid]);
    if (gridDim.z == 1) {
      out[b * o_sb + h * o_sh + tid] = acc / gsum;
    } else {
      float *pp = partial + (((size_t)b * gridDim.x + h) * gridDim.z + blockIdx.z) * (HD + 4);
      pp[tid] = acc;
      if (tid == 0) { pp[HD] = gmax; pp[HD + 1] = gsum; }
    }
  }
}

// Merge the key splits of one (batch, head): softmax-weighted sum of the partials.
__global__ void rel_attention_combine_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                             int HD, int NS, int64_t o_sb, int64_t o_sh) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const float *pp = partial + ((size_t)b * gridDim.x + h) * NS * (HD + 4);
  // (up to eight splits: every value requested before the first is used -- the two loops below were two chains of dependent
  // loads, ~3 memory round trips in a launch that does nothing else; splits beyond NS re-read the last one and are left out)
  if (NS <= 8) {
    float pm[8], pl[8], pv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int sc = s < NS ? s : NS - 1;
      pm[s] = pp[sc * (HD + 4) + HD];
      pl[s] = pp[sc * (HD + 4) + HD + 1];
      pv[s] = pp[sc * (HD + 4) + d];
    }
    float M = -1e30f;
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < NS) M = fmaxf(M, pm[s]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < NS) {
        const float w = __expf(pm[s] - M);
        num += w * pv[s];
        den += w * pl[s];
      }
    }
    out[b * o_sb + h * o_sh + d] = num * (1.0f / den);
    return;
  }
  float M = -1e30f;
  for (int s = 0; s < NS; ++s) M = fmaxf(M, pp[s * (HD + 4) + HD]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < NS; ++s) {
    const float w = __expf(pp[s * (HD + 4) + HD] - M);   // (row_gemv1_kernel merges with the same arithmetic)
    num += w * pp[s * (HD + 4) + d];
    den += w * pp[s * (HD + 4) + HD + 1];
  }
  out[b * o_sb + h * o_sh + d] = num * (1.0f / den);
}

// key splits of one (batch, head) pair: 128 keys each, at most 8 -- and no more than it takes to put ~512 workgroups on the
// chip: a batch of 32 sequences x 8 heads already fills it (2 splits instead of 8: 22.7 + 5.4 us -> see DESIGN.md section 7,
// batched decoding), a single sequence takes all 8
int rel_attention_decode_splits(int Sk, int pairs) {
  const int by_keys = Sk <= 192 ? 1 : (Sk + 127) / 128 > 8 ? 8 : (Sk + 127) / 128;
  const int by_chip = pairs <= 0 ? 8 : (512 + pairs - 1) / pairs;
  return by_keys < by_chip ? by_keys : (by_chip < 1 ? 1 : by_chip);
}

size_t rel_attention_decode_workspace_floats(int B, int H, int head_dim) {
  return (size_t)B * H * 8 * (head_dim + 4);
}

int rel_attention_decode_f32(const isi_attn_args *g, int q_pos, float *workspace, hipStream_t stream) {
  return rel_attention_decode_pos_f32(g, q_pos, nullptr, 0, workspace, stream);
}

// pos != nullptr: the query position is read from device memory at run time (replayable launch); with
// self_keys the key count is position + 1 and g->Sk is its upper bound (it fixes grid and LDS sizes).
int rel_attention_decode_pos_f32(const isi_attn_args *g, int q_pos, const int *pos, int self_keys, float *workspace,
                                 hipStream_t stream) {
  return rel_attention_decode_launch(g, q_pos, pos, self_keys, workspace, /*combine*/ 1, stream);
}

// combine = 0: with several key splits the partials ([B, H, splits, head_dim + 4]: un-normalised output, then the
// split's maximum and sum) stay in `workspace` for the caller to merge (the decoding loop merges them in the prologue
// of the out-projection: one dependent launch less); g->out is then not written.
int rel_attention_decode_launch(const isi_attn_args *g, int q_pos, const int *pos, int self_keys, float *workspace,
                                int combine, hipStream_t stream) {
  if (!g || !g->q || !g->k || !g->v || !g->out) return invalid("attention_decode: null pointer");
  if (g->Sk <= 0 || g->B <= 0 || g->H <= 0 || g->Cq <= 0 || g->Ck <= 0) return invalid("attention_decode: bad shape");
  if (g->Sk > 65536) return unsupported("attention_decode: more than 65536 keys");
  const int ns = workspace ? rel_attention_decode_splits(g->Sk, g->B * g->H) : 1;
  // self_keys with the position by value: position + 1 keys, shared by the split count of the upper bound g->Sk
  // (the same split as the replayable form derives on the device)
  const int Sk = (self_keys && !pos) ? q_pos + 1 : g->Sk;
  if (Sk > g->Sk) return invalid("attention_decode: position beyond the key capacity");
  const int chunk = (Sk + ns - 1) / ns;
  // two splits that are merged right away: both in one workgroup, no partials, no combine launch (NG = 2)
  // (as long as every (batch, head) pair finds a CU of its own: the 512-thread workgroup holds 159 registers per lane, ONE fits a
  // CU where three of the 256-thread ones do -- at 48 sequences x 8 heads the launch became a round and a half, 35.5 against
  // 37.2 k codes/s)
  const bool both = ns == 2 && combine && !knobs().decode_attn_separate_splits && g->B * g->H <= current_device_cu_count();
  const size_t per_group = 8 + 256 * 4 + (((g->Sk + ns - 1) / ns + 3) & ~3);          // red + part[256/G][HD] + scores
  const size_t smem = (both ? 2 * per_group + 2 * (size_t)(g->head_dim + 2) : per_group) * sizeof(float);
  dim3 grid(g->H, g->B, both ? 1 : ns), block(both ? 512 : 256);
#define ISI_DEC(HD)                                                                                         \
  do {                                                                                                      \
    auto kern = both ? rel_attention_decode_f32_kernel<HD, 2> : rel_attention_decode_f32_kernel<HD, 1>;     \
    if (smem > 48 * 1024 &&                                                                                 \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)smem) != hipSuccess)                                                       \
      return check_launch("hipFuncSetAttribute(attention_decode)");                                         \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, g->q, g->k, g->v, g->rel_embeddings, g->out, Sk,    \
                       g->q_sb, g->q_sh, g->k_ss, g->k_sb, g->k_sh, g->v_ss, g->v_sb, g->v_sh, g->o_sb,     \
                       g->o_sh, q_pos, g->Cq, g->Ck, g->Ek, g->rel_rows, g->scale, chunk, workspace, pos,  \
                       self_keys);                                                                          \
  } while (0)
  switch (g->head_dim) {
    case 16: ISI_DEC(16); break;
    case 32: ISI_DEC(32); break;
    case 64: ISI_DEC(64); break;
    default: return unsupported("attention_decode: head_dim must be 16, 32 or 64");
  }
#undef ISI_DEC
  int rc = check_launch("rel_attention_decode_f32");
  if (rc || ns == 1 || !combine || both) return rc;
  hipLaunchKernelGGL(rel_attention_combine_kernel, dim3(g->H, g->B), dim3(g->head_dim), 0, stream, workspace,
                     g->out, g->head_dim, ns, g->o_sb, g->o_sh);
  return check_launch("rel_attention_combine");
}

// ------------------------------------------------------------------ sampling
// One categorical draw per row from logits[row, 0:n] (sample.py:286-295 of the
// reference: temperature, top_k_top_p_filtering (sample.py:36-65), softmax,
// multinomial), with a host-supplied uniform u[row] in [0,1) replacing torch's
// RNG stream: the sample is the first index whose inclusive cumulative
// probability exceeds u * total.  One workgroup per row; n <= 1024.
//
// Round 3: the kernel was 15 us of a 290 us token -- a 45-stage bitonic sort and two Hillis-Steele scans, ~90 block
// barriers.  Now (a) no filter (top_k = 0, top_p = 0: the reference's plain multinomial) sorts nothing; (b) the sort
// runs its stages with partner distance < 64 inside a wave (`__shfl_xor`, no barrier: 39 of the 45 stages at n = 512)
// and only the 6 long-distance stages through LDS; (c) the prefix sums are wave scans (`__shfl_up`) plus one exchange
// of the wave totals: 2 barriers each.
__device__ __forceinline__ float block_inclusive_scan(float v, float *wave_tot, float *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  if (lane == 63) wave_tot[wave] = v;
  __syncthreads();
  float pre = 0.f, tot = 0.f;
  for (int w = 0; w < nw; ++w) {       // fixed order: the same sums on every thread
    const float t = wave_tot[w];
    if (w < wave) pre += t;
    tot += t;
  }
  __syncthreads();                     // wave_tot may be reused by the caller's next scan
  *total = tot;
  return pre + v;
}

__global__ __launch_bounds__(1024) void sample_row_f32_kernel(const float *__restrict__ logits, int stride,
                                                              int n, float inv_temperature, int top_k,
                                                              float top_p, const float *__restrict__ u,
                                                              int64_t *__restrict__ out,
                                                              float *__restrict__ filtered,
                                                              const int *__restrict__ pos, int pos_off,
                                                              const SampleCommit cm) {
  if (pos) u += (size_t)(*pos - pos_off) * gridDim.x;  // replayable launch: this token's uniforms
  __shared__ float val[1024];
  __shared__ int idx[1024];
  __shared__ int keep[1024];
  __shared__ float wave_tot[16];
  __shared__ float sh_f[2];
  __shared__ int sh_i;
  const int tid = threadIdx.x, np = blockDim.x, row = blockIdx.x;
  const float NEGI = -INFINITY;
  const float lg = tid < n ? logits[(size_t)row * stride + tid] * inv_temperature : NEGI;
  float vmax;
  bool kp = tid < n;
  if (top_k > 0 || top_p > 0.f) {
    // ---- bitonic sort, descending by value (ties: lower index first); element t lives in thread t
    float v = lg;
    int ix = tid;
    for (int k = 2; k <= np; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        float c;
        int ic;
        if (j >= 64) {                 // partner in another wave: through LDS
          val[tid] = v; idx[tid] = ix;
          __syncthreads();
          c = val[tid ^ j]; ic = idx[tid ^ j];
          __syncthreads();
        } else {
          c = __shfl_xor(v, j); ic = __shfl_xor(ix, j);
        }
        const bool lower = (tid & j) == 0;                 // this thread keeps the element that comes FIRST of the pair
        const bool desc = (tid & k) == 0;                  // ... in a descending (ascending) run
        const bool mine_first = v > c || (v == c && ix < ic);
        const bool take_mine = (lower == desc) ? mine_first : !mine_first;
        if (!take_mine) { v = c; ix = ic; }
      }
    }
    // sorted position tid = (v, ix)
    val[tid] = v;
    __syncthreads();
    vmax = val[0];
    // top-k: drop everything strictly below the k-th largest value
    const float kth = top_k > 0 ? val[min(top_k, n) - 1] : NEGI;
    const float sv = (top_k > 0 && v < kth) ? NEGI : v;
    bool removed = sv == NEGI;
    __syncthreads();                   // val is reused below
    if (top_p > 0.f) {
      // top-p on the sorted, top-k-filtered row: remove position s when the cumulative probability up to s - 1
      // already exceeds top_p
      const float ex = sv == NEGI ? 0.f : expf(sv - vmax);
      float total;
      const float inc = block_inclusive_scan(ex, wave_tot, &total);
      val[tid] = inc;
      __syncthreads();
      if (tid > 0 && val[tid - 1] / total > top_p) removed = true;
      __syncthreads();
    }
    keep[ix] = removed ? 0 : 1;
    __syncthreads();
    kp = tid < n && keep[tid];
  } else {
    // no filter: the maximum is all the sort was needed for
    float m = lg;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) wave_tot[tid >> 6] = m;
    __syncthreads();
    m = wave_tot[0];
    for (int w = 1; w < (np + 63) / 64; ++w) m = fmaxf(m, wave_tot[w]);
    __syncthreads();
    vmax = m;
  }
  // probabilities in index order, inverse-CDF draw
  const float fl = kp ? lg : NEGI;
  if (filtered && tid < n) filtered[(size_t)row * n + tid] = fl;
  const float pe = kp ? expf(lg - vmax) : 0.f;
  float total;
  const float inc = block_inclusive_scan(pe, wave_tot, &total);
  if (tid == 0) sh_i = n - 1;
  __syncthreads();
  if (tid < n && inc > u[row] * total) atomicMin(&sh_i, tid);
  __syncthreads();
  if (tid == 0) out[row] = sh_i;
  // the decoding loop's commit, in the same launch: code row, and the token's embedding into the next input row
  if (cm.table) {
    const int p = pos ? *pos : cm.p_value;
    const int tok = sh_i;
    if (tid == 0) cm.codes[(size_t)row * cm.codes_stride + (p - cm.i_off)] = tok;
    if (p + 1 < cm.S_t) {
      float *x_next = cm.x_seq + ((size_t)(p + 1) * gridDim.x + row) * cm.x_stride;
      for (int e = tid; e < cm.eff; e += np) x_next[e] = cm.table[(size_t)tok * cm.eff + e];
    }
    if (cm.advance) {                  // (gridDim.x == 1, checked by the launcher)
      __syncthreads();                 // every thread has read the counter
      if (tid == 0) *cm.advance = p + 1;
    }
  }
  (void)sh_f;
}

int sample_row_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                   const float *u, int64_t *out, float *filtered, hipStream_t stream) {
  return sample_row_pos_f32(logits, stride, rows, n, temperature, top_k, top_p, u, out, filtered, nullptr, 0, stream);
}

// pos != nullptr: row r draws with u[(*pos - pos_off) * rows + r]
int sample_row_pos_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                       const float *u, int64_t *out, float *filtered, const int *pos, int pos_off,
                       hipStream_t stream) {
  SampleCommit none;
  memset(&none, 0, sizeof none);
  return sample_row_commit_f32(logits, stride, rows, n, temperature, top_k, top_p, u, out, filtered, pos, pos_off, none, stream);
}

int sample_row_commit_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                          const float *u, int64_t *out, float *filtered, const int *pos, int pos_off,
                          const SampleCommit &cm, hipStream_t stream) {
  if (!logits || !u || !out || rows <= 0 || n <= 0 || temperature <= 0.f) return invalid("sample_row: bad argument");
  if (n > 1024) return unsupported("sample_row: at most 1024 classes");
  if (cm.advance && (rows != 1 || !cm.table)) return invalid("sample_row: the position counter is advanced by a one-row commit only");
  int np = 64;
  while (np < n) np <<= 1;
  hipLaunchKernelGGL(sample_row_f32_kernel, dim3(rows), dim3(np), 0, stream, logits, stride, n, 1.0f / temperature,
                     top_k, top_p, u, out, filtered, pos, pos_off, cm);
  return check_launch("sample_row_f32");
}

}  // namespace isi
