// Internal (C++-linkage) entry points behind the extern "C" API of isi_api.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "isi_hip.h"

namespace isi {

int split_conv_weight_f16(const float *packed, float *out, int64_t n_floats, hipStream_t stream);
int pad_channels4_f32(const float *x, float *out, int B, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                      hipStream_t st);
int add_gate_rows_f32(float *out, const float *a, int64_t lda, const float *b, const float *y, int64_t M, int C, hipStream_t st);
int vq_bwd_rows_f32(float *dz, const float *dq, int64_t ldq, const float *z, const float *q_st, const float *g_diff, int64_t M,
                    int D, hipStream_t st);
int pack_multi(const void *table_dev, int n, int blocks_per_entry, hipStream_t stream);
bool conv_wgrad_halo_route(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, int OH, int OW);
int reduce_jobs_f32(const isi_reduce_job *jobs, int n_jobs, hipStream_t stream);
int conv_wgrad_deferred_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw, int cin_keep, float *db,
                            float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW, int stride,
                            int pad, int flags, hipStream_t stream, isi_reduce_job *jobs_out, int *n_jobs);
int pair_encode_f32(const float *x, float *out, int64_t n, hipStream_t stream);
int pair_decode_f32(const float *in, float *x, int64_t n, hipStream_t stream);
bool conv_first_applicable(const isi_src *s0, const isi_src *s1, const isi_src *res, const isi_dst *dst, int Cout,
                           int KH, int KW, int stride, int pad, int OH, int OW, int nz);
int conv_first_f32(const isi_src *s0, const float *packed_w, const float *bias, const isi_dst *dst, int B, int H,
                   int W, int Cout, int OH, int OW, int64_t in_extent, int flags, hipStream_t stream, float *twin = nullptr,
                   const float *gate = nullptr);
bool conv_pair_sources_ok(int C0, int C1, int Cout, int taps);
// conv_pair_f16.hip: the LDS-DMA implicit-GEMM kernel of the pair pipeline (pair8 sources, blocked W16 weights)
struct PairConvArgs {
  const float *in0, *in1, *w16, *bias;
  float *out;
  unsigned in0_bytes, in1_bytes, w_bytes, out_bytes;
  int C0, C1;
  int s0n, s0h, s0w, s1n, s1h, s1w;    // source element strides (channel stride 1)
  int on, oh, ow;                      // output strides in GEMM-grid pixels (channel stride 1)
  int H, W, OH, OW, Cout, Kpad, KH, KW, stride, pad, relu, M;
  int convT, w_phase_stride, dst_sh, dst_sw;
  int out_pair;
  float *twin;                         // with out_pair: dense channels-last fp32 copy of the output (training tape), or null
};
bool conv_pair_kernel_ok(int C0, int C1, int Cout, int taps);
int conv_pair_f16(const PairConvArgs &c, hipStream_t stream);
int conv_pair_debug_stamps(long long *host, int n);
// convT_pair_f16.hip: ConvTranspose2d(k4,s2,p1) of the pair pipeline with the output phases fused on one staged tile
bool convT_pair_ok(int Cin, int Cout);
bool decoder_tail_ok(int Cin, int Cmid, int Cout);
int decoder_tail_f32(const float *in_pair, const float *packed_w1, const float *bias1, const float *packed_w2,
                     const float *bias2, float *yprime_ws, const isi_dst *dst, int B, int H, int W, int Cin, int Cmid,
                     int Cout, hipStream_t stream);
int convT_pair_debug_stamps(long long *host, int n);
int convT_pair_f16(const float *in, const float *w16, const float *bias, float *out, int B, int H, int W, int Cin,
                   int Cout, int relu, int out_pair, hipStream_t stream, const float *w2 = nullptr, int n2 = 0,
                   float *twin = nullptr);
int conv2d_f32(const isi_src *s0, const isi_src *s1, const float *packed_w, const float *bias,
               const isi_src *res, const isi_dst *dst, int B, int H, int W, int Cout, int KH,
               int KW, int stride, int pad, int relu, hipStream_t stream, const float *gate = nullptr,
               float *twin = nullptr);
int conv2d_batched_f32(const isi_src *s0, const isi_src *s1, const float *packed_w, const float *bias,
                       const isi_src *res, const isi_dst *dst, int B, int H, int W, int Cout, int KH,
                       int KW, int stride, int pad, int relu, int nz, int64_t zs_in0, int64_t zs_w,
                       int64_t zs_res, int64_t zs_out, hipStream_t stream, const float *gate = nullptr,
                       float *twin = nullptr);
int conv_transpose2d_k4s2_f32(const isi_src *s, const float *packed_w, const float *bias,
                              const isi_dst *dst, int B, int H, int W, int Cout, int relu,
                              hipStream_t stream, const float *gate = nullptr, float *twin = nullptr);

bool resblock_fusable(int C, int R);
bool resblock_pair_ok(int C, int R);
bool resblock_pair_preferred(int B, int H, int W, int C, int R);
int resblock_pair_debug_stamps(long long *host, int n);
int resblock_pair_f16(const float *in, const float *w1_16, const float *b1, const float *w2_16, const float *b2, float *out,
                      int B, int H, int W, int C, int relu, int out_pair, hipStream_t stream, float *twin = nullptr,
                      float *hidden = nullptr);
int resblock_f32(const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                 float *out, int B, int H, int W, int C, int R, int relu, hipStream_t stream, float *twin = nullptr,
                 float *hidden = nullptr);
bool convT_small_applicable(int Cin, int Cout);
bool convT_small_pair_ok(int Cin, int Cout);
int convT_gather_f32(const float *yprime, const float *bias, float *out, int B, int H, int W, int Cout, int on, int oc,
                     int oh, int ow, int relu, hipStream_t stream);
int convT_k4s2_small_pair_f16(const float *in, const float *wn, const float *bias, float *out, int B, int H, int W,
                              int Cin, int Cout, int on, int oc, int oh, int ow, int relu, hipStream_t stream);
int pack_convT_small_weight_f32(const float *w, float *packed, int Cin, int Cout, hipStream_t stream);
int convT_k4s2_small_f32(const float *in, const float *wk, const float *bias, float *out, int B, int H,
                         int W, int Cin, int Cout, int64_t in_elems, int sn, int sc, int sh, int sw, int on,
                         int oc, int oh, int ow, int relu, hipStream_t stream);

int rel_attention_f32(const isi_attn_args *g, hipStream_t stream);
size_t rel_attention_workspace_bytes(const isi_attn_args *g);
int rel_attention_debug_stamps(long long *host, int n);
int gemm_split_debug_stamps(long long *host, int n);
int rel_attention_bwd_debug_stamps(long long *host, int n);
size_t rel_attention_bwd_workspace_floats(const isi_attn_args *g);
int rel_attention_bwd_f32(const isi_attn_bwd_args *g, hipStream_t stream);
size_t layernorm_bwd_workspace_floats(int64_t M, int D);
int layernorm_bwd_f32(const float *x, const float *res, const float *gamma, const float *dy, float *dz,
                      float *dgamma, float *dbeta, float *workspace, int64_t M, int D, float eps,
                      hipStream_t stream, float *dx = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0);
int label_smoothing_loss_f32(const float *logits, const int64_t *target, float *row_loss, float *dlogits, int64_t M,
                             int K, int num_classes, float smoothing, float grad_scale, hipStream_t stream);
int layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta, float *out,
                  int64_t M, int D, float eps, hipStream_t stream, float drop_p = 0.f, uint64_t drop_seed = 0);
int linear_rows_f32(const float *x, int x_stride, const float *W, const float *bias, const float *res,
                    int res_stride, float *out, int out_stride, int M, int N, int K, int relu,
                    hipStream_t stream);

int rel_attention_decode_f32(const isi_attn_args *g, int q_pos, float *workspace, hipStream_t stream);
int rel_attention_decode_pos_f32(const isi_attn_args *g, int q_pos, const int *pos, int self_keys, float *workspace,
                                 hipStream_t stream);
size_t rel_attention_decode_workspace_floats(int B, int H, int head_dim);
// optional tail of the sampling kernel (the decoding loop): codes[row, p - i_off] = token; x_seq[p + 1][row, 0:eff] = table[token]
struct SampleCommit {
  const float *table; int eff;
  int64_t *codes; int codes_stride;
  int p_value, i_off, S_t;
  float *x_seq; int x_stride;
  int *advance;      // ONE row only (a single workgroup: no other reader of the counter in the launch): *advance = position + 1
                     // behind the commit -- the decode loop's set_pos launch (4.3 us of a 280 us token) folded in
};
int sample_row_commit_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                          const float *u, int64_t *out, float *filtered, const int *pos, int pos_off,
                          const SampleCommit &cm, hipStream_t stream);
int sample_row_pos_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                       const float *u, int64_t *out, float *filtered, const int *pos, int pos_off,
                       hipStream_t stream);
int sample_row_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k, float top_p,
                   const float *u, int64_t *out, float *filtered, hipStream_t stream);

int rel_attention_decode_splits(int Sk, int pairs);
int attention_tail_rows(int S, int mask_mode, bool dense_mask);   // rel_attention_f32.hip
int rel_attention_decode_launch(const isi_attn_args *g, int q_pos, const int *pos, int self_keys, float *workspace,
                                int combine, hipStream_t stream);
size_t decode_stage_workspace_floats(int M, int N, int K);
int decode_stage_f32(const float *x, int x_stride, const float *ln_g, const float *ln_b, const float *W, const float *bias,
                     const float *res, int res_stride, const float *res_g, const float *res_b, float *out, int out_stride,
                     int M, int N, int K, int relu, float eps, float *workspace, size_t workspace_floats, hipStream_t st);
size_t prior_decode_scratch_floats(const isi_prior_w *w, int B);
int prior_sample_run(const isi_prior_w *w, const isi_prior_state *s, int p_begin, int p_end, float temperature,
                     int top_k, float top_p, hipStream_t stream);

bool gemm_split_applicable(int M, int N, int K, int split_mode);
// Optional extras of gemm_split_f32: a batch of nz independent products (grid y; element strides between them) and a
// band of the K range per 128-row tile -- rows [u win_rpu, (u + 1) win_rpu) only have non-zero A columns in
// [lo_slope u + lo_base, hi_slope u + hi_base] (slopes >= 0), the K chunks outside the tile's band are skipped
// (win_rpu = 0: no band).  The attention backward's dQ += G E^T uses both (rel_attention_bwd_f32.hip); the gate is the
// linear layers' (conv2d_f32 with a gate, 1x1).
struct GemmExtra {
  int nz;
  int64_t zs_a, zs_w, zs_res, zs_out;
  int win_rpu, lo_slope, lo_base, hi_slope, hi_base;
  const float *gate;      // optional (nz = 1): out = gate[m ldg + n] > 0 ? value : 0 (a ReLU's backward mask in the epilogue)
  int64_t ldg;
  float gate_scale;       // the gated value is multiplied by this (0 = 1: the inverse keep probability of a dropout folded in)
  float drop_p;           // inverted dropout on the (rectified) output, keep mask = attention_dropout_keep(seed, m ldo + n); 0: none
  uint64_t drop_seed;
};
// Optional device-resident term of every fused dropout's seed (isi_set_dropout_seed_base): a step replayed from a HIP
// graph carries its seeds as launch constants, the owner of the graph advances this counter between replays.
const uint64_t *dropout_seed_base();
// the keep decision of the fused dropout: a counter-based hash of (seed, flat output index) -- the same function in the
// forward epilogue and wherever a backward needs the mask again
// (round 6: two 32-bit multiplies instead of four -- v_mul_lo_u32 runs at a quarter of the vector rate, and the 128 keep
// decisions per lane of a 256 x 256 GEMM tile's epilogue cost the feed-forward layers 22 us of 86, tools/prof_steady.py:
// the "lowbias32" mixer (xorshift-multiply twice, full avalanche) over the index, the seed's words folded in behind
// each round)
__host__ __device__ inline bool dropout_keep(uint64_t seed, uint32_t idx, uint32_t thresh) {
  uint32_t h = idx;
  h ^= h >> 16; h *= 0x7FEB352Du;
  h ^= (uint32_t)seed;                  // (behind the first round: seeds that differ in a low bit must not just swap neighbours)
  h ^= h >> 15; h *= 0x846CA68Bu;
  h ^= (uint32_t)(seed >> 32);
  h ^= h >> 16;
  return h >= thresh;
}
int gemm_split_f32(const float *a, int64_t lda, const float *w, const float *bias, const float *res, int64_t ldr, float *out,
                   int64_t ldo, int M, int N, int K, int relu, int split_mode, hipStream_t stream,
                   const float *w16 = nullptr, const GemmExtra *extra = nullptr);
size_t conv_wgrad_workspace_floats(int Cout, int K, int M, int nphase);
int conv_wgrad_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                   float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                   int stride, int pad, int transposed, hipStream_t stream, int torch_keep = 0);
// optional hint of conv_wgrad_batched_f32: output channel c of dY is non-zero only on the pixels of units
// [lo_slope c + lo_base, hi_slope c + hi_base] (slopes >= 0), a unit = win_rpu consecutive pixels of the GEMM grid
// Band-limited products over the attention backward's G (GemmExtra.win_rpu, WgradBand): G is zeroed only in margins around
// each row's band (rel_attention_bwd_f32.hip: kMargin), which must cover what a tile of either product reads beyond a row's
// own band: at most one tile of rows / columns plus the K-chunk rounding.  The kernels assert their tiles against these.
constexpr int kBandTileMax = 128;     // rows of a banded GEMM tile, columns of a banded weight-gradient tile: at most
constexpr int kBandChunk = 32;        // K chunk of both kernels
struct WgradBand { int win_rpu, lo_slope, lo_base, hi_slope, hi_base; int required; };   // required: fail rather than read outside the band (dY is only defined there)
int conv_wgrad_batched_f32(const isi_src *s0, const isi_src *s1, const float *dy, float *dw_packed, float *db,
                           float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                           int stride, int pad, int transposed, int nz, int64_t zs_x0, int64_t zs_dy, int64_t zs_dw,
                           hipStream_t stream, int torch_keep = 0, const WgradBand *band = nullptr);
size_t conv_wgrad_batched_workspace_floats(int Cout, int K, int M, int nphase, int nz);
size_t vq_embed_sum_workspace_floats(int D, int K, int64_t N);
int relu_bwd_f32(float *dy, const float *y, int64_t n, hipStream_t st);
int axpy_f32(float *a, const float *b, float alpha, int64_t n, hipStream_t st);
int vq_bwd_f32(float *dz, const float *dq, const float *z, const float *q_st, const float *g_diff, int64_t n,
               hipStream_t st);
int colsum_num_partials(int64_t M);
int colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M, int C, hipStream_t st);
int vq_embed_sum_f32(const float *z, const int64_t *idx, float *embed_sum_dk, float *workspace,
                     size_t workspace_floats, int64_t N, int D, int K, hipStream_t stream);
int vq_ema_update_f32(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                      const float *embed_sum_dk, int D, int K, float decay, float eps, hipStream_t st);

int vq_nearest_f32(const float *z, const float *codes, const float *e2, int64_t *idx, float *q,
                   int32_t *counts, float *sse_part, int64_t N, int D, int K, int flags, hipStream_t stream);
int vq_num_partials(int64_t N);
bool vq_conv1x1_fusable(int C0, int C1, int D, int K);
int vq_debug_stamps(long long *host, int n);
int vq_conv1x1_nearest_f32(const isi_src *s0, const isi_src *s1, const float *w16, const float *bias, const float *codes,
                           const float *e2, int64_t *idx, float *q, float *q_pair, int32_t *counts, float *sse_part,
                           float *workspace, int B, int H, int W, int D, int K, hipStream_t stream, bool zero_counts = false,
                           float *z_out = nullptr, const float *wfrag_packed = nullptr);
int vq_pack_fragments_f32(const float *packed_w16, float *frag_out, int Kpad, hipStream_t stream);
size_t vq_conv1x1_workspace_floats(int C0, int C1, int D);
int vq_zero_counts(int32_t *counts, int K, hipStream_t stream);
int vq_unquantized_scalars(float *scalars2, hipStream_t stream);
int mse_loss_num_partials(int64_t n);
int mse_loss_f32(const float *a, const float *b, int64_t n, float *workspace, float *out, hipStream_t st);
int mse_loss_bwd_f32(const float *a, const float *b, const float *g, int64_t n, float *da, float *db, hipStream_t st);
int vq_finalize_f32(const float *sse_part, int n_part, const int32_t *counts, int K, int64_t N,
                    int D, float *out2, hipStream_t stream);
// both levels in one launch + out5[4] = diff_t + diff_b; the sum alone (out5[0] + out5[2]) for the two-launch paths
int vq_finalize2_f32(const float *sse_t, int np_t, const int32_t *counts_t, int K_t, int64_t N_t, const float *sse_b, int np_b,
                     const int32_t *counts_b, int K_b, int64_t N_b, int D, float *out5, hipStream_t stream);
int vq_scalars_sum_f32(float *out5, hipStream_t stream);
int embed_code_f32(const int64_t *idx, const float *codes, float *out, int64_t N, int D, int K,
                   hipStream_t stream);

int relu_inplace_f32(float *x, int64_t n, hipStream_t stream);

int spec_polar_f32(const float *stft, float *a, float *ph, int B, int T, int F, int mel, hipStream_t st);
int spec_finish_f32(const float *a, const float *ph, float *out, int B, int T, int F, int mel, hipStream_t st);
int spec_inverse_prepare_f32(const float *spec, float *a, float *ph, int B, int T, int F, hipStream_t st);
int spec_to_stft_f32(const float *a, const float *ph, float *stft, int64_t rows, int F, int mel, hipStream_t st);
int spec_affine_mask_f32(const float *x, const float *ref, float *y, int64_t B, int64_t HW, float a0, float b0, float a1,
                         float b1, float thr, int use_mask, hipStream_t st);
int spec_distance_fwd_f32(const float *xp, const float *xt, float *partial, int B, int T, int F, int RS, float eps,
                          int rows_per_block, hipStream_t st);
int spec_distance_bwd_f32(const float *xp, const float *xt, float *dx, const float *clin, const float *clog, int B, int T,
                          int F, int RS, float eps, int kind, hipStream_t st);
int spec_to_stft_bwd_f32(const float *a, const float *ph, const float *dx, float *da, float *dph, int64_t rows, int F,
                         int mel, hipStream_t st);
int spec_inverse_prepare_bwd_f32(const float *spec, const float *da, const float *dph, float *dspec, int B, int T, int F,
                                 hipStream_t st);
int overlap_add_f32(const float *frames, float *audio, int B, int T, int n_fft, int hop, int left, int64_t L,
                    hipStream_t st);

size_t packed_conv_weight_floats(int Cout, int Cin, int KH, int KW);
size_t packed_convT_k4s2_weight_floats(int Cin, int Cout);
int pack_conv_weight_w16_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW, hipStream_t stream);
int pack_conv_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW,
                         hipStream_t stream);
int pack_linear_wT_bf16(const float *w, float *out, int N, int K, hipStream_t stream);
int pack_linear_wT_bf16_multi(const void *table_dev, int n, int blocks_per_weight, hipStream_t stream);
int pack_conv_dgrad_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW, hipStream_t stream);
int pack_convT_k4s2_weight_f32(const float *w, float *packed, int Cin, int Cout,
                               hipStream_t stream);
int pack_codebook_f32(const float *embed, float *codes_kd, float *e2, int D, int K,
                      hipStream_t stream);

int vqvae_pair_activations(const isi_vqvae_w *w);
size_t vqvae_workspace_bytes(const isi_vqvae_w *w, int B, int H, int W);
int vqvae_run(const isi_vqvae_w *w, int mode, const float *x, int B, int H, int W,
              const isi_vqvae_out *out, void *workspace, size_t workspace_bytes,
              hipStream_t stream);

}  // namespace isi
