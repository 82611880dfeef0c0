// extern "C" surface of libisi_hip.so (declared in include/isi_hip.h).
#include "isi_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "isi_internal.h"
#include "knobs.h"
#include "rel_attention.h"
#include "prof.h"

namespace isi {
static std::atomic<const uint64_t *> g_dropout_seed_base{nullptr};
const uint64_t *dropout_seed_base() { return g_dropout_seed_base.load(); }
static thread_local char g_last_error[512] = "";
void set_last_error(const char *msg) {
  strncpy(g_last_error, msg ? msg : "", sizeof g_last_error - 1);
  g_last_error[sizeof g_last_error - 1] = 0;
}

namespace {
struct KnobEntry { const char *name; int Knobs::*field; int dflt; bool measure_only; };
const KnobEntry kKnobTable[] = {
    {"ISI_CONV_FLUSH", &Knobs::conv_flush, 3, false},
    {"ISI_NO_PAIRS", &Knobs::no_pairs, 0, false},
    {"ISI_NO_CONV_FIRST", &Knobs::no_conv_first, 0, false},
    {"ISI_NO_VQ_FUSION", &Knobs::no_vq_fusion, 0, false},
    {"ISI_NO_CONV_PAIR_KERNEL", &Knobs::no_conv_pair_kernel, 0, false},
    {"ISI_NO_RESBLOCK_PAIR_KERNEL", &Knobs::no_resblock_pair_kernel, 0, false},
    {"ISI_NO_CONVT_PAIR_KERNEL", &Knobs::no_convt_pair_kernel, 0, false},
    {"ISI_NO_TAIL_FUSION", &Knobs::no_tail_fusion, 0, false},
    {"ISI_NO_WGRAD_HALO", &Knobs::no_wgrad_halo, 0, false},
    {"ISI_NO_GEMM_KERNEL", &Knobs::no_gemm_kernel, 0, false},
    {"ISI_GEMM_NO_WIDE", &Knobs::gemm_no_wide, 0, false},
    {"ISI_GEMM_NARROW_BELOW", &Knobs::gemm_narrow_below, 0, false},
    {"ISI_CONV_PAIR_BM", &Knobs::conv_pair_bm, 0, false},
    {"ISI_CONV_PAIR_ALL", &Knobs::conv_pair_all, 0, false},
    {"ISI_CONV_TAP_MAJOR", &Knobs::conv_tap_major, 0, false},
    {"ISI_RESPAIR_TH", &Knobs::respair_th, 0, false},
    {"ISI_RESPAIR_ONE_WAVE_PER_ROW", &Knobs::respair_one_wave_per_row, 0, false},
    {"ISI_RES_TH", &Knobs::res_th, 0, false},
    {"ISI_CONVT_TH", &Knobs::convt_th, 0, false},
    {"ISI_CONVT_PAIR_TH", &Knobs::convt_pair_th, 0, false},
    {"ISI_DECODE_NT", &Knobs::decode_nt, 1, false},
    {"ISI_PRIOR_GRAPH", &Knobs::prior_graph, 8, false},
    {"ISI_DECODE_MFMA_ROWS", &Knobs::decode_mfma_rows, 16, false},
    {"ISI_DECODE_NO_STAT_HANDOFF", &Knobs::decode_no_stat_handoff, 0, false},
    {"ISI_DECODE_ATTN_SEPARATE_SPLITS", &Knobs::decode_attn_separate_splits, 0, false},
    {"ISI_DECODE_STATS_GLOBAL", &Knobs::decode_stats_global, 0, false},
    {"ISI_CU_COUNT", &Knobs::cu_count, 0, false},
    {"ISI_ATTN_FULL_ZERO", &Knobs::attn_full_zero, 0, false},
    {"ISI_WGRAD_SPLIT_TARGET", &Knobs::wgrad_split_target, 0, false},
    {"ISI_ATTN_G_FROM_KV", &Knobs::attn_g_from_kv, 1, false},
    {"ISI_ATTN_OLD_FWD", &Knobs::attn_old_fwd, 0, false},
    {"ISI_ATTN_NO_FWD3", &Knobs::attn_no_fwd3, 0, false},
    {"ISI_ATTN_FWD3_ALL", &Knobs::attn_fwd3_all, 0, false},
    {"ISI_CONV_ABLATE", &Knobs::conv_ablate, 0, true},
    {"ISI_VQ_DBG", &Knobs::vq_dbg, 0, true},
    {"ISI_RESPAIR_ABL", &Knobs::respair_abl, 0, true},
};
// value of an environment switch: a number, or 1 for a variable that is merely set ("ISI_NO_PAIRS=" / "=yes")
int env_value(const char *e) {
  char *end = nullptr;
  const long v = strtol(e, &end, 10);
  return end != e ? (int)v : 1;
}
}  // namespace

Knobs &knobs() {
  static Knobs k = [] {
    Knobs init;
    memset(&init, 0, sizeof init);
    for (const KnobEntry &e : kKnobTable) {
      init.*(e.field) = e.dflt;
#ifndef ISI_MEASURE
      if (e.measure_only) continue;       // ablations that produce wrong results exist in measurement builds only
#endif
      if (const char *v = getenv(e.name)) init.*(e.field) = env_value(v);
    }
    if (init.conv_flush < 0) init.conv_flush = 0;
    return init;
  }();
  return k;
}
}  // namespace isi

using namespace isi;
static inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }

extern "C" {

const char *isi_version(void) { return "isi_hip gfx950 1"; }
const char *isi_last_error(void) { return g_last_error; }

size_t isi_abi_struct_bytes(int which) {
  switch (which) {
    case 0: return sizeof(isi_src);
    case 1: return sizeof(isi_dst);
    case 2: return sizeof(isi_conv_w);
    case 3: return sizeof(isi_encoder_w);
    case 4: return sizeof(isi_decoder_w);
    case 5: return sizeof(isi_codebook_w);
    case 6: return sizeof(isi_vqvae_w);
    case 7: return sizeof(isi_vqvae_out);
    case 8: return sizeof(isi_attn_args);
    case 11: return sizeof(isi_attn_bwd_args);
    case 9: return sizeof(isi_prior_w);
    case 10: return sizeof(isi_prior_state);
    case 12: return sizeof(isi_reduce_job);
    default: return 0;
  }
}
int isi_decoder_tail_f32(const float *in_pair, const float *packed_w1, const float *bias1, const float *packed_w2,
                         const float *bias2, float *yprime_ws, const isi_dst *dst, int B, int H, int W, int Cin, int Cmid,
                         int Cout, void *stream) {
  return decoder_tail_f32(in_pair, packed_w1, bias1, packed_w2, bias2, yprime_ws, dst, B, H, W, Cin, Cmid, Cout, S(stream));
}
int isi_knob_set(const char *name, int value) {
  if (!name) return invalid("isi_knob_set: null name");
  for (const KnobEntry &e : kKnobTable)
    if (!strcmp(name, e.name)) {
#ifndef ISI_MEASURE
      if (e.measure_only && value) return unsupported("isi_knob_set: ablation switches need a -DISI_MEASURE build");
#endif
      knobs().*(e.field) = value;
      return 0;
    }
  return invalid("isi_knob_set: unknown switch");
}
int isi_knob_get(const char *name, int *value) {
  if (!name || !value) return invalid("isi_knob_get: null argument");
  for (const KnobEntry &e : kKnobTable)
    if (!strcmp(name, e.name)) { *value = knobs().*(e.field); return 0; }
  return invalid("isi_knob_get: unknown switch");
}
int isi_relu_inplace_f32(float *x, int64_t n, void *stream) { return relu_inplace_f32(x, n, S(stream)); }

int isi_prof_enable(int on) { return prof::enable(on); }
int isi_prof_num_kernels(void) { return prof::K_COUNT; }
const char *isi_prof_kernel_name(int kernel_id) { return prof::kernel_name(kernel_id); }
int isi_prof_read(int kernel_id, long long *launches, double *ms, double *flops, double *bytes) {
  return prof::read(kernel_id, launches, ms, flops, bytes);
}

int isi_pack_conv_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW,
                             void *stream) {
  return pack_conv_weight_f32(w, packed, Cout, Cin, KH, KW, S(stream));
}
int isi_pack_conv_weight_w16_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW, void *stream) {
  return pack_conv_weight_w16_f32(w, packed, Cout, Cin, KH, KW, S(stream));
}
int isi_pack_linear_wT_bf16_multi(const void *table, int n, int blocks_per_weight, void *stream) {
  return pack_linear_wT_bf16_multi(table, n, blocks_per_weight, S(stream));
}
int isi_pack_linear_wT_bf16(const float *w, float *out, int N, int K, void *stream) {
  return pack_linear_wT_bf16(w, out, N, K, S(stream));
}
int isi_pack_conv_dgrad_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH, int KW, void *stream) {
  return pack_conv_dgrad_weight_f32(w, packed, Cout, Cin, KH, KW, S(stream));
}
/* measurements only (not in isi_hip.h): phase timestamps of conv_pair_f16.hip's instrumented variant */
int isi_debug_conv_pair_stamps(long long *host, int n) { return conv_pair_debug_stamps(host, n); }
int isi_debug_convT_pair_stamps(long long *host, int n) { return convT_pair_debug_stamps(host, n); }
int isi_debug_resblock_pair_stamps(long long *host, int n) { return resblock_pair_debug_stamps(host, n); }
int isi_debug_vq_stamps(long long *host, int n) { return vq_debug_stamps(host, n); }
int isi_debug_attention_stamps(long long *host, int n) { return rel_attention_debug_stamps(host, n); }
int isi_debug_attention_fwd2_stamps(long long *host, int n) { return rel_attention_fwd2_debug_stamps(host, n); }
int isi_debug_attention_fwd3_stamps(long long *host, int n) { return rel_attention_fwd3_debug_stamps(host, n); }
int isi_set_dropout_seed_base(const void *device_u64) {
  g_dropout_seed_base.store(static_cast<const uint64_t *>(device_u64));
  return 0;
}
int isi_debug_gemm_stamps(long long *host, int n) { return gemm_split_debug_stamps(host, n); }
int isi_debug_attention_bwd_stamps(long long *host, int n) { return rel_attention_bwd_debug_stamps(host, n); }
int isi_pair_encode_f32(const float *x, float *pairs, int64_t n, void *stream) { return pair_encode_f32(x, pairs, n, S(stream)); }
int isi_pair_decode_f32(const float *pairs, float *x, int64_t n, void *stream) { return pair_decode_f32(pairs, x, n, S(stream)); }
int isi_split_conv_weight_f16(const float *packed_w, float *out, int64_t n_floats, void *stream) {
  return split_conv_weight_f16(packed_w, out, n_floats, S(stream));
}
size_t isi_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW) {
  return packed_conv_weight_floats(Cout, Cin, KH, KW);
}
int isi_pack_convT_k4s2_weight_f32(const float *w, float *packed, int Cin, int Cout, void *stream) {
  return pack_convT_k4s2_weight_f32(w, packed, Cin, Cout, S(stream));
}
size_t isi_packed_convT_k4s2_weight_floats(int Cin, int Cout) {
  return packed_convT_k4s2_weight_floats(Cin, Cout);
}
int isi_pack_codebook_f32(const float *embed, float *codes_kd, float *e2, int D, int K,
                          void *stream) {
  return pack_codebook_f32(embed, codes_kd, e2, D, K, S(stream));
}

int isi_conv2d_f32(const isi_src *src0, const isi_src *src1, const float *packed_w,
                   const float *bias, const isi_src *residual, const isi_dst *dst, int B, int H,
                   int W, int Cout, int KH, int KW, int stride, int pad, int relu, void *stream) {
  return conv2d_f32(src0, src1, packed_w, bias, residual, dst, B, H, W, Cout, KH, KW, stride, pad,
                    relu, S(stream));
}
int isi_conv_transpose2d_k4s2_f32(const isi_src *src, const float *packed_w, const float *bias,
                                  const isi_dst *dst, int B, int H, int W, int Cout, int relu,
                                  void *stream) {
  return conv_transpose2d_k4s2_f32(src, packed_w, bias, dst, B, H, W, Cout, relu, S(stream));
}
int isi_conv2d_gated_f32(const isi_src *src0, const isi_src *src1, const float *packed_w, const float *bias,
                         const isi_src *residual, const float *gate, const isi_dst *dst, int B, int H, int W,
                         int Cout, int KH, int KW, int stride, int pad, int flags, void *stream) {
  if (!gate) return ISI_E_INVALID;
  return conv2d_f32(src0, src1, packed_w, bias, residual, dst, B, H, W, Cout, KH, KW, stride, pad, flags, S(stream),
                    gate);
}
int isi_linear_f32(const isi_linear_args *a, void *stream) {
  if (!a || !a->x || !a->packed_w || !a->out) return invalid("linear: null pointer");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return invalid("linear: bad shape");
  const int mode = (a->flags & ISI_CONV_BF16X6) ? 2 : (a->flags & ISI_CONV_F16X3) ? 3 : (a->flags & ISI_CONV_BF16X3) ? 1 : 0;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->packed_w);
  if (!gemm_split_applicable(a->M, a->N, a->K, mode) || (ptrs & 15) || (a->ldx & 3) || knobs().no_gemm_kernel)
    return unsupported("linear: shape or product mode outside the GEMM kernel");
  GemmExtra gx;
  memset(&gx, 0, sizeof gx);
  gx.nz = 1; gx.gate = a->gate; gx.ldg = a->ldg; gx.gate_scale = a->gate_scale;
  gx.drop_p = a->drop_p; gx.drop_seed = a->drop_seed;
  const size_t kpad = ((size_t)a->K + 31) / 32 * 32;
  return gemm_split_f32(a->x, a->ldx, a->packed_w, a->bias, a->residual, a->residual ? a->ldr : 0, a->out, a->ldo, a->M, a->N,
                        a->K, a->flags & 1, mode, S(stream),
                        ((mode == 3 && (a->flags & ISI_CONV_W16)) || (mode == 1 && (a->flags & ISI_CONV_W16_BF16))) ? a->packed_w + (size_t)a->N * kpad : nullptr,
                        &gx);
}
int isi_conv_transpose2d_k4s2_gated_f32(const isi_src *src, const float *packed_w, const float *bias,
                                        const float *gate, const isi_dst *dst, int B, int H, int W, int Cout,
                                        int flags, void *stream) {
  if (!gate) return ISI_E_INVALID;
  return conv_transpose2d_k4s2_f32(src, packed_w, bias, dst, B, H, W, Cout, flags, S(stream), gate);
}

int isi_resblock_f32(const float *in, const float *packed_w3, const float *b3, const float *packed_w1,
                     const float *b1, float *out, int B, int H, int W, int C, int R, int relu,
                     void *stream) {
  return resblock_f32(in, packed_w3, b3, packed_w1, b1, out, B, H, W, C, R, relu, S(stream));
}
int isi_resblock_fusable(int C, int R) { return resblock_fusable(C, R) ? 1 : 0; }
int isi_pack_multi(const void *table, int n, int blocks_per_entry, void *stream) { return pack_multi(table, n, blocks_per_entry, S(stream)); }
int isi_conv2d_twin_f32(const isi_src *src0, const isi_src *src1, const float *packed_w, const float *bias,
                        const isi_dst *dst, float *twin, int B, int H, int W, int Cout, int KH, int KW, int stride,
                        int pad, int flags, void *stream) {
  if (!twin) return ISI_E_INVALID;
  return conv2d_f32(src0, src1, packed_w, bias, nullptr, dst, B, H, W, Cout, KH, KW, stride, pad, flags, S(stream), nullptr, twin);
}
int isi_conv_transpose2d_k4s2_twin_f32(const isi_src *src, const float *packed_w, const float *bias, const isi_dst *dst,
                                       float *twin, int B, int H, int W, int Cout, int flags, void *stream) {
  if (!twin) return ISI_E_INVALID;
  return conv_transpose2d_k4s2_f32(src, packed_w, bias, dst, B, H, W, Cout, flags, S(stream), nullptr, twin);
}
int isi_resblock_tape_f32(const float *in, const float *packed_w3, const float *b3, const float *packed_w1, const float *b1,
                          float *out, float *twin, float *hidden, int B, int H, int W, int C, int R, int flags, void *stream) {
  if (!twin && !hidden) return ISI_E_INVALID;
  return resblock_f32(in, packed_w3, b3, packed_w1, b1, out, B, H, W, C, R, flags, S(stream), twin, hidden);
}
int isi_resblock_pair_route(int B, int H, int W, int C, int R) { return resblock_pair_preferred(B, H, W, C, R) ? 1 : 0; }
int isi_conv2d_pair_route(int C0, int C1, int Cout, int KH, int KW) {
  // would isi_conv2d_f32 with pair-format sources run the LDS-DMA kernel (the one with the twin epilogue)?
  return (conv_pair_kernel_ok(C0, C1, Cout, KH * KW) && KH <= 4 && KW <= 4 && KH * KW * (C0 + C1) >= 256) ? 1 : 0;
}
int isi_conv_transpose2d_pair_route(int Cin, int Cout) { return convT_pair_ok(Cin, Cout) ? 1 : 0; }
int isi_conv_wgrad_halo_route(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, int OH, int OW) {
  return conv_wgrad_halo_route(Cout, C0, C1, KH, KW, stride, pad, OH, OW) ? 1 : 0;
}

int isi_spec_polar_f32(const float *stft, float *a, float *ph, int B, int T, int F, int mel, void *stream) {
  return spec_polar_f32(stft, a, ph, B, T, F, mel, S(stream));
}
int isi_spec_finish_f32(const float *a, const float *ph, float *spec, int B, int T, int F, int mel, void *stream) {
  return spec_finish_f32(a, ph, spec, B, T, F, mel, S(stream));
}
int isi_spec_inverse_prepare_f32(const float *spec, float *a, float *ph, int B, int T, int F, void *stream) {
  return spec_inverse_prepare_f32(spec, a, ph, B, T, F, S(stream));
}
int isi_spec_to_stft_f32(const float *a, const float *ph, float *stft, int64_t rows, int F, int mel, void *stream) {
  return spec_to_stft_f32(a, ph, stft, rows, F, mel, S(stream));
}
int isi_spec_affine_mask_f32(const float *x, const float *ref, float *y, int64_t B, int64_t HW, float a0, float b0,
                             float a1, float b1, float thr, int use_mask, void *stream) {
  return spec_affine_mask_f32(x, ref, y, B, HW, a0, b0, a1, b1, thr, use_mask, S(stream));
}
int isi_spec_distance_fwd_f32(const float *xp, const float *xt, float *partial, int B, int T, int F, int RS, float eps,
                              int rows_per_block, void *stream) {
  return spec_distance_fwd_f32(xp, xt, partial, B, T, F, RS, eps, rows_per_block, S(stream));
}
int isi_spec_distance_bwd_f32(const float *xp, const float *xt, float *dx, const float *clin, const float *clog, int B,
                              int T, int F, int RS, float eps, int kind, void *stream) {
  return spec_distance_bwd_f32(xp, xt, dx, clin, clog, B, T, F, RS, eps, kind, S(stream));
}
int isi_spec_to_stft_bwd_f32(const float *a, const float *ph, const float *dx, float *da, float *dph, int64_t rows, int F,
                             int mel, void *stream) {
  return spec_to_stft_bwd_f32(a, ph, dx, da, dph, rows, F, mel, S(stream));
}
int isi_spec_inverse_prepare_bwd_f32(const float *spec, const float *da, const float *dph, float *dspec, int B, int T,
                                     int F, void *stream) {
  return spec_inverse_prepare_bwd_f32(spec, da, dph, dspec, B, T, F, S(stream));
}
int isi_overlap_add_f32(const float *frames, float *audio, int B, int T, int n_fft, int hop, int left, int64_t L,
                        void *stream) {
  return overlap_add_f32(frames, audio, B, T, n_fft, hop, left, L, S(stream));
}
int isi_rel_attention_f32(const isi_attn_args *args, void *stream) { return rel_attention_f32(args, S(stream)); }
size_t isi_rel_attention_workspace_bytes(const isi_attn_args *args) { return rel_attention_workspace_bytes(args); }
size_t isi_rel_attention_bwd_workspace_floats(const isi_attn_args *fwd) { return rel_attention_bwd_workspace_floats(fwd); }
int isi_rel_attention_bwd_f32(const isi_attn_bwd_args *args, void *stream) { return rel_attention_bwd_f32(args, S(stream)); }
size_t isi_layernorm_bwd_workspace_floats(int64_t M, int D) { return layernorm_bwd_workspace_floats(M, D); }
int isi_layernorm_bwd_f32(const float *x, const float *residual, const float *gamma, const float *dy, float *dz,
                          float *dgamma, float *dbeta, float *workspace, int64_t M, int D, float eps, void *stream) {
  return layernorm_bwd_f32(x, residual, gamma, dy, dz, dgamma, dbeta, workspace, M, D, eps, S(stream));
}
int isi_label_smoothing_loss_f32(const float *logits, const int64_t *target, float *row_loss, float *dlogits, int64_t M,
                                 int K, int num_classes, float smoothing, float grad_scale, void *stream) {
  return label_smoothing_loss_f32(logits, target, row_loss, dlogits, M, K, num_classes, smoothing, grad_scale, S(stream));
}
int isi_layernorm_dropout_f32(const float *x, const float *residual, const float *gamma, const float *beta, float *out,
                              int64_t M, int D, float eps, float drop_p, uint64_t drop_seed, void *stream) {
  return layernorm_f32(x, residual, gamma, beta, out, M, D, eps, S(stream), drop_p, drop_seed);
}
int isi_layernorm_dropout_bwd_f32(const float *x, const float *residual, const float *gamma, const float *dy, float *dz,
                                  float *dx, float *dgamma, float *dbeta, float *workspace, int64_t M, int D, float eps,
                                  float drop_p, uint64_t drop_seed, void *stream) {
  return layernorm_bwd_f32(x, residual, gamma, dy, dz, dgamma, dbeta, workspace, M, D, eps, S(stream), dx, drop_p, drop_seed);
}
int isi_layernorm_f32(const float *x, const float *residual, const float *gamma, const float *beta, float *out,
                      int64_t M, int D, float eps, void *stream) {
  return layernorm_f32(x, residual, gamma, beta, out, M, D, eps, S(stream));
}
int isi_linear_rows_f32(const float *x, int x_stride, const float *W, const float *bias, const float *residual,
                        int res_stride, float *out, int out_stride, int M, int N, int K, int relu,
                        void *stream) {
  return linear_rows_f32(x, x_stride, W, bias, residual, res_stride, out, out_stride, M, N, K, relu, S(stream));
}

size_t isi_decode_stage_workspace_floats(int M, int N, int K) { return decode_stage_workspace_floats(M, N, K); }
int isi_decode_stage_f32(const float *x, int x_stride, const float *ln_g, const float *ln_b, const float *W, const float *bias,
                         const float *res, int res_stride, const float *res_g, const float *res_b, float *out, int out_stride,
                         int M, int N, int K, int relu, float eps, float *workspace, size_t workspace_floats, void *stream) {
  return decode_stage_f32(x, x_stride, ln_g, ln_b, W, bias, res, res_stride, res_g, res_b, out, out_stride, M, N, K, relu, eps,
                          workspace, workspace_floats, S(stream));
}
int isi_rel_attention_decode_f32(const isi_attn_args *args, int q_pos, float *workspace, void *stream) {
  return rel_attention_decode_f32(args, q_pos, workspace, S(stream));
}
size_t isi_rel_attention_decode_workspace_floats(int B, int H, int head_dim) {
  return rel_attention_decode_workspace_floats(B, H, head_dim);
}
int isi_sample_row_f32(const float *logits, int stride, int rows, int n, float temperature, int top_k,
                       float top_p, const float *u, int64_t *out, float *filtered, void *stream) {
  return sample_row_f32(logits, stride, rows, n, temperature, top_k, top_p, u, out, filtered, S(stream));
}

size_t isi_prior_decode_scratch_floats(const isi_prior_w *w, int B) { return prior_decode_scratch_floats(w, B); }
int isi_prior_sample_run(const isi_prior_w *w, const isi_prior_state *state, int p_begin, int p_end,
                         float temperature, int top_k, float top_p, void *stream) {
  return prior_sample_run(w, state, p_begin, p_end, temperature, top_k, top_p, S(stream));
}

size_t isi_conv_wgrad_workspace_floats(int Cout, int K, int M, int nphase) {
  return conv_wgrad_workspace_floats(Cout, K, M, nphase);
}
int isi_conv_wgrad_f32(const isi_src *src0, const isi_src *src1, const float *dy, float *dw_packed, float *db,
                       float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                       int stride, int pad, int transposed, void *stream) {
  return conv_wgrad_f32(src0, src1, dy, dw_packed, db, workspace, workspace_floats, B, H, W, Cout, KH, KW, stride,
                        pad, transposed, S(stream));
}
int isi_conv_wgrad_torch_f32(const isi_src *src0, const isi_src *src1, const float *dy, float *dw_torch, int cin_keep,
                             float *db, float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH,
                             int KW, int stride, int pad, int flags, void *stream) {
  if (cin_keep < 1) return ISI_E_INVALID;
  return conv_wgrad_f32(src0, src1, dy, dw_torch, db, workspace, workspace_floats, B, H, W, Cout, KH, KW, stride, pad,
                        flags, S(stream), cin_keep);
}
int isi_conv_wgrad_deferred_f32(const isi_src *src0, const isi_src *src1, const float *dy, float *dw_torch, int cin_keep,
                                float *db, float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH,
                                int KW, int stride, int pad, int flags, void *stream, isi_reduce_job *jobs_out, int *n_jobs) {
  if (cin_keep < 1) return ISI_E_INVALID;
  return conv_wgrad_deferred_f32(src0, src1, dy, dw_torch, cin_keep, db, workspace, workspace_floats, B, H, W, Cout, KH, KW,
                                 stride, pad, flags, S(stream), jobs_out, n_jobs);
}
int isi_reduce_jobs_f32(const isi_reduce_job *jobs, int n_jobs, void *stream) { return reduce_jobs_f32(jobs, n_jobs, S(stream)); }
int isi_relu_bwd_f32(float *dy, const float *y, int64_t n, void *stream) { return relu_bwd_f32(dy, y, n, S(stream)); }
int isi_axpy_f32(float *a, const float *b, float alpha, int64_t n, void *stream) {
  return axpy_f32(a, b, alpha, n, S(stream));
}
int isi_vq_bwd_f32(float *dz, const float *dq, const float *z, const float *q_st, const float *g_diff, int64_t n,
                   void *stream) {
  return vq_bwd_f32(dz, dq, z, q_st, g_diff, n, S(stream));
}
int isi_pad_channels4_f32(const isi_src *src, float *out_nhwc4, int B, int H, int W, void *stream) {
  if (!src || !src->ptr) return ISI_E_INVALID;
  return pad_channels4_f32(src->ptr, out_nhwc4, B, src->C, H, W, src->sn, src->sc, src->sh, src->sw, S(stream));
}
int isi_add_gate_rows_f32(float *out, const float *a, int64_t lda, const float *b, const float *y, int64_t M, int C, void *stream) {
  return add_gate_rows_f32(out, a, lda, b, y, M, C, S(stream));
}
int isi_vq_bwd_rows_f32(float *dz, const float *dq, int64_t ldq, const float *z, const float *q_st, const float *g_diff,
                        int64_t M, int D, void *stream) {
  return vq_bwd_rows_f32(dz, dq, ldq, z, q_st, g_diff, M, D, S(stream));
}
int isi_mse_loss_num_partials(int64_t n) { return mse_loss_num_partials(n); }
int isi_mse_loss_f32(const float *a, const float *b, int64_t n, float *workspace, float *out, void *stream) {
  return mse_loss_f32(a, b, n, workspace, out, S(stream));
}
int isi_mse_loss_bwd_f32(const float *a, const float *b, const float *g, int64_t n, float *da, float *db, void *stream) {
  return mse_loss_bwd_f32(a, b, g, n, da, db, S(stream));
}
int isi_colsum_num_partials(int64_t M) { return colsum_num_partials(M); }
int isi_colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M, int C, void *stream) {
  return colsum_f32(x, x_stride, out, workspace, M, C, S(stream));
}
size_t isi_vq_embed_sum_workspace_floats(int D, int K, int64_t N) { return vq_embed_sum_workspace_floats(D, K, N); }
int isi_vq_embed_sum_f32(const float *z, const int64_t *idx, float *embed_sum_dk, float *workspace,
                         size_t workspace_floats, int64_t N, int D, int K, void *stream) {
  return vq_embed_sum_f32(z, idx, embed_sum_dk, workspace, workspace_floats, N, D, K, S(stream));
}
int isi_vq_ema_update_f32(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                          const float *embed_sum_dk, int D, int K, float decay, float eps, void *stream) {
  return vq_ema_update_f32(embed, cluster_size, embed_avg, counts, embed_sum_dk, D, K, decay, eps, S(stream));
}

int isi_vq_nearest_f32(const float *z, const float *codes_kd, const float *e2, int64_t *idx_out,
                       float *q_out, int32_t *counts, float *sse_part, int64_t N, int D, int K,
                       void *stream) {
  return vq_nearest_f32(z, codes_kd, e2, idx_out, q_out, counts, sse_part, N, D, K, 0, S(stream));
}
int isi_vq_conv1x1_nearest_f32(const isi_src *src0, const isi_src *src1, const float *packed_w16, const float *bias,
                               const float *codes_kd, const float *e2, int64_t *idx_out, float *q_out,
                               float *q_pair_out, int32_t *counts, float *sse_part, float *workspace, int B, int H,
                               int W, int D, int K, void *stream) {
  return vq_conv1x1_nearest_f32(src0, src1, packed_w16, bias, codes_kd, e2, idx_out, q_out, q_pair_out, counts, sse_part,
                                workspace, B, H, W, D, K, S(stream));
}
int isi_vq_conv1x1_nearest_tape_f32(const isi_src *src0, const isi_src *src1, const float *packed_w16, const float *bias,
                                    const float *codes_kd, const float *e2, int64_t *idx_out, float *q_out,
                                    float *q_pair_out, float *z_out, int32_t *counts, float *sse_part, float *workspace,
                                    int B, int H, int W, int D, int K, void *stream) {
  return vq_conv1x1_nearest_f32(src0, src1, packed_w16, bias, codes_kd, e2, idx_out, q_out, q_pair_out, counts, sse_part,
                                workspace, B, H, W, D, K, S(stream), /*zero_counts*/ true, z_out);
}
size_t isi_vq_conv1x1_workspace_floats(int C0, int C1, int D) { return vq_conv1x1_workspace_floats(C0, C1, D); }
int isi_vq_pack_fragments_f32(const float *packed_w16, float *frag_out, int Kpad, void *stream) {
  return vq_pack_fragments_f32(packed_w16, frag_out, Kpad, S(stream));
}
int isi_vq_conv1x1_fusable(int C0, int C1, int D, int K) { return vq_conv1x1_fusable(C0, C1, D, K) ? 1 : 0; }
int isi_vq_nearest_flags_f32(const float *z, const float *codes_kd, const float *e2, int64_t *idx_out,
                             float *q_out, int32_t *counts, float *sse_part, int64_t N, int D, int K,
                             int flags, void *stream) {
  return vq_nearest_f32(z, codes_kd, e2, idx_out, q_out, counts, sse_part, N, D, K, flags, S(stream));
}
int isi_vq_num_partials(int64_t N) { return vq_num_partials(N); }
int isi_vq_finalize_f32(const float *sse_part, int n_part, const int32_t *counts, int K, int64_t N,
                        int D, float *out2, void *stream) {
  return vq_finalize_f32(sse_part, n_part, counts, K, N, D, out2, S(stream));
}
int isi_embed_code_f32(const int64_t *idx, const float *codes_kd, float *out, int64_t N, int D,
                       int K, void *stream) {
  return embed_code_f32(idx, codes_kd, out, N, D, K, S(stream));
}

int isi_vqvae_pair_activations(const isi_vqvae_w *w) { return vqvae_pair_activations(w); }
size_t isi_vqvae_workspace_bytes(const isi_vqvae_w *w, int B, int H, int W) {
  return vqvae_workspace_bytes(w, B, H, W);
}
int isi_vqvae_run(const isi_vqvae_w *w, int mode, const float *x, int B, int H, int W,
                  const isi_vqvae_out *out, void *workspace, size_t workspace_bytes, void *stream) {
  return vqvae_run(w, mode, x, B, H, W, out, workspace, workspace_bytes, S(stream));
}

}  // extern "C"
