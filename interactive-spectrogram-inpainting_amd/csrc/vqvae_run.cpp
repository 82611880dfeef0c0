// Host-side launch sequence of the two-level VQ-VAE (no allocation, no sync:
// every kernel is enqueued on the caller's stream, scratch comes from the
// caller's workspace).  Mirrors VQVAE.encode / decode / forward of the
// reference (vqvae/vqvae.py:245-286) and the Rosinality encoder / decoder
// stacks (vqvae/encoder_decoder.py:38-227), with these fusions:
//   * every ReLU is folded into the epilogue of the producing convolution
//     (all ReLUs on this path are in-place in the reference, so the tensor a
//     consumer sees is always the rectified one -- including the residual
//     input of RosinalityResBlock, encoder_decoder.py:22-35);
//   * torch.cat (vqvae.py:270,282) is replaced by two-source convolutions;
//   * NCHW<->NHWC permutes (vqvae.py:260-262,272-274) vanish: internal
//     activations are channels-last, strides describe the API tensors.
#include <algorithm>

#include <cstdlib>
#include "isi_common.h"
#include "isi_internal.h"
#include "knobs.h"

namespace isi {

namespace {

struct Act {  // dense channels-last activation
  float *p;
  int C, H, W;
  size_t elems(int B) const { return (size_t)B * H * W * C; }
};

isi_src src_nhwc(const float *p, int C, int H, int W, int Wstride = -1) {
  if (Wstride < 0) Wstride = W;
  isi_src s;
  s.ptr = p; s.C = C; s.sc = 1; s.sw = C; s.sh = (int64_t)Wstride * C; s.sn = (int64_t)H * Wstride * C;
  return s;
}
isi_dst dst_nhwc(float *p, int C, int H, int W) {
  isi_dst d;
  d.ptr = p; d.sc = 1; d.sw = C; d.sh = (int64_t)W * C; d.sn = (int64_t)H * W * C;
  return d;
}
isi_src src_nchw(const float *p, int C, int H, int W) {
  isi_src s;
  s.ptr = p; s.C = C; s.sw = 1; s.sh = W; s.sc = (int64_t)H * W; s.sn = (int64_t)C * H * W;
  return s;
}
isi_dst dst_nchw(float *p, int C, int H, int W) {
  isi_dst d;
  d.ptr = p; d.sw = 1; d.sh = W; d.sc = (int64_t)H * W; d.sn = (int64_t)C * H * W;
  return d;
}

inline int down_dim(int x) { return (x + 2 - 4) / 2 + 1; }

struct Bump {
  char *base;
  size_t cap, off;
  bool dry;  // size query only
  void *take(size_t bytes) {
    off = round_up(off, 256);
    void *p = dry ? nullptr : base + off;
    off += bytes;
    return p;
  }
  float *floats(size_t n) { return static_cast<float *>(take(n * sizeof(float))); }
};

struct Shapes {
  int Hb, Wb, Cb;  // bottom latent grid / hidden channels
  int Ht, Wt;
  int Wq;          // cropped bottom width (adapt_quantized_durations)
  size_t max_act;  // largest intermediate activation (floats)
};

bool encoder_out_dims(const isi_encoder_w &e, int &H, int &W, int B, size_t &max_act) {
  for (int i = 0; i < e.n_down; ++i) {
    H = down_dim(H); W = down_dim(W);
    if (H <= 0 || W <= 0) return false;
    max_act = std::max(max_act, (size_t)B * H * W * e.down[i].Cout);
  }
  max_act = std::max(max_act, (size_t)B * H * W * e.conv3.Cout);
  return true;
}

void decoder_max_act(const isi_decoder_w &d, int H, int W, int B, size_t &max_act) {
  max_act = std::max(max_act, (size_t)B * H * W * d.conv3.Cout);
  for (int i = 0; i < d.n_up; ++i) {
    H *= 2; W *= 2;
    max_act = std::max(max_act, (size_t)B * H * W * d.up[i].Cout);
  }
}

bool compute_shapes(const isi_vqvae_w &w, int B, int H, int W, Shapes &s) {
  s.max_act = 0;
  int h = H, ww = W;
  if (!encoder_out_dims(w.enc_b, h, ww, B, s.max_act)) return false;
  s.Hb = h; s.Wb = ww; s.Cb = w.enc_b.conv3.Cout;
  if (!encoder_out_dims(w.enc_t, h, ww, B, s.max_act)) return false;
  s.Ht = h; s.Wt = ww;
  int up = 1;
  for (int i = 0; i < w.dec_t.n_up; ++i) up *= 2;
  if (s.Ht * up != s.Hb) return false;  // torch.cat would fail in the reference
  s.Wq = std::min(s.Wt * up, s.Wb);
  decoder_max_act(w.dec_t, s.Ht, s.Wt, B, s.max_act);
  decoder_max_act(w.dec, s.Hb, s.Wq, B, s.max_act);
  return true;
}

// Rosinality residual stack shared by encoder and decoder: `cur` holds the
// rectified input r; each block writes relu(r + conv1(relu(conv3(r)))).
// The last block writes to `final_out` when that is non-null.
// `in_pair` / `out_pair`: the stack's input / output tensors are in the split-f16 pair format (isi_hip.h,
// ISI_CONV_*_PAIR; only set when every block is fusable, see pairs_eligible); tensors between blocks follow `in_pair`
// (the callers pass fp32 in: see run_encoder).
// Will the residual stack that follows a convolution run in the pair format (resblock_pair_f16.hip: every block reads
// and writes pairs)?  Otherwise its blocks read fp32 and only the last one writes pairs.
bool res_stack_in_pairs(bool pairs, int n_res, const isi_conv_w *res3, int B, int H, int W, int C) {
  if (!pairs || n_res == 0) return false;
  for (int i = 0; i < n_res; ++i)
    if (!resblock_pair_preferred(B, H, W, C, res3[i].Cout)) return false;
  return true;
}

int run_res_stack(int n_res, const isi_conv_w *res3, const isi_conv_w *res1, Act &cur, int B,
                  float *s0, float *s1, float *hid, float *final_out, int pf, bool in_pair, bool out_pair,
                  hipStream_t st) {
  for (int i = 0; i < n_res; ++i) {
    const int R = res3[i].Cout;
    float *outp = (i == n_res - 1 && final_out) ? final_out : (cur.p == s0 ? s1 : s0);
    if (resblock_fusable(cur.C, R)) {
      const bool op = (i == n_res - 1) ? out_pair : in_pair;
      int rc = resblock_f32(cur.p, res3[i].w, res3[i].bias, res1[i].w, res1[i].bias, outp, B, cur.H,
                            cur.W, cur.C, R, /*relu*/ 1 | pf | (in_pair ? ISI_CONV_IN0_PAIR : 0) |
                            (op ? ISI_CONV_OUT_PAIR : 0), st);
      if (rc) return rc;
      cur.p = outp;
      continue;
    }
    isi_src in = src_nhwc(cur.p, cur.C, cur.H, cur.W);
    isi_dst dh = dst_nhwc(hid, R, cur.H, cur.W);
    int rc = conv2d_f32(&in, nullptr, res3[i].w, res3[i].bias, nullptr, &dh, B, cur.H, cur.W, R, 3,
                        3, 1, 1, /*relu*/ 1 | pf, st);
    if (rc) return rc;
    isi_src hin = src_nhwc(hid, R, cur.H, cur.W);
    isi_dst dout = dst_nhwc(outp, cur.C, cur.H, cur.W);
    rc = conv2d_f32(&hin, nullptr, res1[i].w, res1[i].bias, &in, &dout, B, cur.H, cur.W, cur.C, 1,
                    1, 1, 0, /*relu*/ 1 | pf, st);
    if (rc) return rc;
    cur.p = outp;
  }
  return ISI_OK;
}

// RosinalityEncoder (encoder_decoder.py:38-126).  `in` may be NCHW.
// `in_pair`: the input is in the pair format; `pairs`: every tensor this encoder writes (including its output) is.
int run_encoder(const isi_encoder_w &e, isi_src in, int B, int H, int W, float *s0, float *s1,
                float *hid, float *final_out, Act &out, int pf, bool in_pair, bool pairs, hipStream_t st) {
  Act cur{nullptr, in.C, H, W};
  isi_src cs = in;
  bool stack_pairs = false;
  const int op = pairs ? ISI_CONV_OUT_PAIR : 0;
  bool cp = in_pair;   // format of the current tensor
  for (int i = 0; i < e.n_down; ++i) {
    const int OH = down_dim(cur.H), OW = down_dim(cur.W);
    float *o = (cur.p == s0) ? s1 : s0;
    isi_dst d = dst_nhwc(o, e.down[i].Cout, OH, OW);
    int rc = conv2d_f32(&cs, nullptr, e.down[i].w, e.down[i].bias, nullptr, &d, B, cur.H, cur.W,
                        e.down[i].Cout, 4, 4, 2, 1, 1 | pf | (cp ? ISI_CONV_IN0_PAIR : 0) | op, st);
    if (rc) return rc;
    cur = Act{o, e.down[i].Cout, OH, OW};
    cs = src_nhwc(cur.p, cur.C, cur.H, cur.W);
    cp = pairs;
  }
  {
    float *o = (e.n_res == 0) ? final_out : ((cur.p == s0) ? s1 : s0);
    isi_dst d = dst_nhwc(o, e.conv3.Cout, cur.H, cur.W);
    stack_pairs = res_stack_in_pairs(pairs, e.n_res, e.res3, B, cur.H, cur.W, e.conv3.Cout);
    int rc = conv2d_f32(&cs, nullptr, e.conv3.w, e.conv3.bias, nullptr, &d, B, cur.H, cur.W,
                        e.conv3.Cout, 3, 3, 1, 1, 1 | pf | (cp ? ISI_CONV_IN0_PAIR : 0) | ((e.n_res && !stack_pairs) ? 0 : op), st);
    if (rc) return rc;
    cur = Act{o, e.conv3.Cout, cur.H, cur.W};
  }
  // the fused residual block stages its input once per slice (not once per tap): it reads and writes fp32 between
  // blocks -- decoding the skip connection from pairs cost more than the conversion saved, 64 -> 72 us per block --
  // and only the stack's last block writes pairs, for the implicit-GEMM consumers of the encoder's output
  int rc = run_res_stack(e.n_res, e.res3, e.res1, cur, B, s0, s1, hid, final_out, pf, stack_pairs, pairs, st);
  if (rc) return rc;
  out = cur;
  return ISI_OK;
}

// RosinalityDecoder (encoder_decoder.py:129-227).  Input = cat(in0, in1) on
// channels (in1 optional); the last transposed conv writes through `final_dst`.
// `in0_pair` / `in1_pair`: formats of the two inputs; `pairs`: internal tensors are written in the pair format
// wherever their consumer reads it (the few-channel transposed-convolution kernel reads fp32); `final_pair`: format
// of the tensor written through `final_dst`.
int run_decoder(const isi_decoder_w &d, const isi_src &in0, const isi_src *in1, int B, int H, int W,
                float *s0, float *s1, float *hid, const isi_dst &final_dst, int pf, bool in0_pair, bool in1_pair,
                bool pairs, bool final_pair, hipStream_t st) {
  Act cur{s0, d.conv3.Cout, H, W};
  bool stack_pairs = false;
  // format in which up[i] wants its input: pairs for the implicit-GEMM kernels and for the pair form of the
  // few-channel kernel (64 input channels, <= 2 outputs, fp32 result: the last layer of the default model); fp32 for
  // its exact-fp32 form
  auto up_reads_pair = [&](int i, int Cin) {
    if (!pairs) return false;
    if (!convT_small_applicable(Cin, d.up[i].Cout)) return true;
    const bool last = (i == d.n_up - 1);
    return convT_small_pair_ok(Cin, d.up[i].Cout) && last && !final_pair && (pf & ISI_CONV_F16X3) && (pf & ISI_CONV_W16);
  };
  {
    isi_dst dd = dst_nhwc(cur.p, cur.C, H, W);
    stack_pairs = res_stack_in_pairs(pairs, d.n_res, d.res3, B, H, W, cur.C);
    const bool op = d.n_res ? stack_pairs : (d.n_up ? up_reads_pair(0, cur.C) : final_pair);   // fp32 into a register-staged stack
    int rc = conv2d_f32(&in0, in1, d.conv3.w, d.conv3.bias, nullptr, &dd, B, H, W, cur.C, 3, 3, 1,
                        1, 1 | pf | (in0_pair ? ISI_CONV_IN0_PAIR : 0) | (in1_pair ? ISI_CONV_IN1_PAIR : 0) |
                        (op ? ISI_CONV_OUT_PAIR : 0), st);
    if (rc) return rc;
  }
  bool cp = pairs;   // format of `cur` after the residual stack
  {
    cp = d.n_up ? up_reads_pair(0, cur.C) : final_pair;
    int rc = run_res_stack(d.n_res, d.res3, d.res1, cur, B, s0, s1, hid, nullptr, pf, stack_pairs, cp, st);
    if (rc) return rc;
  }
  for (int i = 0; i < d.n_up; ++i) {
    const bool last = (i == d.n_up - 1);
    isi_src s = src_nhwc(cur.p, cur.C, cur.H, cur.W);
    float *o = (cur.p == s0) ? s1 : s0;
    // the last two transposed convolutions as one producer / consumer pair (convT_pair_f16.hip, YP mode): the
    // [B, 2H, 2W, 64] activation between them is never written
    if (i == d.n_up - 2 && cp && !final_pair && (pf & ISI_CONV_F16X3) && (pf & ISI_CONV_W16) &&
        decoder_tail_ok(cur.C, d.up[i].Cout, d.up[i + 1].Cout)) {
      int rc = decoder_tail_f32(cur.p, d.up[i].w, d.up[i].bias, d.up[i + 1].w, d.up[i + 1].bias, o, &final_dst, B, cur.H,
                                cur.W, cur.C, d.up[i].Cout, d.up[i + 1].Cout, st);
      if (rc) return rc;
      return ISI_OK;
    }
    isi_dst dd = last ? final_dst : dst_nhwc(o, d.up[i].Cout, 2 * cur.H, 2 * cur.W);
    const bool op = last ? final_pair : up_reads_pair(i + 1, d.up[i].Cout);
    int rc = conv_transpose2d_k4s2_f32(&s, d.up[i].w, d.up[i].bias, &dd, B, cur.H, cur.W,
                                       d.up[i].Cout, (last ? 0 : 1) | pf | (cp ? ISI_CONV_IN0_PAIR : 0) |
                                       (op ? ISI_CONV_OUT_PAIR : 0), st);
    if (rc) return rc;
    cp = op;
    cur = Act{o, d.up[i].Cout, 2 * cur.H, 2 * cur.W};
  }
  return ISI_OK;
}

// Can this model run with its internal activations in the split-f16 pair format (precision 4 with pack-time weight
// pieces)?  Every launch that would READ a pair tensor must be one that accepts it: the split-f16 convolution kernel
// (conv_pair_sources_ok) or the fused residual block.  All-or-nothing: one ineligible layer keeps the whole model on
// fp32 activations (ISI_NO_PAIRS in the environment forces that, for measurements).
bool stack_fusable(int C, int n_res, const isi_conv_w *res3) {
  for (int i = 0; i < n_res; ++i)
    if (!resblock_fusable(C, res3[i].Cout)) return false;
  return true;
}
bool encoder_pairs_ok(const isi_encoder_w &e, int Cin, bool in_pair) {
  int C = Cin;
  bool cp = in_pair;
  for (int i = 0; i < e.n_down; ++i) {
    if (cp && !conv_pair_sources_ok(C, 0, e.down[i].Cout, 16)) return false;
    C = e.down[i].Cout;
    cp = true;
  }
  if (cp && !conv_pair_sources_ok(C, 0, e.conv3.Cout, 9)) return false;
  return stack_fusable(e.conv3.Cout, e.n_res, e.res3);
}
bool decoder_pairs_ok(const isi_decoder_w &d, int C0, int C1, bool in0_pair, bool in1_pair) {
  if ((in0_pair || in1_pair) && !conv_pair_sources_ok(C0, C1, d.conv3.Cout, 9)) return false;
  if (!stack_fusable(d.conv3.Cout, d.n_res, d.res3)) return false;
  int C = d.conv3.Cout;
  for (int i = 0; i < d.n_up; ++i) {
    // the few-channel kernel is handed fp32; the implicit-GEMM phases read pairs
    if (!convT_small_applicable(C, d.up[i].Cout) && !conv_pair_sources_ok(C, 0, d.up[i].Cout, 4)) return false;
    C = d.up[i].Cout;
  }
  return d.n_up >= 1;
}
bool pairs_eligible(const isi_vqvae_w &w) {
  const bool off = knobs().no_pairs != 0;   // tests compare both paths in one process (isi_knob_set)
  if (off || w.precision != 4 || !w.w16) return false;
  // the first layer must be able to WRITE pairs: the 2-channel kernel (conv_first_f32.hip) does, the generic gather
  // kernel does not
  if (knobs().no_conv_first || w.in_channel != 2 || w.enc_b.n_down < 1 ||
      (w.enc_b.down[0].Cout != 32 && w.enc_b.down[0].Cout != 64)) return false;
  const int D = w.quantize_t.D;
  if (w.quantize_b.D != D) return false;
  if (!encoder_pairs_ok(w.enc_b, w.in_channel, false)) return false;
  if (!encoder_pairs_ok(w.enc_t, w.enc_b.conv3.Cout, true)) return false;
  if (!conv_pair_sources_ok(w.enc_t.conv3.Cout, 0, D, 1)) return false;                 // quantize_conv_t
  if (!decoder_pairs_ok(w.dec_t, D, 0, true, false)) return false;      // reads the pair copy of quant_t
  if (convT_small_applicable(w.dec_t.n_up == 1 ? w.dec_t.conv3.Cout : w.dec_t.up[w.dec_t.n_up - 2].Cout,
                             w.dec_t.up[w.dec_t.n_up - 1].Cout)) return false;         // its output is written as pairs
  if (!conv_pair_sources_ok(w.dec_t.up[w.dec_t.n_up - 1].Cout, w.enc_b.conv3.Cout, D, 1)) return false;   // quantize_conv_b
  for (int i = 0; i < w.n_upsample; ++i) {
    if (convT_small_applicable(w.upsample[i].Cin, w.upsample[i].Cout)) return false;
    if (!conv_pair_sources_ok(w.upsample[i].Cin, 0, w.upsample[i].Cout, 4)) return false;
  }
  if (w.no_quantize) return false;   // the pair copies of quant_t / quant_b are made behind the codebook searches
  return decoder_pairs_ok(w.dec, D, D, true, true);
}

int run_quantizer(const isi_codebook_w &cb, const float *z, int64_t N, int64_t *idx, float *q,
                  int32_t *counts, float *sse_part, float *scalars2, int flags, hipStream_t st) {
  if (int rc0 = vq_zero_counts(counts, cb.K, st)) return rc0;
  // With ISI_CONV_F16X3 (the split-f16 mode) the K distances are computed on the f16 matrix pipe only to pick two
  // candidates; the decision between them is taken in fp32 (vq_nearest.hip) -- the same search as the fused kernel's.
  int rc = vq_nearest_f32(z, cb.codes_kd, cb.e2, idx, q, counts, sse_part, N, cb.D, cb.K, flags, st);
  if (rc) return rc;
  return vq_finalize_f32(sse_part, vq_num_partials(N), counts, cb.K, N, cb.D, scalars2, st);
}

// UnquantizedBottleneck.forward (bottleneck.py:107-119): diff = 0, perplexity = inf
int unquantized_scalars(float *scalars2, hipStream_t st) { return vq_unquantized_scalars(scalars2, st); }   // (a kernel: no memset nodes)

// quantize_conv (1x1) + codebook search in ONE launch (vq_nearest.hip: z stays in registers), with the pair-format
// copy of q written alongside.  `w1x1` = the 1x1 layer (its packed weight is followed by the blocked pair copy).
int run_quantizer_fused(const isi_codebook_w &cb, const isi_conv_w &w1x1, const isi_src &a, const isi_src *b, int B, int H,
                        int W, int64_t *idx, float *q, float *q_pair, int32_t *counts, float *sse_part, float *scalars2,
                        float *wfrag_ws, hipStream_t st, bool finalize = true, bool frag_packed = false) {
  const int Kpad = (int)round_up((size_t)w1x1.Cin, kBK);
  // frag_packed (isi_vqvae_w.w16 == 2): the weight's third section is its fragment-major copy and the caller has zeroed
  // both levels' histograms with one launch -- no pre-kernel in front of the search
  int rc = vq_conv1x1_nearest_f32(&a, b, w1x1.w + (size_t)w1x1.Cout * Kpad, w1x1.bias, cb.codes_kd, cb.e2, idx, q, q_pair,
                                  counts, sse_part, wfrag_ws, B, H, W, cb.D, cb.K, st, /*zero_counts*/ !frag_packed, nullptr,
                                  frag_packed ? w1x1.w + (size_t)2 * w1x1.Cout * Kpad : nullptr);
  if (rc || !finalize) return rc;     // (finalize = false: the caller finishes both levels with one launch, vq_finalize2_f32)
  const int64_t N = (int64_t)B * H * W;
  return vq_finalize_f32(sse_part, vq_num_partials(N), counts, cb.K, N, cb.D, scalars2, st);
}

int plan(const isi_vqvae_w &w, int mode, const float *x, int B, int H, int W,
         const isi_vqvae_out *out, Bump &ws, hipStream_t st) {
  Shapes sh;
  if (!compute_shapes(w, B, H, W, sh)) return invalid("vqvae: input too small or odd top/bottom grid");
  const int D = w.quantize_t.D;
  const bool dry = ws.dry;
  if (w.n_upsample != w.dec_t.n_up) return invalid("vqvae: upsample depth must equal dec_t depth");

  float *s0 = ws.floats(sh.max_act);
  float *s1 = ws.floats(sh.max_act);
  const int Rmax = std::max({w.enc_b.n_res ? w.enc_b.res3[0].Cout : 0, w.enc_t.n_res ? w.enc_t.res3[0].Cout : 0,
                             w.dec_t.n_res ? w.dec_t.res3[0].Cout : 0, w.dec.n_res ? w.dec.res3[0].Cout : 0, 1});
  float *hid = ws.floats((size_t)B * sh.Hb * sh.Wb * Rmax);
  float *enc_b = ws.floats((size_t)B * sh.Hb * sh.Wb * sh.Cb);
  float *enc_t = ws.floats((size_t)B * sh.Ht * sh.Wt * w.enc_t.conv3.Cout);
  float *dec_t = ws.floats((size_t)B * sh.Hb * sh.Wb * D);
  float *zbuf = ws.floats((size_t)B * sh.Hb * sh.Wb * D);
  float *up = ws.floats((size_t)B * sh.Hb * sh.Wb * D);
  float *up2 = ws.floats((size_t)B * sh.Hb * sh.Wb * D);
  float *q_t_ws = ws.floats((size_t)B * sh.Ht * sh.Wt * D);
  float *q_b_ws = ws.floats((size_t)B * sh.Hb * sh.Wq * D);
  // pair-format copies of the quantised maps: what the pair pipeline's convolutions read (the API tensors stay fp32)
  float *q_t_pair = ws.floats((size_t)B * sh.Ht * sh.Wt * D);
  float *q_b_pair = ws.floats((size_t)B * sh.Hb * sh.Wq * D);
  int64_t *id_t_ws = static_cast<int64_t *>(ws.take((size_t)B * sh.Ht * sh.Wt * sizeof(int64_t)));
  int64_t *id_b_ws = static_cast<int64_t *>(ws.take((size_t)B * sh.Hb * sh.Wq * sizeof(int64_t)));
  const int Kmax = std::max(w.quantize_t.K, w.quantize_b.K);
  // (the two histograms side by side: one launch zeroes both when the searches bring no pre-kernel of their own)
  int32_t *counts = static_cast<int32_t *>(ws.take((size_t)2 * Kmax * sizeof(int32_t)));
  float *sse_part = ws.floats(256);
  // the top level's statistics stay alive until the bottom search is done: one launch finishes both (fused path)
  int32_t *counts_top = counts + Kmax;
  float *sse_top = ws.floats(256);
  float *wfrag_ws = ws.floats((size_t)D * round_up((size_t)(w.quantize_conv_b.Cin > w.quantize_conv_t.Cin ? w.quantize_conv_b.Cin
                                                                                                            : w.quantize_conv_t.Cin), kBK));
  float *scal_ws = ws.floats(8);
  if (dry) return ISI_OK;
  if (ws.off > ws.cap) {
    set_last_error("vqvae: workspace too small");
    return ISI_E_WORKSPACE;
  }

  float *quant_t = out->quant_t ? out->quant_t : q_t_ws;
  float *quant_b = out->quant_b ? out->quant_b : q_b_ws;
  int64_t *id_t = out->id_t ? out->id_t : id_t_ws;
  int64_t *id_b = out->id_b ? out->id_b : id_b_ws;
  float *scal = out->scalars ? out->scalars : scal_ws;
  int rc;
  // precision: 0 all fp32 | 1 decoder split-bf16 (x3) | 2 everything x3 | 3 index-feeding layers six-term
  // split (x6: fp32-grade products), decoder x3 | 4 split-f16 everywhere (fp32-grade products, f16 range)
  const int f16 = ISI_CONV_F16X3 | (w.w16 ? ISI_CONV_W16 : 0);
  const int pf_enc = w.precision == 4 ? f16 : w.precision == 3 ? ISI_CONV_BF16X6
                   : w.precision == 2 ? ISI_CONV_BF16X3 : 0;   // feeds the quantisers
  const bool pairs = pairs_eligible(w);   // internal activations as split-f16 pairs (isi_hip.h: ISI_CONV_*_PAIR)
  const int pf_dec = w.precision == 4 ? f16 : w.precision >= 1 ? ISI_CONV_BF16X3 : 0;   // final decoder + upsample
  // quantize_conv_{t,b} fused into the codebook searches: the pair pipeline only (pair8 sources, blocked weights)
  const bool fuse_vq = pairs && !w.no_quantize && !knobs().no_vq_fusion;
  const bool frag_packed = fuse_vq && w.w16 == 2 && (mode & ISI_MODE_ENCODE);
  if (frag_packed) {
    rc = vq_zero_counts(counts, 2 * Kmax, st);
    if (rc) return rc;
  }
  bool bottom_pair_done = false;
  bool top_deferred = false;      // top-level scalars not written yet (fused path)
  int64_t N_top = 0;

  if (mode & ISI_MODE_ENCODE) {
    if (!x) return invalid("vqvae: x is null");
    Act eb, et;
    rc = run_encoder(w.enc_b, src_nchw(x, w.in_channel, H, W), B, H, W, s0, s1, hid, enc_b, eb, pf_enc, false, pairs, st);
    if (rc) return rc;
    rc = run_encoder(w.enc_t, src_nhwc(eb.p, eb.C, eb.H, eb.W), B, eb.H, eb.W, s0, s1, hid, enc_t, et, pf_enc, pairs,
                     pairs, st);
    if (rc) return rc;
    // quantize_conv_t + quantize_t (vqvae.py:260-263)
    {
      isi_src s = src_nhwc(et.p, et.C, et.H, et.W);
      if (fuse_vq && vq_conv1x1_fusable(et.C, 0, D, w.quantize_t.K)) {
        // (no fp32 map where the caller takes none and this call decodes from the pair copies)
        rc = run_quantizer_fused(w.quantize_t, w.quantize_conv_t, s, nullptr, B, et.H, et.W, id_t,
                                 (out->quant_t || !(mode & ISI_MODE_DECODE)) ? quant_t : nullptr, q_t_pair,
                                 counts_top, sse_top, scal + 0, wfrag_ws, st, /*finalize*/ false, frag_packed);
        if (rc) return rc;
        top_deferred = true;
        N_top = (int64_t)B * et.H * et.W;
        goto top_done;
      }
      // UnquantizedBottleneck (bottleneck.py:107-119): the 1x1 convolution's output IS quant_t
      isi_dst d = dst_nhwc(w.no_quantize ? quant_t : zbuf, D, et.H, et.W);
      rc = conv2d_f32(&s, nullptr, w.quantize_conv_t.w, w.quantize_conv_t.bias, nullptr, &d, B, et.H,
                      et.W, D, 1, 1, 1, 0, pf_enc | (pairs ? ISI_CONV_IN0_PAIR : 0), st);
      if (rc) return rc;
      rc = w.no_quantize ? unquantized_scalars(scal + 0, st)
                         : run_quantizer(w.quantize_t, zbuf, (int64_t)B * et.H * et.W, id_t, quant_t, counts_top,
                                         sse_part, scal + 0, pf_enc & ISI_CONV_F16X3, st);
      if (rc) return rc;
      if (pairs) {
        rc = pair_encode_f32(quant_t, q_t_pair, (int64_t)B * et.H * et.W * D, st);
        if (rc) return rc;
      }
    }
  top_done:;
    // dec_t (vqvae.py:265): [B,Ht,Wt,D] -> [B,Hb,2^n Wt,D]
    int Wd = et.W;
    for (int i = 0; i < w.dec_t.n_up; ++i) Wd *= 2;
    {
      isi_src s = src_nhwc(pairs ? q_t_pair : quant_t, D, et.H, et.W);
      isi_dst d = dst_nhwc(dec_t, w.dec_t.up[w.dec_t.n_up - 1].Cout, sh.Hb, Wd);
      rc = run_decoder(w.dec_t, s, nullptr, B, et.H, et.W, s0, s1, hid, d, pf_enc, pairs, false, pairs, pairs, st);
      if (rc) return rc;
    }
    // quantize_conv_b on cat([dec_t, enc_b]) cropped to Wq (vqvae.py:266-273)
    {
      const int Cd = w.dec_t.up[w.dec_t.n_up - 1].Cout;
      isi_src a = src_nhwc(dec_t, Cd, sh.Hb, sh.Wq, Wd);
      isi_src b = src_nhwc(eb.p, eb.C, sh.Hb, sh.Wq, eb.W);
      if (fuse_vq && vq_conv1x1_fusable(Cd, eb.C, D, w.quantize_b.K)) {
        rc = run_quantizer_fused(w.quantize_b, w.quantize_conv_b, a, &b, B, sh.Hb, sh.Wq, id_b,
                                 (out->quant_b || !(mode & ISI_MODE_DECODE)) ? quant_b : nullptr, q_b_pair, counts,
                                 sse_part, scal + 2, wfrag_ws, st, /*finalize*/ !top_deferred, frag_packed);
        if (rc) return rc;
        if (top_deferred) {
          const int64_t N_b = (int64_t)B * sh.Hb * sh.Wq;
          rc = vq_finalize2_f32(sse_top, vq_num_partials(N_top), counts_top, w.quantize_t.K, N_top, sse_part,
                                vq_num_partials(N_b), counts, w.quantize_b.K, N_b, D, scal, st);
          if (rc) return rc;
          top_deferred = false;
          bottom_pair_done = true;
          goto scalars_done;
        }
        bottom_pair_done = true;
        goto bottom_done;
      }
      if (top_deferred) {      // (the bottom level takes the two-launch path: finish the top level on its own)
        rc = vq_finalize_f32(sse_top, vq_num_partials(N_top), counts_top, w.quantize_t.K, N_top, D, scal + 0, st);
        if (rc) return rc;
        top_deferred = false;
      }
      isi_dst d = dst_nhwc(w.no_quantize ? quant_b : zbuf, D, sh.Hb, sh.Wq);
      rc = conv2d_f32(&a, &b, w.quantize_conv_b.w, w.quantize_conv_b.bias, nullptr, &d, B, sh.Hb,
                      sh.Wq, D, 1, 1, 1, 0, pf_enc | (pairs ? ISI_CONV_IN0_PAIR | ISI_CONV_IN1_PAIR : 0), st);
      if (rc) return rc;
      rc = w.no_quantize ? unquantized_scalars(scal + 2, st)
                         : run_quantizer(w.quantize_b, zbuf, (int64_t)B * sh.Hb * sh.Wq, id_b, quant_b, counts,
                                         sse_part, scal + 2, pf_enc & ISI_CONV_F16X3, st);
      if (rc) return rc;
    }
  bottom_done:;
    rc = vq_scalars_sum_f32(scal, st);       // scalars[4] = diff_t + diff_b
    if (rc) return rc;
  scalars_done:;
  }
  if (pairs && (mode & ISI_MODE_DECODE)) {
    // forward: quant_t was encoded behind its search (dec_t reads it); decode: both maps arrive as fp32
    if (!(mode & ISI_MODE_ENCODE)) {
      rc = pair_encode_f32(quant_t, q_t_pair, (int64_t)B * sh.Ht * sh.Wt * D, st);
      if (rc) return rc;
    }
    if (!bottom_pair_done) {
      rc = pair_encode_f32(quant_b, q_b_pair, (int64_t)B * sh.Hb * sh.Wq * D, st);
      if (rc) return rc;
    }
  }

  if (mode & ISI_MODE_DECODE) {
    if (!out->dec) return invalid("vqvae: dec output is null");
    if (!(mode & ISI_MODE_ENCODE) && (!out->quant_t || !out->quant_b))
      return invalid("vqvae: decode needs quant_t and quant_b");
    // upsample_top_to_bottom: plain transposed convs, no ReLU (vqvae.py:183-201,281)
    const float *cur = pairs ? q_t_pair : quant_t;
    int h = sh.Ht, ww = sh.Wt;
    for (int i = 0; i < w.n_upsample; ++i) {
      float *o = (i % 2 == 0) ? up : up2;
      isi_src s = src_nhwc(cur, w.upsample[i].Cin, h, ww);
      isi_dst d = dst_nhwc(o, w.upsample[i].Cout, 2 * h, 2 * ww);
      rc = conv_transpose2d_k4s2_f32(&s, w.upsample[i].w, w.upsample[i].bias, &d, B, h, ww,
                                     w.upsample[i].Cout, pf_dec | (pairs ? ISI_CONV_IN0_PAIR | ISI_CONV_OUT_PAIR : 0), st);
      if (rc) return rc;
      cur = o; h *= 2; ww *= 2;
    }
    if (h != sh.Hb || ww != sh.Wq) return invalid("vqvae: upsampled top grid != bottom grid");
    isi_src a = src_nhwc(cur, D, h, ww);
    isi_src b = src_nhwc(pairs ? q_b_pair : quant_b, D, sh.Hb, sh.Wq);
    int OHf = sh.Hb, OWf = sh.Wq;
    for (int i = 0; i < w.dec.n_up; ++i) { OHf *= 2; OWf *= 2; }
    isi_dst d = dst_nchw(out->dec, w.in_channel, OHf, OWf);
    rc = run_decoder(w.dec, a, &b, B, sh.Hb, sh.Wq, s0, s1, hid, d, pf_dec, pairs, pairs, pairs, false, st);
    if (rc) return rc;
  }
  return ISI_OK;
}

}  // namespace

int vqvae_pair_activations(const isi_vqvae_w *w) { return (w && pairs_eligible(*w)) ? 1 : 0; }

size_t vqvae_workspace_bytes(const isi_vqvae_w *w, int B, int H, int W) {
  if (!w || B <= 0 || H <= 0 || W <= 0) return 0;
  Bump ws{nullptr, 0, 0, true};
  if (plan(*w, ISI_MODE_FORWARD, nullptr, B, H, W, nullptr, ws, nullptr) != ISI_OK) return 0;
  return round_up(ws.off, 256);
}

int vqvae_run(const isi_vqvae_w *w, int mode, const float *x, int B, int H, int W,
              const isi_vqvae_out *out, void *workspace, size_t workspace_bytes,
              hipStream_t stream) {
  if (!w || !out || !workspace) return invalid("vqvae_run: null pointer");
  if (mode < 1 || mode > 3) return invalid("vqvae_run: bad mode");
  if (B <= 0 || H <= 0 || W <= 0) return invalid("vqvae_run: bad shape");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return invalid("vqvae_run: workspace must be 256-byte aligned");
  Bump ws{static_cast<char *>(workspace), workspace_bytes, 0, false};
  return plan(*w, mode, x, B, H, W, out, ws, stream);
}

}  // namespace isi
