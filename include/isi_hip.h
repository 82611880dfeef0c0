/*
 * isi_hip.h -- C-ABI of the MI355X (gfx950) compute library behind the
 * `interactive_spectrogram_inpainting.vqvae` module API.
 *
 * Boundary B4 of SURVEY.md section 8(b): plain `extern "C"` functions, raw
 * device pointers + sizes + the caller's HIP stream (passed as void*), no
 * allocation inside (workspace is passed in), no global mutable state,
 * re-entrant.  Every function returns 0 on success or a negative ISI_E_* code;
 * nothing is thrown across the boundary.
 *
 * The reference has no native code: each entry point replaces the stock
 * torch.nn op(s) reached from the cited reference line(s).  Paths are relative
 * to the reference repository root.
 *
 * All activations handled by this library are fp32.  Internal activations are
 * channels-last (N,H,W,C); tensors crossing the reference's API are N,C,H,W and
 * are described to the kernels by explicit element strides.
 */
#ifndef ISI_HIP_H
#define ISI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISI_OK 0
#define ISI_E_INVALID (-1)     /* bad argument / unsupported shape            */
#define ISI_E_LAUNCH (-2)      /* HIP launch or runtime failure               */
#define ISI_E_WORKSPACE (-3)   /* workspace too small                         */
#define ISI_E_UNSUPPORTED (-4) /* configuration outside the implemented range */

/* Library / build identification ("isi_hip gfx950 <n>"). */
const char *isi_version(void);
/* Last HIP error string seen by this thread's most recent failing call. */
const char *isi_last_error(void);

/* sizeof() of the structs below as compiled into the library, so that FFI
 * bindings can verify their own layout: which = 0 isi_src, 1 isi_dst,
 * 2 isi_conv_w, 3 isi_encoder_w, 4 isi_decoder_w, 5 isi_codebook_w,
 * 6 isi_vqvae_w, 7 isi_vqvae_out, 8 isi_attn_args, 9 isi_prior_w,
 * 10 isi_prior_state, 11 isi_attn_bwd_args.  Returns 0 for an unknown id. */
size_t isi_abi_struct_bytes(int which);

/* x = max(x, 0) in place over n floats: the in-place nn.ReLU with which
 * RosinalityResBlock overwrites its caller's tensor (encoder_decoder.py:23). */
int isi_relu_inplace_f32(float *x, int64_t n, void *stream);

/* ------------------------------------------------ audio <-> spectrogram */
/* HBM-bound stages of the GANSynth front-end the reference reaches through the absent
 * GANsynth_pytorch.spectrograms_helper (SpectrogramsHelper.to_spectrogram / to_audio;
 * utils/misc.py:10-29, train_vqvae.py:392-400, sample.py:599, flask_server.py:596,1016);
 * specification: oracle/spectrogram_oracle.py (parity unpinned).  The contractions
 * (windowed DFT, mel projections and their inverses) are isi_conv2d_f32 calls.
 *   stft  [B,T,2F]  real block | imaginary block, DC bin dropped (F = n_fft / 2)
 *   a, ph [B,T,F]   channels-last intermediates
 *   spec  [B,2,F,T] channel 0 log-magnitude (mel: log mel power), channel 1 instantaneous
 *                   frequency in units of pi
 * isi_spec_polar: mel = 0: a = log(|X| + 1e-6), ph = angle; mel = 1: a = |X|^2, ph = unwrapped angle.
 * isi_spec_finish: spec0 = mel ? log(a + 1e-6) : a, spec1 = wrapped time difference of ph / pi.
 * isi_spec_inverse_prepare: a = exp(spec0), ph = running sum over time of spec1 * pi.
 * isi_spec_to_stft: mag = mel ? sqrt(max(a, 0) + 1e-6) : a; stft = mag (cos ph | sin ph).
 * isi_overlap_add: audio[b,n] = sum_t frames[b,t,left + n - t*hop], frames [B,T,n_fft]. */
int isi_spec_polar_f32(const float *stft, float *a, float *ph, int B, int T, int F, int mel,
                       void *stream);
int isi_spec_finish_f32(const float *a, const float *ph, float *spec, int B, int T, int F,
                        int mel, void *stream);
int isi_spec_inverse_prepare_f32(const float *spec, float *a, float *ph, int B, int T, int F,
                                 void *stream);
int isi_spec_to_stft_f32(const float *a, const float *ph, float *stft, int64_t rows, int F,
                         int mel, void *stream);
int isi_overlap_add_f32(const float *frames, float *audio, int B, int T, int n_fft, int hop,
                        int left, int64_t L, void *stream);
/* Adjoints of isi_spec_to_stft_f32 and isi_spec_inverse_prepare_f32: the backward of `to_audio`, which the
 * reference obtains from autograd through GANsynth_pytorch when a `_fromSpectrogram` loss trains the VQ-VAE
 * (utils/losses/spectral.py:106-118, train_vqvae.py:88-96).  `a`, `ph` are the forward's inputs. */
int isi_spec_to_stft_bwd_f32(const float *a, const float *ph, const float *dx, float *da, float *dph,
                             int64_t rows, int F, int mel, void *stream);
int isi_spec_inverse_prepare_bwd_f32(const float *spec, const float *da, const float *dph,
                                     float *dspec, int B, int T, int F, void *stream);
/* One scale of the multi-scale spectral loss (reference utils/losses/spectral.py:78-118, where the STFTs come
 * from torch.stft): xp / xt = STFT rows [B*T][RS] (re block | im block of F bins, RS >= 2F) of the predicted and
 * the target audio.  fwd: per (sample, chunk of rows_per_block frames) the sums
 *   sum |dm|, sum dm^2, sum |dl|, sum dl^2,  dm = |Xp| - |Xt|, dl = log(|Xp| + eps) - log(|Xt| + eps)
 * into partial[B][ceil(T / rows_per_block)][4] (L1 / MSE means and per-sample L2 norms are sums of these).
 * bwd: dx = d/dXp of sum_b clin[b] g(dm) + clog[b] g(dl), g' = sign (kind 0) or identity (kind 1). */
int isi_spec_distance_fwd_f32(const float *xp, const float *xt, float *partial, int B, int T, int F,
                              int RS, float eps, int rows_per_block, void *stream);
int isi_spec_distance_bwd_f32(const float *xp, const float *xt, float *dx, const float *clin,
                              const float *clog, int B, int T, int F, int RS, float eps, int kind,
                              void *stream);
/* Per-channel affine map of a dense [B,2,H,W] spectrogram plus the masked-phase rule, one pass:
 * y0 = a0 x0 + b0; y1 = a1 x1 + b1, set to 0 where the log-magnitude (channel 0 of `ref` when given,
 * else y0) is <= thr (use_mask != 0).  Replaces GANsynth_pytorch's DataNormalizer.normalize /
 * denormalize and make_masked_phase_transform at the reference's call sites vqvae.py:254-255,297-302
 * (package absent from the tree; semantics in oracle/spectrogram_oracle.py).  HW % 4 == 0. */
int isi_spec_affine_mask_f32(const float *x, const float *ref, float *y, int64_t B, int64_t HW,
                             float a0, float b0, float a1, float b1, float thr, int use_mask,
                             void *stream);

/* ------------------------------------------------- measurement (bench.py) */
/* Per-launch timing with HIP events recorded on the launch stream.  State is
 * per calling thread; `on` = 1 starts (earlier records cleared), 0 pauses, 2 resumes without
 * clearing (bench.py times every n-th step only).  isi_prof_read blocks
 * until the recorded events have completed and returns, for one kernel id in
 * [0, isi_prof_num_kernels()), the number of launches, their summed duration
 * (ms) and their summed algorithmic FLOPs / bytes. */
/* Execution switches (A/B comparisons in tests, measurements): each is read from the environment variable of the
 * same name once, at first use of the library; isi_knob_set changes it afterwards (process-wide, not thread-safe
 * against concurrent launches).  Names: ISI_CONV_FLUSH (accumulator flush period of the LDS-DMA convolution, default
 * 3, 0 = never), ISI_NO_PAIRS, ISI_NO_CONV_FIRST, ISI_NO_VQ_FUSION, ISI_NO_CONV_PAIR_KERNEL,
 * ISI_NO_RESBLOCK_PAIR_KERNEL, ISI_NO_CONVT_PAIR_KERNEL, ISI_NO_TAIL_FUSION, ISI_NO_WGRAD_HALO (per-tap weight-gradient
 * kernel instead of the halo-staged one), ISI_NO_GEMM_KERNEL (linear layers and their weight gradients on the 1x1
 * convolution kernels instead of the GEMM kernels), ISI_CONV_PAIR_BM, ISI_CONV_PAIR_ALL,
 * ISI_CONV_TAP_MAJOR, ISI_RESPAIR_TH, ISI_RES_TH, ISI_CONVT_TH, ISI_CONVT_PAIR_TH, ISI_DECODE_NT, ISI_PRIOR_GRAPH,
 * ISI_DECODE_MFMA_ROWS (rows from which a decoding stage runs as matrix tiles, 16), ISI_DECODE_NO_STAT_HANDOFF (every
 * decoding launch forms the LayerNorm statistics of its residual rows itself), ISI_DECODE_STATS_GLOBAL (the tile kernel's
 * input statistics from a second load of the rows), ISI_ATTN_NO_FWD3 / ISI_ATTN_FWD3_ALL (the plane-staged attention
 * forward never / also for single-term products), ISI_ATTN_OLD_FWD, ISI_DECODE_ATTN_SEPARATE_SPLITS (two key splits of the
 * cached attention as two workgroups + a merging launch) -- all select between kernels that
 * compute the SAME result (to rounding).  The ablation switches ISI_CONV_ABLATE / ISI_VQ_DBG / ISI_RESPAIR_ABL
 * (wrong results by design) exist only in -DISI_MEASURE builds: the default build rejects them. */
int isi_knob_set(const char *name, int value);
int isi_knob_get(const char *name, int *value);
int isi_prof_enable(int on);
int isi_prof_num_kernels(void);
const char *isi_prof_kernel_name(int kernel_id);
int isi_prof_read(int kernel_id, long long *launches, double *ms, double *flops,
                  double *bytes);

/* ---------------------------------------------------------------- packing */

/* Conv2d weight [Cout,Cin,KH,KW] (torch layout) -> [Cout][Kpad] with
 * k = (kh*KW + kw)*Cin + ci, zero padded up to Kpad = roundup(KH*KW*Cin, 32).
 * Replaces nothing in the reference: layout preparation for
 * isi_conv2d_f32 (encoder_decoder.py:95-112,138; vqvae.py:149-150,175-177). */
int isi_pack_conv_weight_f32(const float *w, float *packed, int Cout, int Cin,
                             int KH, int KW, void *stream);
/* isi_pack_conv_weight_f32 followed by isi_split_conv_weight_f16 at packed + isi_packed_conv_weight_floats(...), in one
 * launch: the layout ISI_CONV_W16 / isi_vqvae_w.w16 expect (packed: twice the packed size, 16-byte aligned). */
int isi_pack_conv_weight_w16_f32(const float *w, float *packed, int Cout, int Cin,
                                 int KH, int KW, void *stream);
/* Packed weight of the INPUT-GRADIENT convolution of a stride-1 nn.Conv2d with weight w [Cout][Cin][KH][KW]
 * (autograd's conv backward-data behind train_vqvae.py:181): [Cin][KH*KW*Cout padded to 32] with the window rotated by
 * 180 degrees -- what isi_pack_conv_weight_f32 would make of w.flip(2, 3).transpose(0, 1), in one launch.  Run it
 * through isi_conv2d_f32 with padding K - 1 - p. */
int isi_pack_conv_dgrad_weight_f32(const float *w, float *packed, int Cout, int Cin, int KH,
                                   int KW, void *stream);
/* nn.Linear weight w [N][K] -> operand of its input-gradient GEMM dX = dY W (autograd's linear backward behind
 * train_autoregressive_model.py:199-201): out[0 .. K N) = W^T ([K][N] fp32: the "packed weight" of a 1x1 convolution
 * with K outputs and N inputs), out[K N .. 2 K N) = its split-bf16 pair copy (groups of 8 as {hi[8] | lo[8]}).  Pass it
 * to isi_conv2d_f32 / isi_linear_f32 with ISI_CONV_BF16X3 | ISI_CONV_W16_BF16: the GEMM kernel stages the weight tile by plain copies.  N, K
 * multiples of 32; out: 2 K N floats, 16-byte aligned. */
int isi_pack_linear_wT_bf16(const float *w, float *out, int N, int K, void *stream);
/* The same for n weights in ONE launch: `table` = n rows of four 64-bit words {w, out, N, K} in device memory (w, out:
 * addresses; every N, K a multiple of 32; out: 2 N K floats each).  blocks_per_weight: workgroups per weight (each walks its
 * weight's 32 x 32 tiles with that stride). */
int isi_pack_linear_wT_bf16_multi(const void *table, int n, int blocks_per_weight, void *stream);
/* Every weight layout a VQ-VAE training step needs, re-packed by ONE launch per optimizer step (vqvae/_train.py
 * PackGroup; the reference re-uses nn.Conv2d weights in place, train_vqvae.py:181-183 -- the layouts are this library's):
 * `table` = n entries of 8 int64 in device memory {kind, src, dst, d0, d1, KH, KW, aux}:
 *   kind 0 / 1: isi_pack_conv_weight_f32 / isi_pack_conv_weight_w16_f32 of a weight [d0 = Cout][d1 = Cin][KH][KW];
 *   kind 2: isi_pack_conv_dgrad_weight_f32 of the same; kind 3 / 4: isi_pack_convT_k4s2_weight_f32 of [d0 = Cin][d1 = Cout][4][4]
 *   in the phase-matrix layout (4: followed by isi_split_conv_weight_f16); kind 5: the few-channel layout + pair copy;
 *   kind 6: isi_pack_codebook_f32 of embed [d0 = D][d1 = K] (codes at dst, |e|^2 at aux).  Bit-identical to those calls. */
int isi_pack_multi(const void *table, int n, int blocks_per_entry, void *stream);
/* Split-f16 pair copy of a packed weight (any of the packed layouts; n_floats % 4 == 0, 16-byte aligned):
 * every quad of floats becomes {hi0..hi3 | lo0..lo3}, the f16 pieces of 1024 w, in the same 16 bytes. */
int isi_split_conv_weight_f16(const float *packed_w, float *out, int64_t n_floats, void *stream);
size_t isi_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW);

/* ConvTranspose2d(k=4,s=2,p=1) weight [Cin,Cout,4,4] -> 4 phase matrices
 * [phase=py*2+px][Cout][Kpad], k = (ty*2+tx)*Cin + ci, tap (ty,tx) of phase
 * (py,px) reading torch tap (ky,kx) = (3-py-2ty, 3-px-2tx).
 * When Cout <= 4 and Cin % 32 == 0 (the decoder's last layer) the layout is
 * instead [(ky*4+kx)*Cout+co][Cin] for the small-Cout kernel (GEMM + col2im gather); the choice is a
 * function of (Cin, Cout) only and isi_conv_transpose2d_k4s2_f32 applies the
 * same rule.  (encoder_decoder.py:199-215; vqvae.py:193-201). */
int isi_pack_convT_k4s2_weight_f32(const float *w, float *packed, int Cin,
                                   int Cout, void *stream);
size_t isi_packed_convT_k4s2_weight_floats(int Cin, int Cout);

/* Codebook `embed` [D,K] (column = code, bottleneck.py:47-51) -> row-major
 * [K][D] plus e2[K] = sum_d embed[d,k]^2 (the third term of bottleneck.py:59). */
int isi_pack_codebook_f32(const float *embed, float *codes_kd, float *e2, int D,
                          int K, void *stream);

/* ------------------------------------------------------------ convolution */

/* One source of a (possibly channel-concatenated) convolution input. */
typedef struct isi_src {
  const float *ptr;
  int C;                   /* channels taken from this source              */
  int64_t sn, sc, sh, sw;  /* element strides                              */
} isi_src;

typedef struct isi_dst {
  float *ptr;
  int64_t sn, sc, sh, sw; /* element strides of the FULL output tensor     */
} isi_dst;

/* `relu` argument of the convolution entry points is a flag word: */
#define ISI_CONV_RELU 1   /* rectify the output                                            */
#define ISI_CONV_BF16X3 2 /* opt-in: split-bf16 products (a_hi b_hi + a_hi b_lo + a_lo b_hi on
                           * the bf16 matrix pipe, fp32 accumulate; per-product relative error
                           * ~2^-16 instead of 2^-24).  Default is exact fp32.              */
#define ISI_CONV_BF16X6 4 /* opt-in: six-term split (x = hi + mid + lo exactly; hi.hi, hi.mid, mid.hi,
                           * hi.lo, lo.hi, mid.mid): every term above 2^-24 of a product is kept --
                           * fp32-grade products at 6/16 of the fp32 pipe's matrix time.   */
#define ISI_CONV_F16X3 8  /* opt-in: split-f16 products (two 11-bit pieces, three terms on the f16 matrix
                           * pipe): per-product relative error ~2^-23 like ISI_CONV_BF16X6 at the
                           * cost of ISI_CONV_BF16X3, inside f16's RANGE -- activations must satisfy
                           * |x| < 16384 and weights |w| < 64 (operands are scaled by 2^2 / 2^10 before
                           * the split; the scaling is undone exactly).  An operand outside the range
                           * gives Inf / NaN in the output, never a silently wrong value.        */

#define ISI_CONV_W16_BF16 256 /* with ISI_CONV_BF16X3 on a GEMM-shaped launch (isi_linear_f32, 1x1 isi_conv2d_f32): the packed
                           * weight is followed by its split-bf16 pair copy (isi_pack_linear_wT_bf16).  A bit of its own: the
                           * split-f16 copy of ISI_CONV_W16 at the same address would give silently wrong products.      */
#define ISI_CONV_W16 16   /* with ISI_CONV_F16X3: the packed weight is followed in memory by its split-f16 pair copy
                           * (isi_split_conv_weight_f16 written at packed_w + the packed size in floats: the weights'
                           * pieces are then prepared once instead of every time a tile is staged; same results bit
                           * for bit).  Ignored by the launches that do not run split products.          */

/* Activations as split-f16 PAIRS (with ISI_CONV_F16X3 | ISI_CONV_W16): an element's 4 bytes hold hi = f16(4 x) in the
 * low half and lo = f16(4 x - hi) in the high half, the pieces the split-f16 kernels otherwise compute from the fp32
 * value every time they stage it (once per tap and output tile).  A producer writes them once in its epilogue
 * (ISI_CONV_OUT_PAIR), a consumer reads them with a de-interleave (ISI_CONV_IN0_PAIR / ISI_CONV_IN1_PAIR per source).
 * Both need the launch to run the split-f16 kernel with pack-time weight pieces (ISI_CONV_F16X3 | ISI_CONV_W16, vectorised
 * channels-last sources, Cout > 32, K >= 128; the fused residual block likewise) -- the 2-channel first layer can write
 * them too; other launches return ISI_E_UNSUPPORTED.
 * Same matrix operands bit for bit; a residual-block skip connection reads (hi + lo) / 4, within 2^-24 of x.
 * isi_pair_encode_f32 / isi_pair_decode_f32 convert whole tensors (tests, boundaries).                        */
#define ISI_CONV_IN0_PAIR 32
#define ISI_CONV_IN1_PAIR 64
#define ISI_CONV_OUT_PAIR 128
int isi_pair_encode_f32(const float *x, float *pairs, int64_t n, void *stream);
int isi_pair_decode_f32(const float *pairs, float *x, int64_t n, void *stream);

/* Conv2d, groups=1, square stride, symmetric zero padding, fp32.
 *   out = [relu]( conv(cat(src0, src1), W) + bias [+ residual] )
 * src1.ptr may be NULL (single source); residual.ptr may be NULL.  `residual`
 * has the logical shape of the output.  `packed_w` comes from
 * isi_pack_conv_weight_f32.  Replaces nn.Conv2d (+ nn.ReLU, + the residual add
 * and torch.cat) at encoder_decoder.py:22-35,95-112,138 and vqvae.py:260,270-272,282. */
int isi_conv2d_f32(const isi_src *src0, const isi_src *src1,
                   const float *packed_w, const float *bias,
                   const isi_src *residual, const isi_dst *dst, int B, int H,
                   int W, int Cout, int KH, int KW, int stride, int pad,
                   int relu, void *stream);

/* ConvTranspose2d(kernel 4, stride 2, padding 1), groups=1, fp32:
 *   out[B, 2H, 2W, Cout] = [relu]( convT(src, W) + bias )
 * computed as four stride-1 2x2 phase convolutions.  `packed_w` comes from
 * isi_pack_convT_k4s2_weight_f32.  Replaces nn.ConvTranspose2d (+ nn.ReLU) at
 * encoder_decoder.py:199-215 and vqvae.py:193-201. */
int isi_conv_transpose2d_k4s2_f32(const isi_src *src, const float *packed_w,
                                  const float *bias, const isi_dst *dst, int B,
                                  int H, int W, int Cout, int relu,
                                  void *stream);

/* The decoder's tail in the pair pipeline (RosinalityDecoder, vqvae/encoder_decoder.py:196-209):
 *   ConvTranspose2d(Cin -> Cmid = 64, k4 s2 p1) + ReLU + ConvTranspose2d(64 -> Cout <= 2, k4 s2 p1)
 * without the [B, 2H, 2W, 64] activation between them: the first kernel projects every pixel of its result onto the
 * second layer's 16 taps x Cout outputs while it still sits in registers (Y', 32 floats per pixel), the second one is
 * left with the col2im sum.  in_pair: dense channels-last pair-format [B, H, W, Cin]; packed_w1: the packed phase
 * matrices of layer 1 followed by their split-f16 pair copy (isi_pack_convT_k4s2_weight_f32 + isi_split_conv_weight_f16);
 * packed_w2: the packed weight of layer 2 ([16 Cout][64], isi_pack_convT_k4s2_weight_f32); yprime_ws: workspace of
 * B * 2H * 2W * 32 floats; dst: [B, Cout, 4H, 4W] fp32, arbitrary strides.  Same products as the two calls of
 * isi_conv_transpose2d_k4s2_f32 (ISI_CONV_F16X3 | ISI_CONV_W16, pair hand-over), other summation order. */
int isi_decoder_tail_f32(const float *in_pair, const float *packed_w1, const float *bias1, const float *packed_w2,
                         const float *bias2, float *yprime_ws, const isi_dst *dst, int B, int H, int W, int Cin, int Cmid,
                         int Cout, void *stream);

/* Fused RosinalityResBlock on a rectified input r (encoder_decoder.py:22-35):
 *   out = [relu]( r + conv1x1(relu(conv3x3(r) + b3)) + b1 )
 * in/out dense channels-last [B,H,W,C]; packed_w3 = isi_pack_conv_weight_f32 of
 * conv.1 ([R,C,3,3]), packed_w1 of conv.3 ([C,R,1,1]).  Available when
 * isi_resblock_fusable(C, R) (C % 32 == 0, C <= 128, R <= 32); otherwise compose
 * two isi_conv2d_f32 calls. */
int isi_resblock_fusable(int C, int R);
int isi_resblock_f32(const float *in, const float *packed_w3, const float *b3,
                     const float *packed_w1, const float *b1, float *out, int B,
                     int H, int W, int C, int R, int relu, void *stream);

/* Training-mode forms of the pair pipeline's producers (train_vqvae.py:168-181: `out, latent_loss, ... = model(img)` under
 * model.train()): the same launches as isi_conv2d_f32 / isi_conv_transpose2d_k4s2_f32 / isi_resblock_f32 with
 * ISI_CONV_F16X3 | ISI_CONV_W16 | ISI_CONV_OUT_PAIR (pair-format sources, except the 2-channel first layer), which ALSO
 * write what the hand-written backward reads back -- `twin`: the output once more as dense channels-last fp32
 * [B,OH,OW,Cout] (weight-gradient operand, ReLU masks); `hidden` (residual block): relu(conv3x3(r) + b3) as dense fp32
 * [B,H,W,R].  The next layer keeps staging the pair tensor by LDS-DMA; nothing is converted twice.  Launches that would
 * not reach a kernel with the twin epilogue return ISI_E_UNSUPPORTED (ask the *_route predicates first: 1 = yes). */
int isi_conv2d_twin_f32(const isi_src *src0, const isi_src *src1, const float *packed_w, const float *bias,
                        const isi_dst *dst, float *twin, int B, int H, int W, int Cout, int KH, int KW, int stride,
                        int pad, int flags, void *stream);
int isi_conv_transpose2d_k4s2_twin_f32(const isi_src *src, const float *packed_w, const float *bias, const isi_dst *dst,
                                       float *twin, int B, int H, int W, int Cout, int flags, void *stream);
int isi_resblock_tape_f32(const float *in, const float *packed_w3, const float *b3, const float *packed_w1, const float *b1,
                          float *out, float *twin, float *hidden, int B, int H, int W, int C, int R, int flags, void *stream);
int isi_conv2d_pair_route(int C0, int C1, int Cout, int KH, int KW);
/* 1 if isi_conv_wgrad_torch_f32 (three-term products, dense channels-last operands) would run the halo-staged kernel for
 * this plain convolution -- the weight-gradient kernel that also reads PAIR-format sources: flags ISI_CONV_IN0_PAIR /
 * ISI_CONV_IN1_PAIR in the flag word of isi_conv_wgrad_f32 / _torch_f32 / _deferred_f32 (other launches: ISI_E_UNSUPPORTED). */
int isi_conv_wgrad_halo_route(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, int OH, int OW);
int isi_conv_transpose2d_pair_route(int Cin, int Cout);
int isi_resblock_pair_route(int B, int H, int W, int C, int R);

/* ----------------------------------------------------- training (backward) */
/* These replace the autograd kernels behind `loss.backward()` (train_vqvae.py:181)
 * and the in-forward EMA codebook update (vqvae/bottleneck.py:79-92).
 * Input gradients of the convolutions reuse the forward entry points:
 *   d/dx Conv2d(k, s=1)      = isi_conv2d_f32 with the flipped / transposed weight
 *   d/dx Conv2d(k4,s2,p1)    = isi_conv_transpose2d_k4s2_f32 with the same weight tensor
 *   d/dx ConvTranspose(k4s2) = isi_conv2d_f32(k4,s2,p1) with the same weight tensor. */

/* The two convolutions with a GATED epilogue: out = gate > 0 ? (conv + bias + residual) : 0, where `gate` is an fp32
 * tensor laid out exactly like `dst` (same strides).  With gate = the rectified activation a layer consumed, the
 * input-gradient convolution applies that ReLU's backward mask itself (autograd's threshold_backward behind
 * train_vqvae.py:181; encoder_decoder.py:24,31,95-112 `nn.ReLU`) instead of a separate pass over the gradient.
 * fp32 tensors only (no ISI_CONV_*_PAIR), one launch (the implicit-GEMM kernels). */
#define ISI_CONV_GATE_PAIR 512 /* the gate of a gated convolution is a pair-format tensor (ISI_CONV_OUT_PAIR layout, same shape
                                * and strides as the output): an element passes where its hi piece is positive -- the training
                                * forward's pair tensors serve as ReLU masks without an fp32 copy (Cout % 8 == 0) */
int isi_conv2d_gated_f32(const isi_src *src0, const isi_src *src1, const float *packed_w,
                         const float *bias, const isi_src *residual, const float *gate,
                         const isi_dst *dst, int B, int H, int W, int Cout, int KH, int KW,
                         int stride, int pad, int flags, void *stream);
int isi_conv_transpose2d_k4s2_gated_f32(const isi_src *src, const float *packed_w,
                                        const float *bias, const float *gate, const isi_dst *dst,
                                        int B, int H, int W, int Cout, int flags, void *stream);

/* Weight gradient dW (packed like the forward weight: [nphase][Cout][Kpad]) of a
 * convolution (transposed = 0) or ConvTranspose2d(k4,s2,p1) (transposed = 1) whose input
 * was cat(src0, src1) [B,*,H,W] and whose output gradient is dy, dense channels-last
 * [B, OH, OW, Cout] (transposed: [B, 2H, 2W, Cout]).  Deterministic split reduction;
 * workspace of isi_conv_wgrad_workspace_floats(Cout, K, M, nphase) floats with
 * K = KH*KW*Cin (transposed: 4*Cin), M = B*OH*OW (transposed: B*H*W), nphase = 1 (4).
 * db [Cout] (optional) receives the bias gradient = column sums of dy, computed on the
 * staged dY tiles at no extra pass.  `transposed` is a flag word: bit 0 = transposed
 * convolution, ISI_CONV_BF16X3 / ISI_CONV_BF16X6 select split-bf16 products (vectorised
 * channels-last operands only; other layouts use the fp32 pipe). */
size_t isi_conv_wgrad_workspace_floats(int Cout, int K, int M, int nphase);
int isi_conv_wgrad_f32(const isi_src *src0, const isi_src *src1, const float *dy,
                       float *dw_packed, float *db, float *workspace, size_t workspace_floats,
                       int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                       int transposed, void *stream);
/* The same gradient written in torch's weight layout [Cout][cin_keep][KH][KW] (the parameter's own `.grad`
 * storage: no packed intermediate, no permuting copy): the k padding and the input channels >= cin_keep (a source
 * zero-padded to a multiple of 4 channels) are dropped.  Plain convolutions only (bit 0 of `flags` clear); a
 * ConvTranspose2d's gradient is the adjoint stride-2 convolution's with the roles of input and output gradient
 * swapped (train_vqvae.py:181 via autograd; vqvae/_train.py conv_wgrad). */
int isi_conv_wgrad_torch_f32(const isi_src *src0, const isi_src *src1, const float *dy,
                             float *dw_torch, int cin_keep, float *db, float *workspace,
                             size_t workspace_floats, int B, int H, int W, int Cout, int KH, int KW,
                             int stride, int pad, int flags, void *stream);
/* The same call with its split REDUCTION deferred (round 5): the pixel-reduction GEMM is launched, the sum over its splits
 * (into dw_torch / db) is returned as up to 4 jobs in `jobs_out` instead of being launched behind it.  The caller runs the
 * jobs of many layers in one launch (isi_reduce_jobs_f32: the job table travels in the kernel arguments, 48 jobs per
 * launch) before anything reads the gradients -- the end of the step, or a data-parallel bucket's all-reduce -- and keeps
 * every `workspace` alive until then.  A VQ-VAE training step ran 31 reduction launches of ~9 us (train_vqvae.py:181). */
typedef struct isi_reduce_job {
  const float *partial;     /* [nsplit] slices of `stride` floats                                   */
  float *out;               /* n floats (or the torch-layout weight gradient when map_Kpad != 0)    */
  int64_t n, stride;
  int32_t nsplit, accumulate, vec;
  int32_t map_K, map_Kpad, map_cin, map_taps, map_keep;   /* packed [Cout][Kpad] -> torch [Cout][keep][taps] */
} isi_reduce_job;
int isi_conv_wgrad_deferred_f32(const isi_src *src0, const isi_src *src1, const float *dy, float *dw_torch, int cin_keep,
                                float *db, float *workspace, size_t workspace_floats, int B, int H, int W, int Cout, int KH,
                                int KW, int stride, int pad, int flags, void *stream, isi_reduce_job *jobs_out, int *n_jobs);
int isi_reduce_jobs_f32(const isi_reduce_job *jobs, int n_jobs, void *stream);
/* dy *= (y > 0) : ReLU backward through an output rectified in the producer's epilogue. */
int isi_relu_bwd_f32(float *dy, const float *y, int64_t n, void *stream);
/* a += alpha * b */
int isi_axpy_f32(float *a, const float *b, float alpha, int64_t n, void *stream);
/* Quantiser backward (bottleneck.py:94-95): dz = dq + 2 g_diff (z - q_st) / n. */
int isi_vq_bwd_f32(float *dz, const float *dq, const float *z, const float *q_st,
                   const float *g_diff, int64_t n, void *stream);
/* out[m][c] = (y[m][c] > 0) * (a[m][c] + b[m][c]) over M rows of C channels: two gradient contributions summed and masked by
 * the ReLU of the tensor they belong to in one pass; `a` may be a channel slice of a wider channels-last tensor (row stride
 * lda), b / y dense [M, C] or NULL (no second term / no mask).  Same, for the quantiser backward, with dq read through a
 * row stride (isi_vq_bwd_f32 on a channel slice without a dense copy first). */
/* src [B, C <= 4, H, W] (any strides) -> dense channels-last [B, H, W, 4] with the channels >= C zero: the 2-channel
 * spectrogram side as an operand of the vectorised weight-gradient kernels (train_vqvae.py:181 through vqvae/_train.py). */
int isi_pad_channels4_f32(const isi_src *src, float *out_nhwc4, int B, int H, int W, void *stream);
int isi_add_gate_rows_f32(float *out, const float *a, int64_t lda, const float *b, const float *y, int64_t M, int C, void *stream);
int isi_vq_bwd_rows_f32(float *dz, const float *dq, int64_t ldq, const float *z, const float *q_st, const float *g_diff,
                        int64_t M, int D, void *stream);
/* Reconstruction loss of the VQ-VAE training step (reference train_vqvae.py:168-176, `nn.MSELoss()`): out[0] = mean((a - b)^2)
 * over n elements (dense, 16-byte aligned), fixed summation order; `workspace`: isi_mse_loss_num_partials(n) floats.
 * Backward: da = 2 (a - b) g[0] / n, db = -da (either may be null); g: the incoming scalar gradient on the device. */
int isi_mse_loss_num_partials(int64_t n);
int isi_mse_loss_f32(const float *a, const float *b, int64_t n, float *workspace, float *out, void *stream);
int isi_mse_loss_bwd_f32(const float *a, const float *b, const float *g, int64_t n, float *da, float *db, void *stream);

/* out[C] = column sums of x [M, C] (bias gradients); workspace isi_colsum_num_partials(M)*C floats. */
int isi_colsum_num_partials(int64_t M);
int isi_colsum_f32(const float *x, int64_t x_stride, float *out, float *workspace, int64_t M,
                   int C, void *stream);
/* embed_sum_dk [D,K] = z^T @ onehot(idx) (bottleneck.py:83): the pixel-reduction GEMM of
 * isi_conv_wgrad_f32 with the one-hot operand generated on the fly (never materialised);
 * deterministic.  workspace: isi_vq_embed_sum_workspace_floats(D, K, N) floats. */
size_t isi_vq_embed_sum_workspace_floats(int D, int K, int64_t N);
int isi_vq_embed_sum_f32(const float *z, const int64_t *idx, float *embed_sum_dk,
                         float *workspace, size_t workspace_floats, int64_t N, int D, int K,
                         void *stream);
/* EMA codebook update (bottleneck.py:80-92) on the [D,K] buffers from batch statistics
 * counts [K] (float) and embed_sum_dk [D,K] (all-reduced across ranks by the caller). */
int isi_vq_ema_update_f32(float *embed, float *cluster_size, float *embed_avg,
                          const float *counts, const float *embed_sum_dk, int D, int K,
                          float decay, float eps, void *stream);

/* ------------------------------------------------------ transformer prior */
/* The prior's layers are instantiated by the reference from the absent package
 * VQCPCB.transformer.transformer_custom (priors/transformer.py:12-15,370-417);
 * their arithmetic is specified in oracle/prior_oracle.py (parity unpinned). */

/* Multi-head attention with relative-position logits, flash style:
 *   logit[i,j] = (q_i.k_j + q_i.e[h, r(i,j)]) * scale + mask(i,j)
 *   r(i,j) = floor(i/Cq) - floor(j/Ck) + Ek - 1 ;  out_i = softmax_j(logit) v_j
 * q/k/v/out are addressed as [seq, batch, head, head_dim] with element strides
 * (ss, sb, sh) and contiguous head_dim (16, 32 or 64).  rel_embeddings
 * [H, rel_rows, head_dim] may be NULL (no bias).  mask_mode: 0 none, 1 causal
 * (j <= i), 2 anti-causal (j >= i); dense_mask [Sq,Sk] additive, may be NULL. */
typedef struct isi_attn_args {
  const float *q, *k, *v, *rel_embeddings, *dense_mask;
  float *out;
  int Sq, Sk, B, H, head_dim;
  int64_t q_ss, q_sb, q_sh, k_ss, k_sb, k_sh, v_ss, v_sb, v_sh, o_ss, o_sb, o_sh;
  int Cq, Ck, Ek, rel_rows;
  int mask_mode;
  float scale;
  float *lse;   /* optional [B,H,Sq]: log-sum-exp of every query's logits (kept for the backward) */
  int precision; /* products of the three contractions: 0 = fp32 matrix pipe, 1 = three-term split-bf16
                  * (hi.hi + hi.lo + lo.hi on the bf16 pipe, fp32 accumulation; logits / softmax fp32),
                  * 2 = single-term bf16 (operands rounded to bf16, fp32 accumulation / logits / softmax),
                  * 3 = single-term f16 (operands rounded to f16: 11 significand bits, |q k v e| < 65504; error
                  * ~4e-4 of the output's maximum where mode 2 gives ~3e-3; its backward runs three-term products) */
  float *logits; /* optional [B,H,Sq,logits_ld] (16-bit modes only, precision >= 1): the forward stores the logit of every
                  * (query, key) pair the mask allows -- (q.k + q.e[r]) * scale * log2(e) + mask * log2(e), i.e. in units
                  * of exp2 -- and isi_rel_attention_bwd_f32, handed the same buffer, reads them instead of forming
                  * Q K^T, the band product Q E^T and its skew a second and a third time (60 % of the backward's time at
                  * S = 1025).  Entries of masked pairs are never read.  NULL: nothing is stored / everything is recomputed */
  int64_t logits_ld; /* row stride of `logits` in floats: a multiple of 4, >= Sk rounded up to a multiple of 32 */
  void *workspace;   /* optional, isi_rel_attention_workspace_bytes(args) bytes, 256-byte aligned, forward only: with it the
                      * 16-bit modes (precision >= 1, Cq = Ck = 1, head_dim 32 / 64) first split K, V and the table into
                      * 16-bit planes (one launch) and then run the LDS-DMA-staged kernel (rel_attention_fwd3.hip) on them;
                      * same arithmetic and results as without.  NULL: the register-staged kernels. */
  size_t workspace_bytes;
} isi_attn_args;
int isi_rel_attention_f32(const isi_attn_args *args, void *stream);
/* 0 when the call would not use a workspace (precision 0, several channels per event, head_dim 16, planes >= 2 GiB) */
size_t isi_rel_attention_workspace_bytes(const isi_attn_args *args);

/* Backward of isi_rel_attention_f32 (replaces autograd through the attention of the absent
 * VQCPCB.transformer.transformer_custom behind `loss.backward()`,
 * train_autoregressive_model.py:257).  fwd = the forward call's arguments with out = its
 * output and lse = the log-sum-exp it wrote.  d_out has the strides of out; dq / dk / dv
 * are written with the strides of q / k / v (e.g. slices of one [S,B,3d] buffer); d_rel
 * [H, rel_rows, head_dim] is overwritten (NULL iff rel_embeddings is NULL).  workspace:
 * isi_rel_attention_bwd_workspace_floats(&fwd) floats, 16-byte aligned.  dK, dV, dQ and dE
 * are reduced in a fixed order (no cross-wave float atomics). */
typedef struct isi_attn_bwd_args {
  isi_attn_args fwd;
  const float *d_out;
  float *dq, *dk, *dv, *d_rel;
  float *workspace;
  size_t workspace_floats;
} isi_attn_bwd_args;
size_t isi_rel_attention_bwd_workspace_floats(const isi_attn_args *fwd);
int isi_rel_attention_bwd_f32(const isi_attn_bwd_args *args, void *stream);

/* isi_layernorm_f32 / isi_layernorm_bwd_f32 with the dropout that precedes the residual add in the absent package's
 * layers folded in (train mode): out = LayerNorm(drop(x) + residual), drop = inverted dropout whose keep mask is a
 * counter-based hash of (drop_seed, flat element index) -- the same function the backward evaluates again, so no mask
 * is stored and no dropout kernel runs.  The backward returns dz = d loss / d (drop(x) + residual) (the residual's
 * gradient) and dx = the gradient of x (dz where kept, scaled by 1 / (1 - drop_p)).  drop_p = 0: the plain kernels. */
int isi_layernorm_dropout_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                              float *out, int64_t M, int D, float eps, float drop_p, uint64_t drop_seed,
                              void *stream);
int isi_layernorm_dropout_bwd_f32(const float *x, const float *residual, const float *gamma, const float *dy,
                                  float *dz, float *dx, float *dgamma, float *dbeta, float *workspace, int64_t M,
                                  int D, float eps, float drop_p, uint64_t drop_seed, void *stream);

/* A device-resident 64-bit term added to the seed of every fused dropout (the linear layer's epilogue, the LayerNorm
 * kernels): a training step replayed from a HIP graph carries its seeds as launch constants, so the owner of the graph
 * advances this counter between replays (one element-wise add inside the graph) to draw fresh masks per step.  The
 * forward and backward launches of one step read the same value.  NULL (the default) = no such term.  Process-wide. */
int isi_set_dropout_seed_base(const void *device_u64);

/* One linear layer of the prior (`nn.Linear` inside the absent package's layers, priors/transformer.py:370-417): rows
 *   out[m, n] = sum_k x[m, k] w[n, k] + bias[n] (+ residual[m, n]) (ReLU) (gate) (dropout)
 * on the split-product GEMM kernel, with the element-wise tails of a training step folded into its epilogue:
 *   gate       out = gate[m, n] > 0 ? value * gate_scale : 0  -- a ReLU's backward mask (and the inverse keep probability
 *              of a dropout that followed the ReLU) applied by the input-gradient GEMM;
 *   drop_p     inverted dropout of the (rectified) output: element (m, n) is kept iff a counter-based hash of
 *              (drop_seed, m * ldo + n) clears drop_p, kept values are scaled by 1 / (1 - drop_p).  The mask is a pure
 *              function of (seed, index): nothing is stored; a dropped element of a rectified output is 0 and gates its
 *              own gradient.  (`F.dropout` in train mode: same distribution, not torch's random stream.)
 * flags: ISI_CONV_RELU | one of ISI_CONV_BF16X3 / ISI_CONV_F16X3 | ISI_CONV_W16 as for isi_conv2d_f32.  Shapes the GEMM
 * kernel does not take (K % 32, K < 128, N <= 32, M < 256) return ISI_E_UNSUPPORTED: the caller keeps isi_conv2d_f32. */
typedef struct isi_linear_args {
  const float *x;
  int64_t ldx;
  const float *packed_w, *bias, *residual;
  int64_t ldr;
  const float *gate;
  int64_t ldg;
  float gate_scale;
  float *out;
  int64_t ldo;
  int M, N, K, flags;
  float drop_p;
  uint64_t drop_seed;
} isi_linear_args;
int isi_linear_f32(const isi_linear_args *args, void *stream);

/* Backward of isi_layernorm_f32: dz = d loss / d (x + residual) (the gradient of both x and
 * residual), dgamma / dbeta [D] (overwritten; deterministic two-stage reduction).
 * workspace: isi_layernorm_bwd_workspace_floats(M, D) floats. */
size_t isi_layernorm_bwd_workspace_floats(int64_t M, int D);
int isi_layernorm_bwd_f32(const float *x, const float *residual, const float *gamma, const float *dy,
                          float *dz, float *dgamma, float *dbeta, float *workspace, int64_t M,
                          int D, float eps, void *stream);

/* LabelSmoothingLoss (utils/losses/prediction.py:5-20) over rows of logits [M,K] with
 * int64 targets [M]: row_loss[m] = sum_k -true_dist[k] log_softmax(logits[m])[k]; when
 * dlogits is not NULL also (softmax - true_dist) * grad_scale (the gradient of the mean
 * loss for grad_scale = upstream / M), in the same pass. */
int isi_label_smoothing_loss_f32(const float *logits, const int64_t *target, float *row_loss,
                                 float *dlogits, int64_t M, int K, int num_classes,
                                 float smoothing, float grad_scale, void *stream);


/* out[m,:] = LayerNorm(x[m,:] + residual[m,:]) * gamma + beta  (residual may be NULL). */
int isi_layernorm_f32(const float *x, const float *residual, const float *gamma,
                      const float *beta, float *out, int64_t M, int D, float eps,
                      void *stream);

/* Single-token decoding: out[m,n] = [relu](x[m,:].W[n,:] + bias[n] + residual[m,n])
 * for M <= 8 rows; W in torch layout [N,K].  (Whole sequences go through
 * isi_conv2d_f32 with a 1x1 kernel, which is a GEMM.) */
int isi_linear_rows_f32(const float *x, int x_stride, const float *W, const float *bias,
                        const float *residual, int res_stride, float *out, int out_stride,
                        int M, int N, int K, int relu, void *stream);

/* One stage of the KV-cached decoding loop on M <= 256 new rows (sample.py:268-305 -> priors/transformer.py:763-774 evaluate a
 * whole decoder pass per token; the loop evaluates its stages on the new rows only):
 *   out[m][n] = act( sum_k LN(x)[m][k] W[n][k] + bias[n] + LN_res(res)[m][n] )
 * LN = LayerNorm over the row with (ln_g, ln_b) (nullable: none), LN_res likewise with (res_g, res_b) over the N features of
 * the residual row (nullable; res nullable), act = ReLU when `relu`.  Kernels by row count: one row = one memory round trip
 * (GEMV), up to ISI_DECODE_MFMA_ROWS (16) rows the register-resident GEMV looping over groups of rows (each row: the operations and the order
 * of its one-row result), beyond that 32-row tiles on the fp32 matrix pipe (exact fp32 products, another summation order).
 * K a multiple of 4 (8 for the tiles), <= 2048; x, W 16-byte aligned.  `workspace`: isi_decode_stage_workspace_floats(M, N, K)
 * floats, 16-byte aligned (K-chunk sums of the tile kernel; may be null: the chunks then run one after the other). */
size_t isi_decode_stage_workspace_floats(int M, int N, int K);
int isi_decode_stage_f32(const float *x, int x_stride, const float *ln_g, const float *ln_b, const float *W, const float *bias,
                         const float *res, int res_stride, const float *res_g, const float *res_b, float *out, int out_stride,
                         int M, int N, int K, int relu, float eps, float *workspace, size_t workspace_floats, void *stream);

/* Decoding step: ONE query row per (batch, head) at sequence position q_pos
 * against args->Sk cached keys/values (same logits as isi_rel_attention_f32;
 * args->Sq, q_ss, o_ss, mask_mode and dense_mask are ignored: the caller passes
 * Sk = q_pos + 1 for causal self-attention).  With a workspace of
 * isi_rel_attention_decode_workspace_floats(B, H, head_dim) floats the keys are split
 * over up to 8 workgroups per (batch, head) and merged by a second small kernel;
 * workspace may be NULL (single workgroup per head). */
size_t isi_rel_attention_decode_workspace_floats(int B, int H, int head_dim);
int isi_rel_attention_decode_f32(const isi_attn_args *args, int q_pos, float *workspace,
                                 void *stream);

/* One categorical draw per row (sample.py:286-295): logits/temperature ->
 * top_k_top_p_filtering (sample.py:36-65) -> softmax -> inverse-CDF draw with the
 * host-supplied uniform u[row] (first index whose cumulative probability exceeds
 * u * total).  filtered [rows, n] (optional) receives the filtered logits
 * (-inf where removed).  n <= 1024. */
int isi_sample_row_f32(const float *logits, int stride, int rows, int n, float temperature,
                       int top_k, float top_p, const float *u, int64_t *out,
                       float *filtered, void *stream);

/* Native key/value-cached sampling loop of the decoder (replaces the per-token
 * full decoder pass of sample.py:268-305).  Weights are the torch-layout
 * parameters themselves ([N,K] row-major), no packing. */
#define ISI_MAX_LAYERS 16
typedef struct isi_attn_w {
  const float *in_proj_weight, *in_proj_bias;   /* [3d,d], [3d]                      */
  const float *out_proj_weight, *out_proj_bias; /* [d,d], [d]                        */
  const float *rel_embeddings;                  /* [H, rel_rows, d/H] or NULL        */
  int rel_rows;
} isi_attn_w;
typedef struct isi_decoder_layer_w {
  isi_attn_w self_attn, cross_attn;
  const float *linear1_w, *linear1_b, *linear2_w, *linear2_b;
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
} isi_decoder_layer_w;
typedef struct isi_prior_w {
  int d_model, nhead, dim_feedforward, n_layers, n_class;
  int Cd, Ed, Ce, Ee;          /* channels / events (incl. start symbol) of decoder and encoder sequences */
  isi_decoder_layer_w layers[ISI_MAX_LAYERS];
  const float *logits_w, *logits_b;             /* project_transformer_outputs_to_logits */
  const float *embed_table;                     /* [n_class, eff_dim] = Linear(Embedding) of target tokens */
  int eff_dim;
} isi_prior_w;
typedef struct isi_prior_state {
  float *x_seq;            /* [S_t, B, d] decoder input rows; rows of sampled tokens are rewritten    */
  float *kv_cache;         /* [n_layers, S_t, B, 2d] self-attention keys|values                       */
  const float *memory_kv;  /* [n_layers, S_src, B, 2d] projected encoder memory keys|values           */
  int64_t *codes;          /* [B, S] codes in sequence order (in: known codes, out: sampled)          */
  const uint8_t *mask;     /* [S] HOST array, 1 = sample this position                                */
  const float *uniforms;   /* [S, B] device, uniforms in [0,1)                                        */
  float *scratch;          /* device, isi_prior_decode_scratch_floats(w, B) floats                    */
  size_t scratch_floats;
  int S_t, S_src, S, B, start_len;
} isi_prior_state;
size_t isi_prior_decode_scratch_floats(const isi_prior_w *w, int B);
/* Enqueues positions [p_begin, p_end) of the decoder (one new row each, all layers),
 * and for every masked position the logits head, the draw (isi_sample_row_f32
 * semantics) and the write of the sampled token into codes / the next input row.
 * B <= 256 (rows are processed in groups of 8).  Execution switch ISI_PRIOR_GRAPH = W (default 8): windows of W consecutive
 * positions that are all sampled / all kept are replayed from hipGraphs captured on a private stream (every
 * position-dependent address read from a device counter): the host leaves the loop.  The executables are kept by the
 * library, keyed on the bytes of *w, *state (the host mask aside), the sampling parameters and the switches -- a later call
 * with equal arguments replays them, nothing is captured again and the call never waits for the stream (at most 8 argument
 * sets are kept; the least recently used one is dropped once its last replay has finished).  While the CALLER is capturing `stream`
 * and with ISI_PRIOR_GRAPH = 0: ~66 direct launches per position.  Same kernels, same codes either way. */
int isi_prior_sample_run(const isi_prior_w *w, const isi_prior_state *state, int p_begin,
                         int p_end, float temperature, int top_k, float top_p, void *stream);

/* ----------------------------------------------------------- quantization */

/* L2 nearest-neighbour vector quantisation, eval mode
 * (QuantizedBottleneck.forward, bottleneck.py:53-61,75-77,94-101).
 *   z        [N, D]   channels-last pre-quantisation vectors
 *   codes_kd [K, D], e2 [K]  from isi_pack_codebook_f32
 *   idx_out  [N] int64   argmin_k ( (|z|^2 - 2 z.e_k) + |e_k|^2 ), ties -> lowest k
 *   q_out    [N, D]   z + (e_idx - z)      (straight-through value, :95)
 *   counts   [K] int32  histogram of idx (caller zeroes; accumulated)
 *   sse_part [isi_vq_num_partials(N)] fp32 per-workgroup sum (e_idx - z)^2
 * Requirements: D in {8,16,32,64}; with Kp = K rounded up to 32: Kp*(D+4)*4 + Kp*8 <= 150 KiB
 * (the LDS image is padded with rows that can never win). */
int isi_vq_nearest_f32(const float *z, const float *codes_kd, const float *e2,
                       int64_t *idx_out, float *q_out, int32_t *counts,
                       float *sse_part, int64_t N, int D, int K, void *stream);
/* Same, with a flag word: ISI_CONV_F16X3 computes z.e_k with split-f16 products (three f16 MFMA terms, per-product
 * error ~2^-23; |z| < 16384, |e| < 64) when D == 64 and the codebook's two f16 planes fit in LDS (K <= 512), five
 * times fewer matrix-pipe cycles than the exact-fp32 products that bound isi_vq_nearest_f32.  0 = exact.
 * A vector with a non-finite component (or distances) gets idx -1, q = NaN, and makes the squared error NaN; so does
 * every vector when a code vector is not finite.  FINITE code vectors beyond the range (|e| >= 64: the unused codes of a
 * trained codebook) are exact: they are kept out of the f16 search and the fp32 decision proves per vector that none of
 * them can win ((min |e_far| - |z|)^2 > winner's distance) or compares the winner with each of them in fp32. */
int isi_vq_nearest_flags_f32(const float *z, const float *codes_kd, const float *e2,
                             int64_t *idx_out, float *q_out, int32_t *counts,
                             float *sse_part, int64_t N, int D, int K, int flags, void *stream);
int isi_vq_num_partials(int64_t N);

/* quantize_conv (1x1, vqvae.py:260,272: Conv2d(C0 [+ C1] -> D, 1) on cat(src0, src1)) FUSED with the search above
 * (SURVEY K4 + K5): z = W a + bias is produced in registers and consumed by the distance products, it never goes to
 * memory.  Pair pipeline only: sources in the pair format (ISI_CONV_*_PAIR, channels-last, 32-channel multiples),
 * `packed_w16` = the packed 1x1 weight's blocked pair copy (ISI_CONV_W16: packed weight + D * Kpad floats), D = 64.
 * Pixel n = (b H + y) W + x reads src0 / src1 through their strides.  Outputs as isi_vq_nearest_f32 (bit-identical
 * to isi_conv2d_f32(ISI_CONV_F16X3 | ISI_CONV_W16) followed by it); `q_pair_out` (nullable): pair-format copy of q.
 * `workspace`: isi_vq_conv1x1_workspace_floats(C0, C1, D) floats, 16-byte aligned (a fragment-major copy of the
 * weight is written there by a small pre-kernel). */
int isi_vq_conv1x1_nearest_f32(const isi_src *src0, const isi_src *src1, const float *packed_w16, const float *bias,
                               const float *codes_kd, const float *e2, int64_t *idx_out, float *q_out,
                               float *q_pair_out, int32_t *counts, float *sse_part, float *workspace, int B, int H,
                               int W, int D, int K, void *stream);
/* The same launch for the TRAINING step (train_vqvae.py:168-192 -> bottleneck.py:53-101): `z_out` ([N][64] fp32,
 * nullable) also receives z -- the commitment gradient 2 (z - q) / numel and the EMA sums of bottleneck.py:75-92 read it --
 * and `counts` is zeroed by the launch's pre-kernel (isi_vq_conv1x1_nearest_f32 expects zeroed counts). */
int isi_vq_conv1x1_nearest_tape_f32(const isi_src *src0, const isi_src *src1, const float *packed_w16, const float *bias,
                                    const float *codes_kd, const float *e2, int64_t *idx_out, float *q_out,
                                    float *q_pair_out, float *z_out, int32_t *counts, float *sse_part, float *workspace,
                                    int B, int H, int W, int D, int K, void *stream);
size_t isi_vq_conv1x1_workspace_floats(int C0, int C1, int D);
/* Pack-time half of the fused launch: the fragment-major copy ([D][Kpad] floats) of a packed 1x1 weight's pair copy.  A
 * weight laid out {packed [D][Kpad] | pair copy | fragments} with isi_vqvae_w.w16 = 2 lets isi_vqvae_run skip the pre-kernel in
 * front of each search (its two histograms are then zeroed by one launch). */
int isi_vq_pack_fragments_f32(const float *packed_w16, float *frag_out, int Kpad, void *stream);
/* 1 when the fused launch covers the shape (D = 64, 32-channel multiples, C0 + C1 <= 256, both code planes in LDS) */
int isi_vq_conv1x1_fusable(int C0, int C1, int D, int K);

/* diff = sum(sse_part)/(N*D); perplexity = exp(-sum p log(max(p,1e-7))),
 * p = counts/N  (bottleneck.py:94,97-100).  Writes out2[0]=diff, out2[1]=perplexity. */
int isi_vq_finalize_f32(const float *sse_part, int n_part, const int32_t *counts,
                        int K, int64_t N, int D, float *out2, void *stream);

/* embed_code: out[N, D] = codes_kd[idx[n], :]  (bottleneck.py:103-104).
 * Index range is not reported from the device: callers validate 0 <= idx < K
 * (out-of-range values are clamped so that no read leaves the codebook). */
int isi_embed_code_f32(const int64_t *idx, const float *codes_kd, float *out,
                       int64_t N, int D, int K, void *stream);

/* ------------------------------------------------- whole VQ-VAE-2 forward */

#define ISI_MAX_STAGES 4
#define ISI_MAX_RES 8

typedef struct isi_conv_w {
  const float *w;    /* packed weight */
  const float *bias; /* [Cout]        */
  int Cin, Cout;
} isi_conv_w;

typedef struct isi_encoder_w { /* RosinalityEncoder, encoder_decoder.py:38-126 */
  int n_down;                          /* number of k4 s2 convs              */
  isi_conv_w down[ISI_MAX_STAGES];
  isi_conv_w conv3;                    /* k3 conv after the strided stack    */
  int n_res;
  isi_conv_w res3[ISI_MAX_RES];        /* ResBlock conv.1 (k3, C -> R)       */
  isi_conv_w res1[ISI_MAX_RES];        /* ResBlock conv.3 (k1, R -> C)       */
} isi_encoder_w;

typedef struct isi_decoder_w { /* RosinalityDecoder, encoder_decoder.py:129-227 */
  isi_conv_w conv3;
  int n_res;
  isi_conv_w res3[ISI_MAX_RES];
  isi_conv_w res1[ISI_MAX_RES];
  int n_up;
  isi_conv_w up[ISI_MAX_STAGES];       /* packed transposed convs            */
} isi_decoder_w;

typedef struct isi_codebook_w {
  const float *codes_kd; /* [K][D] */
  const float *e2;       /* [K]    */
  int D, K;
} isi_codebook_w;

typedef struct isi_vqvae_w { /* VQVAE.__init__, vqvae.py:126-216 */
  int in_channel;
  isi_encoder_w enc_b, enc_t;
  isi_conv_w quantize_conv_t, quantize_conv_b;
  isi_codebook_w quantize_t, quantize_b;
  isi_decoder_w dec_t, dec;
  int n_upsample;
  isi_conv_w upsample[ISI_MAX_STAGES];
  int w16;       /* 1: every packed convolution weight above is followed by its isi_split_conv_weight_f16 copy
                  * (used when precision == 4); 2: ... and quantize_conv_t / _b by a third section, the fragment-major
                  * copy of isi_vq_pack_fragments_f32                                                  */
  int precision; /* products of the convolutions (data and accumulation are always fp32):
                  * 0: fp32 matrix pipe everywhere.  1: ISI_CONV_BF16X3 in `dec` and `upsample` only
                  * (no code index depends on them).  2: ISI_CONV_BF16X3 in every convolution (near-tie
                  * indices move).  3: ISI_CONV_BF16X6 (fp32-grade six-term split) in every layer that
                  * feeds a code index, ISI_CONV_BF16X3 in `dec` and `upsample`.  4: ISI_CONV_F16X3
                  * (fp32-grade three-term split-f16 products, f16 operand range) in every convolution. */
  int no_quantize; /* 1: UnquantizedBottleneck (bottleneck.py:107-119, selected at vqvae.py:152-160): the two
                    * codebook searches are skipped, quant_t / quant_b receive the 1x1 convolutions' outputs,
                    * id_t / id_b are not written, scalars = {0, inf, 0, inf}; the codebooks may be null */
} isi_vqvae_w;

/* Outputs of VQVAE.encode / forward (vqvae.py:245-278).  Any pointer may be
 * NULL except the ones the requested mode needs. */
typedef struct isi_vqvae_out {
  float *dec;        /* [B, in_channel, H, W]  NCHW      (forward/decode)   */
  float *quant_t;    /* [B, Ht, Wt, D]  channels-last    (encode/forward; null in a FORWARD call: the map stays in the
                      * pair format the decoders read and no fp32 copy is written -- the fused search's stores bound it) */
  float *quant_b;    /* [B, Hb, Wb, D]  channels-last                        */
  int64_t *id_t;     /* [B, Ht, Wt]                                          */
  int64_t *id_b;     /* [B, Hb, Wb]                                          */
  float *scalars;    /* [5]: diff_t, perplexity_t, diff_b, perplexity_b, diff_t + diff_b (what VQVAE.forward returns as
                      * `diff`, vqvae.py:263,275,277)                                                        */
} isi_vqvae_out;

#define ISI_MODE_ENCODE 1  /* VQVAE.encode        vqvae.py:251-278 */
#define ISI_MODE_DECODE 2  /* VQVAE.decode        vqvae.py:280-286 (quant_t/b are inputs) */
#define ISI_MODE_FORWARD 3 /* VQVAE.forward       vqvae.py:245-249 */

/* Bytes of scratch `isi_vqvae_run` needs for this shape. */
/* 1 if isi_vqvae_run keeps this model's internal activations in the split-f16 pair format (precision 4, w16, every
 * layer that would read a pair tensor able to: see ISI_CONV_*_PAIR), 0 if it runs on fp32 activations. */
int isi_vqvae_pair_activations(const isi_vqvae_w *w);
size_t isi_vqvae_workspace_bytes(const isi_vqvae_w *w, int B, int H, int W);

/* x: [B, in_channel, H, W] NCHW contiguous fp32 (ignored in DECODE mode).
 * H and W must be divisible by the total down-sampling factor. */
int isi_vqvae_run(const isi_vqvae_w *w, int mode, const float *x, int B, int H,
                  int W, const isi_vqvae_out *out, void *workspace,
                  size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISI_HIP_H */
