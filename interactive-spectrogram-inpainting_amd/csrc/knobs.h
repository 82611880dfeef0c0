// Execution switches of the library (A/B comparisons in the tests, measurements).  Every switch is read from the
// environment ONCE, when the library is first used, and can afterwards be changed only through isi_knob_set (the
// tests' A/B entry point) -- no getenv on the launch paths, and a stray environment variable set after start-up
// changes nothing.  Switches that produce WRONG RESULTS by design (ablations: ISI_CONV_ABLATE, ISI_VQ_DBG,
// ISI_RESPAIR_ABL) exist only in -DISI_MEASURE builds (`make EXTRA=-DISI_MEASURE`); the default build ignores them.
#pragma once

namespace isi {

struct Knobs {
  int conv_flush;               // ISI_CONV_FLUSH: chunks between accumulator flushes of conv_pair_kernel (3; 0 = never)
  int no_pairs;                 // ISI_NO_PAIRS: fused forward on fp32 activations instead of the pair pipeline
  int no_conv_first;            // ISI_NO_CONV_FIRST: generic gather kernel for the 2-channel first layer
  int no_vq_fusion;             // ISI_NO_VQ_FUSION: quantize_conv and search as two launches
  int no_conv_pair_kernel;      // ISI_NO_CONV_PAIR_KERNEL: register-staged kernel instead of the LDS-DMA one
  int no_resblock_pair_kernel;  // ISI_NO_RESBLOCK_PAIR_KERNEL
  int no_convt_pair_kernel;     // ISI_NO_CONVT_PAIR_KERNEL: four phase launches instead of the fused transposed conv
  int no_tail_fusion;           // ISI_NO_TAIL_FUSION: the decoder's last two transposed convolutions as two full layers
  int no_gemm_kernel;           // ISI_NO_GEMM_KERNEL: the 1x1 implicit-GEMM convolution kernel for the linear layers
  int gemm_narrow_below;        // ISI_GEMM_NARROW_BELOW: 128 x 64 tiles when 128 x 128 ones would number fewer than this (0: the CU count)
  int gemm_no_wide;             // ISI_GEMM_NO_WIDE: 128 x 128 tiles where the linear-layer GEMM would take 256 x 128 ones
  int no_wgrad_halo;            // ISI_NO_WGRAD_HALO: per-tap im2col weight-gradient kernel instead of the halo-staged one
  int conv_pair_bm;             // ISI_CONV_PAIR_BM: 128 / 256 forces the tile height of the LDS-DMA convolution (0: per shape)
  int conv_pair_all;            // ISI_CONV_PAIR_ALL: DMA kernel also for the shapes it is not preferred on
  int conv_tap_major;           // ISI_CONV_TAP_MAJOR: K order of the register-staged kernel (measurement)
  int respair_th, res_th, convt_th, convt_pair_th;   // forced tile heights (tests, measurement)
  int respair_one_wave_per_row; // ISI_RESPAIR_ONE_WAVE_PER_ROW: the 4-row residual-block tiles with four waves (one per row, until round 6) instead of eight
  int decode_nt;                // ISI_DECODE_NT: non-temporal weight loads in the batch-1 decode GEMVs (default 1)
  int attn_g_from_kv;           // ISI_ATTN_G_FROM_KV: (kept logits) the key-stationary backward kernel stores dS into G (default 1)
  int wgrad_split_target;       // ISI_WGRAD_SPLIT_TARGET: workgroups the split weight-gradient kernel aims at (0 = 768)
  int attn_full_zero;           // ISI_ATTN_FULL_ZERO: the attention backward zeroes all of G, not only the margins of its band
  int attn_old_fwd;             // ISI_ATTN_OLD_FWD: the round-3 forward kernel (32-key tiles) for the 16-bit modes (A/B switch)
  int attn_no_fwd3;             // ISI_ATTN_NO_FWD3: the round-4 forward kernel (rel_attention_fwd2.hip) where the plane-staged one would run
  int attn_fwd3_all;            // ISI_ATTN_FWD3_ALL: the plane-staged forward kernel for the single-term modes too (default: three-term only)
  int prior_graph;              // ISI_PRIOR_GRAPH: positions per replayed hipGraph of the decode loop (8; 0 = direct launches)
  int decode_mfma_rows;         // ISI_DECODE_MFMA_ROWS: batched decoding runs a stage as a 32-row GEMM tile on the fp32 matrix pipe for MORE
  int decode_stats_global;      // ISI_DECODE_STATS_GLOBAL: the tile path's input-row LayerNorm statistics from a second load of the rows (until round 6) instead of from the staged tile
  int decode_attn_separate_splits;   // ISI_DECODE_ATTN_SEPARATE_SPLITS: two key splits of the cached attention as two workgroups + the combine launch (until round 6)
  int decode_no_stat_handoff;   // ISI_DECODE_NO_STAT_HANDOFF: every launch of the tile path computes the LayerNorm statistics of its residual rows itself
                                // rows than this (16); up to it, the GEMV kernels (batch 1's operation order)
  int cu_count;                 // ISI_CU_COUNT: compute units the persistent kernels size their grids for (0: the device's); for
                                // launches on a stream that owns a SUBSET of the chip (hipExtStreamCreateWithCUMask; measurements)
  int conv_ablate, vq_dbg, respair_abl;   // ISI_MEASURE builds only
};

Knobs &knobs();

}  // namespace isi
