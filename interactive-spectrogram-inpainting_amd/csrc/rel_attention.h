// Kernel-side arguments of the relative-attention forward kernels (rel_attention_f32.hip: exact-fp32 and the round-2/3
// split kernels; rel_attention_fwd2.hip: the 64-key-tile kernels) and the launcher the dispatcher hands over to.
#pragma once
#include <hip/hip_runtime.h>

namespace isi {

struct AttnKArgs {
  const float *q, *k, *v, *e, *mask;
  float *out, *lse;
  unsigned q_bytes, k_bytes, v_bytes, e_bytes;
  int Sq, Sk, H, B;
  int nblk;       // query blocks the split kernels run (all, or only the full ones: the tail rows go to the one-row kernel)
  int q_ss, q_sb, q_sh, k_ss, k_sb, k_sh, v_ss, v_sb, v_sh, o_ss, o_sb, o_sh;  // element strides
  int Cq, Ck, Ek, R;
  int mask_mode;  // 0 none, 1 causal (j <= i), 2 anti-causal (j >= i)
  float scale;
  int split;      // 1: three-term split-bf16 products (rel_attention_split_kernel), 2: single-term bf16
  float *logits;  // optional [B,H,Sq,ldl]: base-2 logits of the allowed pairs, kept for the backward (rel_attention_fwd2.hip)
  int ldl;
};

// rel_attention_fwd2.hip.  precision: 1 three-term split-bf16, 2 single-term bf16, 3 single-term f16.  Returns
// ISI_E_UNSUPPORTED (without touching the last-error text) for a shape it does not take.
bool rel_attention_fwd2_ok(const AttnKArgs &a, int head_dim);
int rel_attention_fwd2(const AttnKArgs &a, int head_dim, int precision, hipStream_t stream);
int rel_attention_fwd2_debug_stamps(long long *host, int n);

// rel_attention_fwd3.hip: one channel per event (Cq = Ck = 1), head_dim 32 / 64, precision 1 .. 3; K, V and e are split into
// 16-bit planes in `workspace` (rel_attention_fwd3_workspace_bytes, 256-byte aligned) by a pack launch in front of the kernel.
bool rel_attention_fwd3_ok(const AttnKArgs &a, int head_dim, int precision);
size_t rel_attention_fwd3_workspace_bytes(const AttnKArgs &a, int head_dim, int precision);
int rel_attention_fwd3(const AttnKArgs &a, int head_dim, int precision, void *workspace, size_t workspace_bytes, hipStream_t stream);
int rel_attention_fwd3_debug_stamps(long long *host, int n);

}  // namespace isi
