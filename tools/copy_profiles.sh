#!/bin/bash
# gpurun_out/round/* (tools/round_profiles.sh on the GPU box) -> profiles/rNN_* (tracked)
R=${1:-r01}; O=gpurun_out/round
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
hdr() { { echo "# $1"; cat "$2"; } > "$3"; }
hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-prior --no-train   (VQ-VAE forward, B=64, default split_f16 mode = template tail 3; includes warm-up and the alt-precision passes)" $O/fwd_summary.txt profiles/${R}_kernel_trace_stats.txt
hdr "rocprofv3 --kernel-trace -- python3 tools/bench_train.py   (VQ-VAE training step, $(grep 'ms/step' $O/vqvae_train.txt | tail -1))" $O/vqvae_train_summary.txt profiles/${R}_vqvae_train_kernel_trace.txt
hdr "rocprofv3 --kernel-trace -- python3 tools/bench_prior_train.py --batch 8 --steps 2   ($(grep 'ms/step' $O/prior_train.txt | tail -1))" $O/prior_train_summary.txt profiles/${R}_prior_train_kernel_trace.txt
hdr "rocprofv3 --kernel-trace -- python3 tools/bench_prior.py   (KV-cached sampling, 4 x 1024 tokens + one full forward)" $O/prior_sampling_summary.txt profiles/${R}_prior_sampling_kernel_trace.txt
hdr "rocprofv3 --kernel-trace -- python3 tools/bench_frontend.py   (audio <-> mel/IF spectrogram, B=64)" $O/frontend_summary.txt profiles/${R}_frontend_kernel_trace.txt
cp $O/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
{ echo "# tools/bench_attention.py (B8 H8 S1025 hd64, fp32 kernels vs split-bf16 kernels; TF = dense-count convention)"; grep -v amdgpu.ids $O/attention.txt; } > profiles/${R}_attention.txt
{ echo "# SQ counters of the VQ-VAE forward's kernels, B=64 (two rocprofv3 --kernel-trace --pmc passes of"
  echo "# \`python3 bench.py --no-cpu-baseline --no-prior --no-train --steps 3\`, which also times the other precision modes; per-dispatch"
  echo "# means; tools/round_profiles.sh, tools/pmc_sq_summary.py).  Template tail of conv_igemm / resblock: 0 exact fp32,"
  echo "# 1 bf16x3, 2 bf16x6, 3 f16x3 (the default mode's kernels)."
  echo
  cat $O/pmc_sq_forward.txt; } > profiles/${R}_pmc_sq_forward.txt
{ echo "# SQ counters of the attention kernels, B8 H8 S1025 hd64 causal, three-term (bf16x3) and single-term (bf16) products: two rocprofv3"
  echo "# --kernel-trace --pmc passes of tools/bench_attention.py --modes 1 --precisions bf16x3 bf16 (tools/round_profiles.sh); per-dispatch"
  echo "# means.  rel_attn_fwd3_kernel<HD, terms, f16> + attn_pack_kernel: forward, three-term products (round 6); rel_attn_fwd2_kernel<HD, terms,"
  echo "# f16, unit>: forward, single-term products; *_split_kernel<HD, single-term, kept logits>: backward."
  echo
  cat $O/pmc_attention.txt; } > profiles/${R}_pmc_attention.txt
hdr "rocprofv3 --kernel-trace -- python3 tools/bench_attention.py --modes 1 0 --precisions bf16x3   (attention op forward + backward, causal and unmasked)" $O/attention_kernel_trace.txt profiles/${R}_attention_kernel_trace.txt
{ echo "# tools/bench_linear.py, tools/bench_linear_wgrad.py (M = 8200 rows; us and TFLOP/s-equivalent per product mode)"; grep -v amdgpu.ids $O/linear.txt; } > profiles/${R}_linear_layers.txt
{ echo "# tools/bench_prior_train.py --batch 8 --steps 4 (eager) and --steps 10 --graph (the step replayed from a HIP graph)"; grep -v "amdgpu.ids\|UserWarning\|detach\|tokens/s  loss" $O/prior_train.txt; grep "ms" $O/prior_train.txt | grep -v Warn; echo "--- graph replay"; grep "ms" $O/prior_train_graph.txt | grep -v Warn; } > profiles/${R}_prior_train_step.txt
[ -f $O/prior_sampling_b32_summary.txt ] && hdr "rocprofv3 --kernel-trace -- python3 tools/prof_sampling.py 32   (batched decoding, B = 32)" $O/prior_sampling_b32_summary.txt profiles/${R}_prior_sampling_b32_kernel_trace.txt
[ -f $O/stamps.txt ] && { echo "# in-kernel cycle stamps (-DISI_MEASURE library): GEMM 256x256 and 128x64 tiles, attention backward (key-stationary kernel), attention forward (fwd2: register-staged; fwd3: plane-staged)"; grep -v amdgpu.ids $O/stamps.txt; } > profiles/${R}_cycle_stamps.txt
