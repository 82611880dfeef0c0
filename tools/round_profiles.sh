#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/round/ (copied to profiles/ afterwards):
# bench line, kernel traces (VQ-VAE forward, VQ-VAE training, prior training, prior sampling, front-end) and
# the HBM-traffic PMC passes of the forward.  Counters are collected in their own runs (kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt_fwd -o fwd -- python3 bench.py --no-cpu-baseline --no-prior --no-train > $O/bench_under_rocprof.json 2>/dev/null
python tools/prof_summary.py $O/kt_fwd/fwd_results.db 0 > $O/fwd_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_vt -o vt -- python3 tools/bench_train.py > $O/vt.log 2>&1
python tools/prof_summary.py $O/kt_vt/vt_results.db 0 > $O/vqvae_train_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_pt -o pt -- python3 tools/bench_prior_train.py --batch 8 --steps 2 > $O/pt.log 2>&1
python tools/prof_summary.py $O/kt_pt/pt_results.db 0 > $O/prior_train_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_ps -o ps -- python3 tools/bench_prior.py > $O/ps.log 2>&1
python tools/prof_summary.py $O/kt_ps/ps_results.db 0 > $O/prior_sampling_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_ps32 -o ps -- python3 tools/prof_sampling.py 32 > $O/ps32.log 2>&1
python tools/prof_summary.py $O/kt_ps32/ps_results.db 0 > $O/prior_sampling_b32_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_fe -o fe -- python3 tools/bench_frontend.py > $O/fe.log 2>&1
python tools/prof_summary.py $O/kt_fe/fe_results.db 0 > $O/frontend_summary.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-prior --no-train --steps 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --no-prior --no-train --steps 5 > /dev/null 2>&1
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm_traffic.json > $O/pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq1 -- python3 bench.py --no-cpu-baseline --no-prior --no-train --steps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sq2 -- python3 bench.py --no-cpu-baseline --no-prior --no-train --steps 3 > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc_sq1 $O/pmc_sq2 -name "*counter_collection.csv") > $O/pmc_sq_forward.txt 2>&1
# SQ counters of the attention kernels (forward 64-key-tile kernels, backward with kept logits): two passes
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_at1 -- python3 tools/bench_attention.py --modes 1 --precisions bf16x3 bf16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_at2 -- python3 tools/bench_attention.py --modes 1 --precisions bf16x3 bf16 > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc_at1 $O/pmc_at2 -name "*counter_collection.csv") > $O/pmc_attention.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_at -o at -- python3 tools/bench_attention.py --modes 1 0 --precisions bf16x3 > /dev/null 2>&1
python tools/prof_summary.py $O/kt_at/at_results.db 0 > $O/attention_kernel_trace.txt 2>&1
python tools/bench_attention.py --precisions f32 bf16x3 bf16 f16 > $O/attention.txt 2>&1
python tools/bench_prior_train.py --batch 8 --steps 4 > $O/prior_train.txt 2>&1
python tools/bench_train.py > $O/vqvae_train.txt 2>&1
python tools/bench_prior.py > $O/prior_sampling.txt 2>&1
python tools/bench_frontend.py > $O/frontend.txt 2>&1
# round 4: the linear layers' GEMM / weight-gradient kernels, the training step replayed from a HIP graph, cycle stamps
(python tools/bench_linear.py; python tools/bench_linear_wgrad.py) > $O/linear.txt 2>&1
python tools/bench_prior_train.py --batch 8 --steps 10 --graph > $O/prior_train_graph.txt 2>&1
if [ -f interactive-spectrogram-inpainting_amd/lib_measure/libisi_hip.so ]; then
  (echo "## tools/stamps_gemm.py 2048 512 f16x3"; python tools/stamps_gemm.py 2048 512 f16x3; echo "## tools/stamps_gemm.py 512 512 f16x3"; python tools/stamps_gemm.py 512 512 f16x3
   echo "## tools/stamps_attention_bwd.py 1"; python tools/stamps_attention_bwd.py 1; echo "## tools/stamps_fwd2.py bf16 1"; python tools/stamps_fwd2.py bf16 1
   echo "## tools/stamps_fwd2.py bf16x3 1"; python tools/stamps_fwd2.py bf16x3 1
   echo "## tools/stamps_fwd3.py bf16x3 1"; python tools/stamps_fwd3.py bf16x3 1; echo "## tools/stamps_fwd3.py bf16 1"; python tools/stamps_fwd3.py bf16 1) > $O/stamps.txt 2>&1
fi
rm -rf $O/kt_* $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_at1 $O/pmc_at2     # the sqlite / csv dumps are large; the summaries are what is kept
tail -c 600 $O/bench.json
