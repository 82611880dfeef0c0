#!/bin/bash
# SQ counters of the training-path kernels built in round 3 (halo weight gradient, GEMM, row-major linear weight
# gradient): two rocprofv3 --kernel-trace --pmc passes (counters in their own runs) of tools/bench_wgrad.py and
# tools/bench_linear.py, summarised into gpurun_out/pmc_train_kernels.txt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_tk; mkdir -p $O
for prog in bench_wgrad bench_linear fuzz_wgrad; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${prog}_1 -- python3 tools/$prog.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/${prog}_2 -- python3 tools/$prog.py > /dev/null 2>&1
done
python tools/pmc_sq_summary.py $(find $O -name "*counter_collection.csv") > gpurun_out/pmc_train_kernels.txt 2>&1
rm -rf $O
head -60 gpurun_out/pmc_train_kernels.txt
