#!/bin/bash
# SQ counters of the attention kernels (two rocprofv3 --kernel-trace --pmc passes of tools/bench_attention.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_att; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 tools/bench_attention.py --modes 1 --precisions bf16x3 bf16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p2 -- python3 tools/bench_attention.py --modes 1 --precisions bf16x3 bf16 > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O -name "*counter_collection.csv") > gpurun_out/pmc_attention.txt 2>&1
rm -rf $O
grep -E "^rel_att|^attn_pack" -A1 gpurun_out/pmc_attention.txt
