#!/bin/bash
# Counters of the forward attention kernels (rocprofv3 --kernel-trace --pmc passes of tools/check_attention_fwd2.py).
# usage: tools/pmc_fwd2.sh <out.txt> [extra args of check_attention_fwd2.py]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/pmc_fwd2.txt}; shift
O=gpurun_out/pmc_fwd2_raw; mkdir -p $O
ARGS="--no-sweep --new-only $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 tools/check_attention_fwd2.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p2 -- python3 tools/check_attention_fwd2.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p3 -- python3 tools/check_attention_fwd2.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p4 -- python3 tools/check_attention_fwd2.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum --output-format csv -d $O/p5 -- python3 tools/check_attention_fwd2.py $ARGS > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O -name "*counter_collection.csv") > $OUT 2>&1
rm -rf $O
grep -E "rel_attn_fwd2|rel_attention_split" -A2 $OUT | cut -c1-400
