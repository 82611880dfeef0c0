#!/bin/bash
# SQ / LDS counter passes of one layer for both split-f16 convolution kernels: tools/prof_conv.sh [case]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CASE=${1:-c3}
O=gpurun_out/prof_conv_$CASE; rm -rf $O; mkdir -p $O
for W in ${2:-old dma}; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1_$W -- python3 tools/prof_conv.py $W $CASE 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/sq2_$W -- python3 tools/prof_conv.py $W $CASE 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/sq3_$W -- python3 tools/prof_conv.py $W $CASE 6 > /dev/null 2>&1
  echo "=== $W" >> $O/summary.txt
  python tools/pmc_sq_summary.py $(find $O/sq1_$W $O/sq2_$W $O/sq3_$W -name "*counter_collection.csv") 2>&1 | grep -A40 "conv_" | head -60 >> $O/summary.txt
done
rm -rf $O/sq*
cat $O/summary.txt
