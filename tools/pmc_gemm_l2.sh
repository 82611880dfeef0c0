#!/bin/bash
# L2 / fabric counters of one linear-layer GEMM: pmc_gemm_l2.sh N K
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_gl2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/p1 -- python3 tools/run_linear_once.py $1 $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 tools/run_linear_once.py $1 $2 > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O -name "*counter_collection.csv") 2>&1 | grep -A12 "^gemm_split"
rm -rf $O
