#!/bin/bash
# kernel trace of the top prior's eval forward (B = 8): tools/bench_prior_eval_graph.py -> gpurun_out/kt_prior_eval.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_pe; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o pe -- python3 tools/bench_prior_eval_graph.py > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/pe_results.db 0 > gpurun_out/kt_prior_eval.txt 2>&1
rm -rf $O
head -40 gpurun_out/kt_prior_eval.txt
