#!/bin/bash
# kernel trace of the top prior's training step (tools/bench_prior_train.py --batch 8 --steps 2) -> gpurun_out/kt_prior_train.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_pt; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o pt -- python3 tools/bench_prior_train.py --batch 8 --steps 2 > $O/log.txt 2>&1
(echo "# rocprofv3 --kernel-trace -- python3 tools/bench_prior_train.py --batch 8 --steps 2   ($(grep 'prior training step' $O/log.txt))"; python tools/prof_summary.py $O/kt/pt_results.db 0) > gpurun_out/kt_prior_train.txt 2>&1
rm -rf $O
head -45 gpurun_out/kt_prior_train.txt
