#!/bin/bash
# steady-state per-kernel breakdown of the top prior's training step -> gpurun_out/kt_prior_train_steady.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_pts; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o pt -- python3 tools/bench_prior_train.py --batch 8 --steps 4 > $O/log.txt 2>&1
(echo "# rocprofv3 --kernel-trace -- python3 tools/bench_prior_train.py --batch 8 --steps 4   ($(grep 'prior training step' $O/log.txt))"; python tools/prof_steady.py $O/kt/pt_results.db multi_tensor_apply 2 gemm_split wgrad reduce_partials) > gpurun_out/kt_prior_train_steady.txt 2>&1
rm -rf $O
head -140 gpurun_out/kt_prior_train_steady.txt
