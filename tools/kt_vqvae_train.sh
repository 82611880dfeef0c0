#!/bin/bash
# per-kernel stats + the per-dispatch timeline of the last VQ-VAE training step (tools/bench_train.py, B = 64)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_vtrain; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o f -- python3 tools/bench_train.py > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/f_results.db 260 > gpurun_out/kt_vqvae_train.txt 2>&1
head -60 gpurun_out/kt_vqvae_train.txt; tail -3 $O/log.txt
