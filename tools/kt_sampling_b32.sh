cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_b32; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o f -- python3 tools/prof_sampling.py 32 > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/f_results.db 14 > gpurun_out/kt_b32.txt 2>&1
