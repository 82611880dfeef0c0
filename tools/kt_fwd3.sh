cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_f3; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o a -- python3 tools/check_attention_fwd3.py --no-sweep --modes 1 --precs bf16x3 bf16 > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/a_results.db 8 > gpurun_out/kt_f3.txt 2>&1
rm -rf $O/kt
head -9 gpurun_out/kt_f3.txt | cut -c1-120
