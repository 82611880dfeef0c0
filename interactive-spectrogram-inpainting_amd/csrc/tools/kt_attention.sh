#!/bin/bash
# kernel trace of tools/bench_attention.py (forward + backward of the attention op, causal, bf16x3) -> gpurun_out/kt_attn_bwd.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_attn; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o a -- python3 tools/bench_attention.py --modes ${1:-1} --precisions ${2:-bf16x3} > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/a_results.db 14 > gpurun_out/kt_attn_bwd.txt 2>&1
rm -rf $O
cat gpurun_out/kt_attn_bwd.txt
