#!/bin/bash
# durations of the cached-attention launches of batched decoding (B = 32), self (even calls) vs cross (odd), by position quartile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_da; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o f -- python3 tools/prof_sampling.py ${1:-32} > $O/log.txt 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/kt_da/kt/f_results.db")
rows = [(e - s) / 1e3 for s, e in db.execute("select start, end from kernels where name like '%rel_attention_decode_f32_kernel%' order by start")]
n = len(rows) // 16        # positions
for which, name in ((0, "self"), (1, "cross")):
    d = rows[which::2]
    per_pos = [sum(d[p * 8:(p + 1) * 8]) / 8 for p in range(len(d) // 8)]
    q = len(per_pos) // 4
    print(name, "avg us per launch by position quartile:", [round(sum(per_pos[i * q:(i + 1) * q]) / q, 2) for i in range(4)])
PY
rm -rf $O
