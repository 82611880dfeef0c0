#!/bin/bash
# per-dispatch kernel trace of the last VQ-VAE forward of tools/bench_latency-like loop: bench.py forward only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_fwd1; rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import sys, pathlib
ROOT = pathlib.Path.cwd()
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch, bench
dev = torch.device("cuda:0")
model = bench._build_model(dev)[0]
x = torch.randn(64, 2, 128, 512, device=dev)
with torch.no_grad():
    for _ in range(30):
        model(x)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace -d $O/kt -o f -- python3 $O/run.py > $O/log.txt 2>&1
python tools/prof_summary.py $O/kt/f_results.db 40 > gpurun_out/kt_forward.txt 2>&1
tail -50 gpurun_out/kt_forward.txt; tail -3 $O/log.txt
