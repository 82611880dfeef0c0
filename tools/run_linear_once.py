#!/usr/bin/env python3
"""One linear layer's GEMM a few times (for rocprofv3 passes): run_linear_once.py N K [precision] [reps]   (M = 8200)"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
N, K = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
M = 8200
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
pw = _ops.pack_linear_weight(w, range_check="now")
for _ in range(reps):
    _ops.linear(x, pw, b, N, precision=prec)
torch.cuda.synchronize()
