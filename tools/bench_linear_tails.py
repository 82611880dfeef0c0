#!/usr/bin/env python3
"""The feed-forward GEMMs of the prior with the element-wise tails a training step fuses into their epilogues
(M = 8200): FFN1 forward (N 2048, K 512) plain / + ReLU / + ReLU + dropout; FFN2 input gradient (N 2048, K 512, bf16x3)
plain / + gate / + gate and dropout scale."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch
from interactive_spectrogram_inpainting.priors import _ops

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device("cuda:0")
M, N, K = 8200, 2048, 512
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
pw = _ops.pack_linear_weight(w, range_check="now")
gate = torch.randn(M, N, device=dev)
for prec in ("f16x3", "bf16x3"):
    t0 = timed(lambda: _ops.linear(x, pw, b, N, precision=prec))
    t1 = timed(lambda: _ops.linear(x, pw, b, N, relu=True, precision=prec))
    t2 = timed(lambda: _ops.linear(x, pw, b, N, relu=True, precision=prec, dropout_p=0.1, dropout_seed_=1234567))
    t3 = timed(lambda: _ops.linear(x, pw, None, N, precision=prec, gate=gate, gate_scale=1.0 / 0.9))
    print(f"{prec}: plain {t0:.1f} us, relu {t1:.1f}, relu + dropout {t2:.1f}, gate (+ scale) {t3:.1f}")
