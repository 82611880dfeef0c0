#!/usr/bin/env python3
"""Fused residual block at the forward's two resolutions (B = 64, C = 128, R = 32): fp32 vs pair-format tensors."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, C, R = 64, 128, 32
w3 = torch.randn(R, C, 3, 3, generator=g) * 0.03
b3 = (torch.randn(R, generator=g) * 0.1).to(dev)
w1 = torch.randn(C, R, 1, 1, generator=g) * 0.1
b1 = (torch.randn(C, generator=g) * 0.1).to(dev)
p3, p1 = _ops.pack_conv_weight(w3.to(dev), with_f16=True), _ops.pack_conv_weight(w1.to(dev), with_f16=True)
for H, W in ((32, 128), (16, 64)):
    x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev)
    xd = x.permute(0, 3, 1, 2)
    xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
    flops = 2.0 * B * H * W * R * 10 * C
    res = {}
    for name, fn in (("fp32 in/out", lambda: _ops.resblock(xd, p3, b3, p1, b1, R, True, bf16x3=4)),
                     ("pair in/out", lambda: _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)),
                     ("pair in / fp32 out", lambda: _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0)),
                     ("six-term bf16 (fp32 in/out)", lambda: _ops.resblock(xd, p3, b3, p1, b1, R, True, bf16x3=2))):
        t = min(timed(fn) for _ in range(3))
        res[name] = t
    print(f"{H}x{W}: " + " | ".join(f"{k} {v:6.1f} us ({flops / v / 1e6:5.1f} TF)" for k, v in res.items()))
    import os
    for ab in ("1", "2", "3", "4", "8", "12"):
        with _hip.knob("ISI_RESPAIR_ABL", int(ab)):   # (-DISI_MEASURE builds only)
            t = min(timed(lambda: _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)) for _ in range(2))
        print(f"   pair kernel, ablation {ab} (1: one stage of 8, 2: no second GEMM / epilogue, 4: no skip re-read, 8: no stores; sums combine): {t:6.1f} us")
