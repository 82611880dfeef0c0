#!/usr/bin/env python3
"""Ablation of conv_pair_f16.hip on one layer (ISI_CONV_ABLATE: 1 no MFMAs, 2 no DMA inside the loop)."""
import os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 64
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, cin, cout, k, s, H, W in (("3x3 128->128", 128, 128, 3, 1, 32, 128),):
    x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
    xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
    pw = _ops.pack_conv_weight((torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev), with_f16=True)
    run = lambda: _ops.conv2d(xp, pw, None, cout, k, s, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
    out = []
    for ab, fl in ((0, 3), (1, 3), (2, 3), (3, 3), (16, 3), (17, 3), (8, 3)):
        with _hip.knob("ISI_CONV_ABLATE", ab), _hip.knob("ISI_CONV_FLUSH", fl):   # (-DISI_MEASURE builds honour the ablation)
            out.append(f"ablate {ab} flush {fl}: {timed(run):7.1f} us")
    print(name, " | ".join(out))
