#!/usr/bin/env python3
"""One big layer of the forward (3x3 128->128 at B = 64 by default), N launches, for rocprofv3 passes:
    prof_conv.py [old|dma] [case] [n]      case: c3 (3x3 128->128 @32x128) | k4 (k4s2 64->128) | ct (convT 128->64)"""
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
which = sys.argv[1] if len(sys.argv) > 1 else "dma"
case = sys.argv[2] if len(sys.argv) > 2 else "c3"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if which == "old":
    os.environ["ISI_NO_CONV_PAIR_KERNEL"] = "1"
import torch  # noqa: E402
from interactive_spectrogram_inpainting.vqvae import _ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 64
cin, cout, k, s, H, W, tr = {"c3": (128, 128, 3, 1, 32, 128, False), "k4": (64, 128, 4, 2, 64, 256, False),
                             "ct": (128, 64, 4, 2, 32, 128, True), "n64": (128, 64, 4, 2, 32, 128, False)}[case]
x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
flags = _ops.PAIR_IN0 | (0 if which == "old" else _ops.PAIR_OUT)
if tr:
    pw = _ops.pack_convT_weight((torch.randn(cin, cout, 4, 4, generator=g) * 0.05).to(dev), with_f16=True)
    run = lambda: _ops.conv_transpose2d_k4s2(xp, pw, None, cout, relu=True, bf16x3=4, extra_flags=flags)
else:
    pw = _ops.pack_conv_weight((torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev), with_f16=True)
    run = lambda: _ops.conv2d(xp, pw, None, cout, k, s, 1 if k > 1 else 0, relu=True, bf16x3=4, extra_flags=flags)
for _ in range(n):
    run()
torch.cuda.synchronize()
print("done", which, case, n)
