#!/usr/bin/env python3
"""Phase timestamps of the split-bf16 attention forward (csrc/rel_attention_f32.hip; -DISI_MEASURE build, see
tools/stamps_convT.py): heaviest workgroup of (h, b) = (0, 0), waves 0 (key tile of parity 0) and 4 (parity 1), fifth
key-pair iteration; B8 H8 S1025 hd64 causal."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
B, H, S, hd = 8, 8, 1025, 64
d = H * hd
torch.manual_seed(0)
q, k, v = (torch.randn(S, B, d, device=dev) for _ in range(3))
rel = torch.randn(H, 2 * S - 1, hd, device=dev) * 0.1
_ops.ATTENTION_PRECISION = "bf16x3"
for _ in range(3):
    _ops.rel_attention(q, k, v, rel, H, 1, 1, S, mask_mode=1)
torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
assert _hip.lib().isi_debug_attention_stamps(buf, 64) == 0
for grp in range(2):
    r = [buf[grp * 16 + i] for i in range(9)]
    print(f"wave {4 * grp}: prefetch issue {r[1] - r[0]}, QK^T {r[2] - r[1]}, rel term + skew {r[3] - r[2]}, softmax {r[4] - r[3]}, "
          f"P split + PV {r[5] - r[4]}, barrier {r[6] - r[5]}, commit {r[7] - r[6]}, barrier {r[8] - r[7]}; iteration {r[8] - r[0]} cycles")
