#!/usr/bin/env python3
"""Timeline of the key-stationary attention backward kernel (rel_attention_bwd_kv_split_kernel, kept logits, dS -> G):
workgroup 0 (causal: key block 0, the longest query range), waves 0 and 4; -DISI_MEASURE build
(`make -C interactive-spectrogram-inpainting_amd/csrc EXTRA=-DISI_MEASURE OUT=$PWD/interactive-spectrogram-inpainting_amd/lib_measure`)."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
dev = torch.device("cuda:0")
B, H, S, hd = 8, 8, 1025, 64
d = H * hd
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_ops.ATTENTION_PRECISION = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
torch.manual_seed(0)
qkv = torch.randn(S, B, 3 * d, device=dev, requires_grad=True)
rel = (torch.randn(H, 2 * S - 1, hd, device=dev) * 0.1).requires_grad_(True)
w = torch.randn(S, B, d, device=dev)
for _ in range(3):
    out = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, mode, None)
    (out * w).sum().backward()
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert _hip.lib().isi_debug_attention_bwd_stamps(buf, 512) == 0
for grp in range(2):
    r = [buf[grp * 256 + i] for i in range(256)]
    t0 = r[0]
    print(f"wave {4 * grp}: loop starts 0, loop ends +{r[1] - t0}")
    s = 0
    while 4 + 12 * s + 9 < 256 and r[4 + 12 * s] >= t0 and (s == 0 or r[4 + 12 * s] > r[4 + 12 * (s - 1)]):
        x = r[4 + 12 * s: 4 + 12 * s + 10]
        if x[2] > x[1]:
            mid = (f"P ready {x[2] - x[1]:5d} | dV {x[3] - x[2]:5d} | dP, dS {x[4] - x[3]:5d} | G stores {x[5] - x[4]:5d} | dK {x[6] - x[5]:5d}")
        else:
            mid = f"(tile skipped) {x[6] - x[1]:5d}"
        print(f"  step {s:2d} @{x[0] - t0:7d}: prefetch issue {x[1] - x[0]:5d} | {mid} | barrier {x[7] - x[6]:5d} | commit {x[8] - x[7]:5d} | barrier {x[9] - x[8]:5d} | step {x[9] - x[0]:6d}")
        s += 1

# ---- the query-stationary kernel (dS read back from G): workgroup 0 = (causal) the query block with the longest key range
assert _hip.lib().isi_debug_attention_bwd_stamps(buf, -512) == 0
for grp in range(2):
    r = [buf[grp * 256 + i] for i in range(256)]
    t0 = r[0]
    print(f"q kernel, wave {4 * grp}: loop starts 0, loop ends +{r[1] - t0}")
    s = 0
    while 4 + 8 * s + 7 < 256 and r[4 + 8 * s] >= t0 and (s == 0 or r[4 + 8 * s] > r[4 + 8 * (s - 1)]):
        x = r[4 + 8 * s: 4 + 8 * s + 8]
        mid = (f"dS ready {x[2] - x[1]:5d} | dQ {x[3] - x[2]:5d} | G part {x[4] - x[3]:5d}") if x[2] > x[1] else f"(tile skipped) {x[4] - x[1]:5d}"
        print(f"  step {s:2d} @{x[0] - t0:7d}: prefetch issue {x[1] - x[0]:5d} | {mid} | barrier {x[5] - x[4]:5d} | commit {x[6] - x[5]:5d} | barrier {x[7] - x[6]:5d} | step {x[7] - x[0]:6d}")
        s += 1
