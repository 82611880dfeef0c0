#!/usr/bin/env python3
"""Timeline of workgroup 0's first block in rel_attn_fwd2_kernel (csrc/rel_attention_fwd2.hip; -DISI_MEASURE build:
`make -C interactive-spectrogram-inpainting_amd/csrc EXTRA=-DISI_MEASURE OUT=$PWD/interactive-spectrogram-inpainting_amd/lib_measure`),
waves 0 (key group 0) and 4 (key group 1); B8 H8 S1025 hd64."""
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
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_ops.ATTENTION_PRECISION = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for _ in range(3):
    _ops.rel_attention(q, k, v, rel, H, 1, 1, S, mask_mode=mode)
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert _hip.lib().isi_debug_attention_fwd2_stamps(buf, 512) == 0
for grp in range(2):
    r = [buf[grp * 256 + i] for i in range(256)]
    t0 = r[0]
    print(f"wave {4 * grp}: block start 0, loads landed +{r[1] - t0}, committed +{r[2] - t0}, barrier +{r[3] - t0}")
    s = 0
    while 4 + 12 * s + 11 < 240 and r[4 + 12 * s] > t0 and (s == 0 or r[4 + 12 * s] > r[4 + 12 * (s - 1)]):
        b = 4 + 12 * s
        x = r[b:b + 12]
        def dd(i, j):
            return x[i] - x[j] if x[i] > 0 and x[j] > 0 and x[i] >= x[j] else -1
        print(f"  step {s} @{x[0] - t0}: prefetch issue {dd(1, 0)} | sb0: mfma {dd(2, 1)} skew {dd(3, 2)} softmax {dd(4, 3)} PV {dd(5, 4)}"
              f" | sb1: mfma {dd(6, 5)} skew {dd(7, 6)} softmax {dd(8, 7)} PV {dd(9, 8)} | barrier {x[10] - max(x[1:10])} commit+barrier {dd(11, 10)}"
              f" | step {x[11] - x[0]}")
        s += 1
    print(f"  merge start +{r[240] - t0}, merged +{r[241] - t0}, stored +{r[242] - t0}")
