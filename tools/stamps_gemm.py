#!/usr/bin/env python3
"""Timeline of workgroup 8 (waves 0 and 1) of gemm_split_kernel (csrc/gemm_split_f32.hip; -DISI_MEASURE build:
`make -C interactive-spectrogram-inpainting_amd/csrc EXTRA=-DISI_MEASURE OUT=$PWD/interactive-spectrogram-inpainting_amd/lib_measure`).
usage: stamps_gemm.py [N K [precision]]   (M = 8200)"""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
M = 8200
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
pw = _ops.pack_linear_weight(w, range_check="now")
for _ in range(5):
    _ops.linear(x, pw, b, N, precision=prec)
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert _hip.lib().isi_debug_gemm_stamps(buf, 512) == 0
for wv in range(2):
    r = [buf[wv * 256 + i] for i in range(256)]
    t0 = r[0]
    print(f"N={N} K={K} {prec} wave {wv}: prologue +{r[1] - t0}; K loop ends +{r[2] - t0}; epilogue ends +{r[3] - t0}")
    for c in range(K // 32):
        x4 = r[4 + 4 * c: 8 + 4 * c]
        nxt = r[4 + 4 * (c + 1)] if c + 1 < K // 32 else r[2]
        print(f"  chunk {c:2d} @{x4[0] - t0:6d}: loads issued {x4[1] - x4[0]:5d} | mfma {x4[2] - x4[1]:5d} | convert+store {x4[3] - x4[2]:5d} | barrier {nxt - x4[3]:5d} | chunk {nxt - x4[0]:5d}")
