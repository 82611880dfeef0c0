#!/usr/bin/env python3
"""Phase timestamps of conv_pair_f16.hip (ISI_CONV_ABLATE=32): block 8, waves 0 (group A) and 4 (group B), chunks 8..15."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))   # -DISI_MEASURE build
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, cin, cout, k, s, H, W = 64, 128, 128, 3, 1, 32, 128
x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
pw = _ops.pack_conv_weight((torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev), with_f16=True)
run = lambda: _ops.conv2d(xp, pw, None, cout, k, s, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
for _ in range(3): run()
_hip.check(_hip.lib().isi_knob_set(b"ISI_CONV_ABLATE", 32), "isi_knob_set")
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
wall_us = a.elapsed_time(b) * 1e3
buf = (C.c_longlong * 256)()
L = _hip.lib()
assert L.isi_debug_conv_pair_stamps(buf, 256) == 0
print(f"instrumented launch: {wall_us:.1f} us wall; workgroup 8 lived {buf[127] - buf[126]} cycles -> {(buf[127] - buf[126]) / wall_us / 1e3:.2f} GHz if it spanned the launch")
base = buf[0]
print("cycles relative to group A chunk 8 start; columns: start | reads issued | DMAs issued | waits done | barrier passed | MFMAs done | 2nd barrier passed")
for grp in range(2):
    for c in range(8):
        r = [buf[(grp * 8 + c) * 8 + i] - base for i in range(7)]
        print("AB"[grp], "chunk", 8 + c, " ".join(f"{v:7d}" for v in r), "| mem", r[3] - r[0], "bar", r[4] - r[3], "mma", r[5] - r[4], "bar2", r[6] - r[5])

print("second tile of workgroup 8, cycles: item start | after entry barrier | tables + staging set up | prologue DMA landed + barrier | loop done | epilogue barrier | transposes written | stores issued")
for grp in range(2):
    r = [buf[130 + grp * 10 + i] for i in range(8)]
    print("AB"[grp], " ".join(f"{v - r[0]:7d}" for v in r), "| steps", [r[i + 1] - r[i] for i in range(7)])
