#!/usr/bin/env python3
"""Phase timestamps of csrc/convT_pair_f16.hip (needs the -DISI_MEASURE build:
`make -C interactive-spectrogram-inpainting_amd/csrc EXTRA=-DISI_MEASURE OUT=$PWD/interactive-spectrogram-inpainting_amd/lib_measure`,
run with ISI_HIP_LIBRARY=.../lib_measure/libisi_hip.so): workgroup 8, waves 0 and 4, second work item."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, cin, cout, H, W = 64, 128, 64, 32, 128
x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
pw = _ops.pack_convT_weight((torch.randn(cin, cout, 4, 4, generator=g) * 0.05).to(dev), with_f16=True)
run = lambda: _ops.conv_transpose2d_k4s2(xp, pw, None, cout, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
print(f"instrumented launch: {a.elapsed_time(b) * 1e3:.1f} us wall")
buf = (C.c_longlong * 128)()
assert _hip.lib().isi_debug_convT_pair_stamps(buf, 128) == 0
ns = cin // 16
for grp in range(2):
    r = [buf[grp * 64 + i] for i in range(64)]
    t0 = r[0]
    print(f"wave {4 * grp}: entry barrier {r[1] - r[0]}, set-up {r[2] - r[1]}, prologue issue {r[3] - r[2]}, "
          f"K loop {r[4] - r[3]}, epilogue {r[5] - r[4]}; item {r[5] - r[0]} cycles")
    prev = r[3]
    for c in range(ns):
        top, waited, passed = r[8 + 3 * c], r[9 + 3 * c], r[10 + 3 * c]
        nxt = r[8 + 3 * (c + 1)] if c + 1 < ns else r[4]
        print(f"   stage {c}: vmcnt wait {waited - top:6d}  barrier {passed - waited:6d}  steps {nxt - passed:6d}")
