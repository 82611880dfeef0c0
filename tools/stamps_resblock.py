#!/usr/bin/env python3
"""Phase timestamps of csrc/resblock_pair_f16.hip (needs the -DISI_MEASURE build, see tools/stamps_convT.py):
workgroup 8, waves 0 and 4, second work item; bottom-resolution block of the forward (B = 64, 32 x 128, C = 128)."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, Cc, R, H, W = 64, 128, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 128   # (16 64: the top resolution)
x = torch.relu(torch.randn(B, H, W, Cc, generator=g)).to(dev)
xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
p3 = _ops.pack_conv_weight((torch.randn(R, Cc, 3, 3, generator=g) * 0.03).to(dev), with_f16=True)
p1 = _ops.pack_conv_weight((torch.randn(Cc, R, 1, 1, generator=g) * 0.1).to(dev), with_f16=True)
b3 = torch.zeros(R, device=dev); b1 = torch.zeros(Cc, device=dev)
run = lambda: _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
print(f"instrumented launch: {a.elapsed_time(b) * 1e3:.1f} us wall")
buf = (C.c_longlong * 128)()
assert _hip.lib().isi_debug_resblock_pair_stamps(buf, 128) == 0
for grp in range(2):
    r = [buf[grp * 64 + i] for i in range(64)]
    print(f"wave {4 * grp}: entry barrier {r[1] - r[0]}, set-up {r[2] - r[1]}, prologue issue {r[3] - r[2]}, K loop {r[4] - r[3]}, "
          f"hidden -> fragments {r[5] - r[4]}, GEMM 2 + epilogue per output tile {r[8] - r[5]} {r[9] - r[8]} {r[10] - r[9]} {r[11] - r[10]}; "
          f"item {r[7] - r[0]} cycles")
