#!/usr/bin/env python3
"""Timeline of workgroup 0's first block in rel_attn_fwd3_kernel (csrc/rel_attention_fwd3.hip; -DISI_MEASURE build:
`make -C interactive-spectrogram-inpainting_amd/csrc EXTRA=-DISI_MEASURE OUT=$PWD/interactive-spectrogram-inpainting_amd/lib_measure`),
waves 0 (key group 0) and 4 (key group 1); B8 H8 S1025 hd64.  Stamps per step: 0 matrix segment starts, 4 P V issued, 5 K Q^T + band issued, 1 its end (at the
barrier), 2 barrier released = vector segment starts, 6 skew read back, 7 P packed, 3 its end (tiles landed, at the barrier)."""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
os.environ.setdefault("ISI_HIP_LIBRARY", str(ROOT / "interactive-spectrogram-inpainting_amd" / "lib_measure" / "libisi_hip.so"))
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
_hip.check(_hip.lib().isi_knob_set(b"ISI_ATTN_FWD3_ALL", 1), "isi_knob_set")
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
L = _hip.lib()
L.isi_debug_attention_fwd3_stamps.restype = C.c_int
buf = (C.c_longlong * 512)()
assert L.isi_debug_attention_fwd3_stamps(buf, 512) == 0
for grp in range(2):
    r = [buf[grp * 256 + i] for i in range(256)]
    t0 = r[0]
    print(f"wave {4 * grp}: block start 0, prologue done (Q split, first tile landed, barrier) +{r[1] - t0}")
    s = 0
    while 8 + 8 * s + 7 < 240 and r[8 + 8 * s] > t0 and (s == 0 or r[8 + 8 * s] > r[8 * s]):
        x = r[8 + 8 * s: 16 + 8 * s]
        nxt = r[16 + 8 * s] if r[16 + 8 * s] > x[3] else r[240]
        print(f"  step {s:2d} @{x[0] - t0:6d}: matrix {x[1] - x[0]:5d} (reads + band {x[4] - x[0]:5d}, K Q^T {x[5] - x[4]:5d}, P V {x[1] - x[5]:4d}) | wait {x[2] - x[1]:5d} |"
              f" vector {x[3] - x[2]:5d} (requests + skew {x[6] - x[2]:5d}, softmax + P {x[7] - x[6]:5d}, wait for tiles {x[3] - x[7]:5d}) | wait {nxt - x[3]:5d} | step {nxt - x[0]:5d}")
        s += 1
    print(f"  loop end +{r[240] - t0}, merged +{r[241] - t0}, stored +{r[242] - t0}")
