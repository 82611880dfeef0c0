import sys, pathlib, os
sys.path.insert(0, "interactive-spectrogram-inpainting_amd")
import torch
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
_hip.check(_hip.lib().isi_knob_set(b"ISI_ATTN_FWD3_ALL", 1), "isi_knob_set")
def run(q, k, v, rel, H, Ek, mode, prec, dense=None):
    _ops.ATTENTION_PRECISION = prec
    return _ops.rel_attention(q, k, v, rel, H, 1, 1, Ek, mask_mode=mode, dense_mask=dense)
reps = int(os.environ.get("REPS", "12"))
ALL = ((64, 8, 8, 1025, 1, "bf16x3"), (64, 8, 8, 1025, 1, "bf16"), (64, 8, 8, 1025, 0, "bf16x3"), (32, 16, 8, 1025, 1, "bf16x3"), (64, 8, 8, 1025, 1, "norel"))
for hd, H, B, S, mode, prec in ALL[:int(os.environ.get("NCASE", "5"))]:
    d = hd * H
    q, k, v = (torch.randn(S, B, d, device=dev) for _ in range(3))
    rel = torch.randn(H, 2 * S - 1, hd, device=dev) * 0.5
    if prec == "norel":
        rel, prec = None, "bf16x3"
    with _hip.knob("ISI_ATTN_NO_FWD3", 1):
        ref = run(q, k, v, rel, H, S, mode, prec)
    nbad = 0
    for rep in range(reps):
        got = run(q, k, v, rel, H, S, mode, prec)
        err = (got - ref).abs().reshape(S, B, H, hd).amax(3)      # [S,B,H]
        bad = err > 1e-4
        if bad.any():
            nbad += 1
            rows = bad.any(2).any(1).nonzero().flatten().tolist()
            bh = bad.any(0).nonzero().tolist()
            print(f"  rep {rep}: max err {err.max().item():.2e} rows {rows[:8]}..{rows[-1]} n={len(rows)} (b,h) {bh[:6]}")
    print(f"hd{hd} H{H} B{B} S{S} mode{mode} {prec} rel={'y' if rel is not None else 'n'}: {nbad}/{reps} runs differ from the round-4 kernel", flush=True)
