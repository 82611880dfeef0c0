#!/usr/bin/env python3
"""Attention backward with the forward's logits kept (ISI_ATTN_SAVE_LOGITS, default) against the recomputing backward:
times at B8 H8 S1025 hd64 per mask mode, and the gradients of both paths against each other."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting.priors import _ops  # noqa: E402
from interactive_spectrogram_inpainting.priors._train import RelAttentionFn  # noqa: E402
from bench_attention import timed  # noqa: E402

dev = torch.device("cuda:0")
B, H, S, hd = 8, 8, 1025, 64
d = H * hd
torch.manual_seed(0)
qkv = torch.randn(S, B, 3 * d, device=dev, requires_grad=True)
rel = (torch.randn(H, 2 * S - 1, hd, device=dev) * 0.1).requires_grad_(True)
w = torch.randn(S, B, d, device=dev)
for mode in (1, 0, 2):
    for prec in ("bf16x3", "bf16"):
        _ops.ATTENTION_PRECISION = prec
        grads = {}
        for keep in (True, False):
            _ops.SAVE_ATTENTION_LOGITS = keep
            t_f = timed(lambda: RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, mode, None))
            out = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, mode, None)

            def bwd():
                qkv.grad = None
                rel.grad = None
                out.backward(w, retain_graph=True)
            t_b = timed(bwd)
            grads[keep] = (qkv.grad.clone(), rel.grad.clone())
            print(f"mode {mode} {prec:7s} logits {'kept      ' if keep else 'recomputed'}: fwd (training) {t_f:7.1f} us  bwd {t_b:7.1f} us", flush=True)
        dq = (grads[True][0] - grads[False][0]).abs().max().item() / grads[False][0].abs().max().item()
        de = (grads[True][1] - grads[False][1]).abs().max().item() / grads[False][1].abs().max().item()
        print(f"   kept vs recomputed gradients: d qkv {dq:.2e}  d rel {de:.2e} (of the maximum)")
