"""Optimizer construction of the training scripts (train_vqvae.py:777, train_autoregressive_model.py:624-640:
`optim.Adam(model.parameters(), lr, eps)`).  Same update rule; on the GPU torch's single-launch ("fused")
implementation is asked for -- the default multi-tensor path issues ~40 launches of 40 us each for the prior's
19 M parameters (1.5 ms of a 50 ms step), the fused one a handful."""
from __future__ import annotations

from typing import Iterable

import torch


def make_adam(params: Iterable[torch.nn.Parameter], lr: float, eps: float = 1e-8, **kw) -> torch.optim.Adam:
    params = list(params)
    on_gpu = bool(params) and all(p.is_cuda for p in params)
    if on_gpu and "fused" not in kw and "foreach" not in kw:
        try:
            return torch.optim.Adam(params, lr=lr, eps=eps, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):     # a torch build without the fused kernels
            pass
    return torch.optim.Adam(params, lr=lr, eps=eps, **kw)
