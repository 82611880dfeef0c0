"""Mean squared error of the VQ-VAE training step on the HIP library (reference train_vqvae.py:168-176 builds
`nn.MSELoss()`; this module is its drop-in on the GPU).  Forward = two launches with a fixed summation order, backward =
one; torch's path is an element-wise kernel, a memset of the reduction's semaphores and two reduction launches (then two
element-wise kernels backward) -- and a memset NODE inside a recorded training step is not reliably ordered with its
neighbours when the step is replayed on ROCm 7.2, which is why the recorded VQ-VAE step uses this."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from ... import _hip


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        L = _hip.lib()
        n = a.numel()
        ws = torch.empty(L.isi_mse_loss_num_partials(n), dtype=torch.float32, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        _hip.check(L.isi_mse_loss_f32(a.data_ptr(), b.data_ptr(), n, ws.data_ptr(), out.data_ptr(), _s(a)), "isi_mse_loss_f32")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        if da is None and db is None:
            return None, None
        g = g.contiguous().float()
        _hip.check(_hip.lib().isi_mse_loss_bwd_f32(a.data_ptr(), b.data_ptr(), g.data_ptr(), a.numel(),
                                                   da.data_ptr() if da is not None else None,
                                                   db.data_ptr() if db is not None else None, _s(a)), "isi_mse_loss_bwd_f32")
        return da, db


def mse_loss(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean((input - target)^2) -- `torch.nn.functional.mse_loss(input, target)` (reduction 'mean') for fp32 tensors of one
    shape on the GPU; anything else goes to torch."""
    if (input.is_cuda and target.is_cuda and input.dtype == torch.float32 and target.dtype == torch.float32
            and input.shape == target.shape and input.numel() > 0
            # the kernels read 16-byte pieces: a contiguous view at an odd storage offset (flat[1:]) stays with torch
            and _aligned16(input) and _aligned16(target)):
        return _MSEFn.apply(input, target)
    return torch.nn.functional.mse_loss(input, target)


def _aligned16(t: torch.Tensor) -> bool:
    return (not t.is_contiguous()) or t.data_ptr() % 16 == 0     # (a non-contiguous operand is copied: fresh allocation)


class MSELoss(nn.Module):
    """Drop-in for `nn.MSELoss()` (reduction 'mean')."""
    reduction = "mean"

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return mse_loss(input, target)
