"""GANSynth range normalisation of [B, 2, F, T] spectrograms (log-magnitude, instantaneous frequency).

The reference imports `DataNormalizer` / `DataNormalizerStatistics` from the package
GANsynth_pytorch, which is absent from its tree; what is built here is the interface its call
sites need (vqvae.py:221-241,254-255,297-300; train_vqvae.py:646-676) around the published
GANSynth normalisation: each channel is mapped affinely so that the range seen in the data
lands on [-margin, +margin] (magnitude margin 0.8, IF margin 1.0)

    normalize:    y_c = a_c x_c + b_c        denormalize:  x_c = (y_c - b_c) / a_c
    a = margin * 2 / (max - min)             b = margin * (1 - 2 max / (max - min))

with (a, b) = (s_a, s_b) for channel 0 and (p_a, p_b) for channel 1.  PARITY UNPINNED (no reference
source or fixture); specification: oracle/spectrogram_oracle.py::normalize / denormalize.
One HBM pass on the GPU (isi_spec_affine_mask_f32), differentiable.
"""
from __future__ import annotations

import ctypes as C
import json
import pathlib
from typing import Iterable, Optional

import torch

from interactive_spectrogram_inpainting import _hip


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


class SpecAffineMaskFunction(torch.autograd.Function):
    """y0 = a0 x0 + b0 ; y1 = a1 x1 + b1, zeroed where y0 <= threshold (threshold None: no mask)."""

    @staticmethod
    def forward(ctx, x, a0, b0, a1, b1, threshold):
        _hip.require_gpu(x, "spectrogram")
        squeeze = x.dim() == 3
        x4 = (x.unsqueeze(0) if squeeze else x).contiguous()
        if x4.dim() != 4 or x4.shape[1] != 2 or x4.dtype != torch.float32:
            raise RuntimeError(f"expected a float32 [B, 2, F, T] spectrogram, got {tuple(x.shape)} {x.dtype}")
        B, _, H, W = x4.shape
        y = torch.empty_like(x4)
        _hip.check(_hip.lib().isi_spec_affine_mask_f32(x4.data_ptr(), None, y.data_ptr(), B, H * W, a0, b0, a1, b1,
                                                       0.0 if threshold is None else threshold,
                                                       int(threshold is not None), _s(x4)),
                   "isi_spec_affine_mask_f32")
        ctx.coef = (a0, a1, threshold, squeeze)
        if threshold is not None:
            ctx.save_for_backward(y)
        return y[0] if squeeze else y

    @staticmethod
    def backward(ctx, dy):
        a0, a1, threshold, squeeze = ctx.coef
        dy4 = (dy.unsqueeze(0) if squeeze else dy).contiguous()
        B, _, H, W = dy4.shape
        ref = ctx.saved_tensors[0] if threshold is not None else None
        dx = torch.empty_like(dy4)
        _hip.check(_hip.lib().isi_spec_affine_mask_f32(dy4.data_ptr(), ref.data_ptr() if ref is not None else None,
                                                       dx.data_ptr(), B, H * W, a0, 0.0, a1, 0.0,
                                                       0.0 if threshold is None else threshold,
                                                       int(threshold is not None), _s(dy4)),
                   "isi_spec_affine_mask_f32")
        return (dx[0] if squeeze else dx), None, None, None, None, None


class DataNormalizerStatistics(dict):
    """{s_a, s_b, p_a, p_b}: a plain mapping, so that it JSON-round-trips inside the VQ-VAE's
    instantiation parameters and `DataNormalizerStatistics(**statistics)` rebuilds it (vqvae.py:224-225)."""

    def __init__(self, s_a: float, s_b: float, p_a: float, p_b: float):
        super().__init__(s_a=float(s_a), s_b=float(s_b), p_a=float(p_a), p_b=float(p_b))

    s_a = property(lambda self: self["s_a"])
    s_b = property(lambda self: self["s_b"])
    p_a = property(lambda self: self["p_a"])
    p_b = property(lambda self: self["p_b"])


class DataNormalizer:
    def __init__(self, statistics: Optional[DataNormalizerStatistics] = None, dataloader: Optional[Iterable] = None,
                 magnitude_margin: float = 0.8, IF_margin: float = 1.0):
        if (statistics is None) == (dataloader is None):
            raise ValueError("give either precomputed statistics or a dataloader to measure them on")
        if statistics is None:
            statistics = self._measure(dataloader, magnitude_margin, IF_margin)
        if not isinstance(statistics, DataNormalizerStatistics):
            statistics = DataNormalizerStatistics(**statistics)
        self.statistics = statistics

    @staticmethod
    def _measure(dataloader, magnitude_margin, IF_margin) -> DataNormalizerStatistics:
        lo = [float("inf")] * 2
        hi = [float("-inf")] * 2
        for batch in dataloader:
            spec = batch[0] if isinstance(batch, (tuple, list)) else batch
            mn, mx = spec.amin(dim=(0, 2, 3)).tolist(), spec.amax(dim=(0, 2, 3)).tolist()
            lo = [min(a, b) for a, b in zip(lo, mn)]
            hi = [max(a, b) for a, b in zip(hi, mx)]
        coef = []
        for c, margin in enumerate((magnitude_margin, IF_margin)):
            rng = hi[c] - lo[c]
            if not rng > 0:
                raise ValueError("cannot normalise a constant channel")
            coef += [margin * 2.0 / rng, margin * (1.0 - 2.0 * hi[c] / rng)]
        return DataNormalizerStatistics(*coef)

    def normalize(self, spec: torch.Tensor) -> torch.Tensor:
        s = self.statistics
        return SpecAffineMaskFunction.apply(spec, s.s_a, s.s_b, s.p_a, s.p_b, None)

    def denormalize(self, spec: torch.Tensor, threshold: Optional[float] = None) -> torch.Tensor:
        """`threshold` fuses the masked-phase transform on the de-normalised log-magnitude into the same pass."""
        s = self.statistics
        return SpecAffineMaskFunction.apply(spec, 1.0 / s.s_a, -s.s_b / s.s_a, 1.0 / s.p_a, -s.p_b / s.p_a, threshold)

    def dump_statistics(self, path: pathlib.Path) -> None:
        with open(path, "w") as f:
            json.dump(dict(self.statistics), f)

    @classmethod
    def load_statistics(cls, path: pathlib.Path) -> "DataNormalizer":
        with open(path, "r") as f:
            return cls(DataNormalizerStatistics(**json.load(f)))
