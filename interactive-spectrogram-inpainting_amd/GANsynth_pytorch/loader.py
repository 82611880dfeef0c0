"""Waveform -> spectrogram batches on the GPU, and the masked-phase transform.

Interface of the absent package GANsynth_pytorch.loader at the reference's call sites
(train_vqvae.py:582-643, extract_code.py:172-206, vqvae.py:238-241): the loaders wrap a
torch DataLoader over (audio, *labels) samples and turn every audio batch into a [B, 2, F, T]
spectrogram with the helper's HIP front-end; `make_masked_phase_transform(min_magnitude)` zeroes the
instantaneous frequency wherever the log-magnitude is at or below `min_magnitude` (the phase of a
silent bin carries no information).  PARITY UNPINNED; specification:
oracle/spectrogram_oracle.py::mask_phase.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch.utils.data import DataLoader

from .normalizer import SpecAffineMaskFunction


def make_masked_phase_transform(min_magnitude: float) -> Callable[[torch.Tensor], torch.Tensor]:
    def transform(spec: torch.Tensor) -> torch.Tensor:
        return SpecAffineMaskFunction.apply(spec, 1.0, 0.0, 1.0, 0.0, float(min_magnitude))
    return transform


class WavToSpectrogramDataLoader:
    """Iterates `DataLoader(dataset, **kwargs)`; yields (spectrogram on the helper's device, *rest)."""

    def __init__(self, dataset, spectrograms_helper, transform: Optional[Callable] = None, **kwargs):
        self.spectrograms_helper = spectrograms_helper
        self.transform = transform
        self.dataset = dataset
        self.dataloader = DataLoader(dataset, **kwargs)
        self.batch_size = self.dataloader.batch_size
        self.sampler = self.dataloader.sampler

    def __len__(self):
        return len(self.dataloader)

    def __iter__(self):
        device = self.spectrograms_helper.device
        for batch in self.dataloader:
            audio, rest = (batch[0], tuple(batch[1:])) if isinstance(batch, (tuple, list)) else (batch, ())
            spec = self.spectrograms_helper.to_spectrogram(audio.to(device, non_blocking=True).float())
            if self.transform is not None:
                spec = self.transform(spec)
            yield (spec, *rest)


class MaskedPhaseWavToSpectrogramDataLoader(WavToSpectrogramDataLoader):
    def __init__(self, dataset, spectrograms_helper, **kwargs):
        super().__init__(dataset, spectrograms_helper,
                         transform=make_masked_phase_transform(spectrograms_helper.safelog_eps), **kwargs)
