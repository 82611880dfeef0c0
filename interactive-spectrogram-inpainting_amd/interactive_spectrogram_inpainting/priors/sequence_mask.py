"""Training-time inpainting mask samplers (reference `priors/sequence_mask.py:7-77`).
Host-side random draws; bool masks `[batch, sequence]`, True = masked."""
from __future__ import annotations

import math
import random

import torch


class SequenceMask:
    def __init__(self, sequence_duration: int, mask_token_index: int):
        self.sequence_duration = sequence_duration
        self.mask_token_index = mask_token_index

    def sample_mask(self, batch_size: int = 1) -> torch.Tensor:
        raise NotImplementedError("subclass this")

    def apply_mask(self, input: torch.Tensor) -> torch.Tensor:
        mask = self.sample_mask(batch_size=input.shape[0]).to(input.device)
        return input.masked_fill(mask, self.mask_token_index)


class BernoulliSequenceMask(SequenceMask):
    """Every position masked independently with a fixed probability."""

    def __init__(self, probability: float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.probability = probability

    def sample_mask(self, batch_size: int = 1) -> torch.Tensor:
        return torch.rand(batch_size, self.sequence_duration) < self.probability


class UniformProbabilityBernoulliSequenceMask(SequenceMask):
    """Bernoulli mask whose probability is drawn uniformly in [low, high] once per batch."""

    def __init__(self, low: float = 0., high: float = 1., *args, **kwargs):
        assert 0 <= low < high <= 1
        super().__init__(*args, **kwargs)
        self.low, self.high = low, high

    def sample_mask(self, batch_size: int = 1) -> torch.Tensor:
        p = random.uniform(self.low, self.high)
        return torch.rand(batch_size, self.sequence_duration) < p


class UniformMaskedAmountSequenceMask(SequenceMask):
    """Exactly n positions masked per row, n ~ U{ceil(ratio * S), ..., S} once per batch."""

    def __init__(self, min_masking_ratio: float = 0., *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.min_masking_ratio = min_masking_ratio
        self.min_masked_amount = math.ceil(self.sequence_duration * self.min_masking_ratio)

    def sample_mask(self, batch_size: int = 1) -> torch.Tensor:
        n = int(torch.randint(self.min_masked_amount, self.sequence_duration + 1, (1,)).item())
        # the n smallest of S i.i.d. uniforms = a uniformly random n-subset
        ranks = torch.rand(batch_size, self.sequence_duration).argsort(dim=1).argsort(dim=1)
        return ranks < n


class ContiguousZonesSequenceMask(SequenceMask):
    def sample_mask(self, batch_size: int = 1) -> torch.Tensor:
        raise NotImplementedError("TODO")  # unimplemented in the reference too (sequence_mask.py:80-82)
