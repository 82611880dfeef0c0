"""2-D codemap <-> 1-D sequence orderings (reference `priors/codemaps_helpers.py:16-243`).

The reference builds the orderings with `unfold/permute/reshape` chains; here
each helper owns an explicit permutation `perm[s] = f * T + t` (sequence
position -> cell of the flattened [F,T] map) and its inverse, so both
directions are one gather.  Orderings (pinned by `tests/golden/codemaps.npz`,
generated from the reference):

  Simple        s = t * F + f                       (time-major, low frequencies first)
  ZigZag(pf,pt) s = ((tp * F/pf + fp) * pt + ti) * pf + fi   with t = tp*pt + ti, f = fp*pf + fi
                (source time patch -> source frequency patch -> in-patch time -> in-patch frequency)

Pure index movement: works on any device, on index maps `[B,F,T]` and on
embedding / logit maps `[B,F,T,E]`.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, Tuple

import torch


class CodemapsHelper(ABC):
    def __init__(self, frequencies: int, duration: int):
        self.frequencies = frequencies
        self.duration = duration
        self.predict_frequencies_first = True
        self.predict_low_frequencies_first = True
        self._cache: Dict[Tuple[str, torch.device], torch.Tensor] = {}

    @abstractmethod
    def _permutation(self) -> torch.Tensor:
        """int64 [F*T]: flattened-map index (f*T + t) of every sequence position."""

    def _perm(self, device: torch.device, inverse: bool) -> torch.Tensor:
        key = ("inv" if inverse else "fwd", device)
        if key not in self._cache:
            perm = self._permutation()
            if inverse:
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(perm.numel())
                perm = inv
            self._cache[key] = perm.to(device)
        return self._cache[key]

    def to_sequence(self, codemap: torch.Tensor) -> torch.Tensor:
        if codemap.dim() not in (3, 4):
            raise ValueError(f"Unexpected number of dimensions {codemap.dim()} for input codemap")
        B, F, T = codemap.shape[:3]
        if (F, T) != (self.frequencies, self.duration):
            raise ValueError(f"expected a [{self.frequencies},{self.duration}] map, got [{F},{T}]")
        flat = codemap.reshape(B, F * T, *codemap.shape[3:])
        return flat.index_select(1, self._perm(codemap.device, inverse=False))

    def to_time_frequency_map(self, sequence: torch.Tensor,
                              permute_output_as_logits: bool = False) -> torch.Tensor:
        if sequence.dim() not in (2, 3):
            raise ValueError(f"Unexpected number of dimensions {sequence.dim()} for input sequence")
        B = sequence.shape[0]
        flat = sequence.index_select(1, self._perm(sequence.device, inverse=True))
        tf_map = flat.reshape(B, self.frequencies, self.duration, *sequence.shape[2:])
        if sequence.dim() == 3 and permute_output_as_logits:
            tf_map = tf_map.permute(0, 3, 1, 2)
        return tf_map


class SimpleCodemapsHelper(CodemapsHelper):
    def _permutation(self) -> torch.Tensor:
        s = torch.arange(self.frequencies * self.duration)
        t, f = s // self.frequencies, s % self.frequencies
        return f * self.duration + t


class ZigZagCodemapsHelper(CodemapsHelper):
    def __init__(self, frequencies: int, duration: int, patch_frequencies: int, patch_duration: int):
        super().__init__(frequencies, duration)
        if frequencies % patch_frequencies or duration % patch_duration:
            raise ValueError("patch sizes must divide the codemap shape")
        self.patch_frequencies = patch_frequencies
        self.patch_duration = patch_duration

    def _permutation(self) -> torch.Tensor:
        pf, pt = self.patch_frequencies, self.patch_duration
        n_fp = self.frequencies // pf
        s = torch.arange(self.frequencies * self.duration)
        fi = s % pf
        ti = (s // pf) % pt
        fp = (s // (pf * pt)) % n_fp
        tp = s // (pf * pt * n_fp)
        return (fp * pf + fi) * self.duration + (tp * pt + ti)
