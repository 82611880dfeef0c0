"""Vector-quantisation bottleneck on MI355X.

Drop-in for the reference's `vqvae/bottleneck.py:30-119`
(`QuantizedBottleneck`, `UnquantizedBottleneck`): same buffers (`embed [D,K]`,
`cluster_size [K]`, `embed_avg [D,K]`), same return tuple
`(quantize, diff, embed_ind, perplexity)`.  The L2 nearest-neighbour search runs
in `isi_vq_nearest_f32` (codebook resident in LDS, exact-fp32 matrix pipe,
lane-local arg-min) without materialising the [N,K] distance / one-hot matrices.
`QuantizedBottleneckWithRestarts` (`:122-166`) wraps an absent third-party
package and is out of scope.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import nn, Tensor

from .. import _hip
from . import _ops


class QuantizedBottleneck(nn.Module):
    cluster_size: Tensor

    def __init__(self, dim: int, n_embed: int, decay: float = 0.99, eps: float = 1e-5,
                 embeddings_initial_variance: float = 1,
                 corruption_weights: Optional[List[float]] = None):
        super().__init__()
        self.dim = dim
        self.n_embed = n_embed
        self.decay = decay
        self.eps = eps
        self.corruption_weights = corruption_weights
        self.embeddings_initial_variance = embeddings_initial_variance
        embed = torch.randn(dim, n_embed) * np.sqrt(self.embeddings_initial_variance)
        self.register_buffer('embed', embed)
        self.register_buffer('cluster_size', torch.zeros(n_embed))
        self.register_buffer('embed_avg', embed.clone())
        self._packed = None
        self._packed_key = None

    def packed(self):
        """(codes [K,D], e2 [K]) for the HIP kernels, cached per buffer version."""
        key = (_hip.version_of(self.embed), self.embed.data_ptr(), self.embed.device)
        if self._packed is None or self._packed_key != key:
            self._packed = _ops.pack_codebook(self.embed)
            self._packed_key = key
        return self._packed

    def forward(self, input: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """input [..., D] channels-last.  Train mode also corrupts the indices when
        `corruption_weights` is set and updates the codebook buffers (bottleneck.py:63-73,79-92)."""
        if self.training:
            from ._train import QuantizeTrainFunction
            return QuantizeTrainFunction.apply(self, input)
        codes, e2 = self.packed()
        return _ops.vq_nearest(input, codes, e2)

    def embed_code(self, embed_id: Tensor) -> Tensor:
        # range check on the host like F.embedding's (one device sync; skipped while a HIP graph is being captured)
        if embed_id.numel() and embed_id.is_cuda and not torch.cuda.is_current_stream_capturing() and \
                (int(embed_id.min()) < 0 or int(embed_id.max()) >= self.n_embed):
            raise IndexError("index out of range in self")  # same failure class as F.embedding
        codes, _ = self.packed()
        return _ops.embed_code(embed_id, codes)


class UnquantizedBottleneck(QuantizedBottleneck):
    def forward(self, input):
        output = input
        diff = torch.zeros((1,), dtype=input.dtype, device=input.device)
        embed_ind = None
        code_assignation_perplexity = torch.as_tensor([np.inf], device=input.device)
        return output, diff, embed_ind, code_assignation_perplexity

    def embed_code(self, embed_ind):
        raise NotImplementedError
