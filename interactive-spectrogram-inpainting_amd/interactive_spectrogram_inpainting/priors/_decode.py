"""Incremental (key/value-cached) decoding of the prior's decoder stack.

The reference re-runs the full decoder over all positions for every sampled
token (sample.py:268-283): O(S^2) layer passes per codemap.  Decoder
self-attention is causal (priors/transformer.py:483-500) and inputs at positions
< p never change once written, so row p of a full pass equals the row computed
from cached keys / values of rows <= p; the encoder memory and its projected
keys / values are computed once.  `tests/test_prior_gpu.py` checks that the
incremental rows equal the full-sequence pass."""
from __future__ import annotations

from typing import List

import torch

from . import _ops


class IncrementalDecoder:
    def __init__(self, model, memory: torch.Tensor, batch_size: int):
        """memory: [S_src, B, d] encoder output."""
        dec = model.transformer.decoder
        self.layers = list(dec.layers)
        self.model = model
        d = model.d_model
        S_t = model.target_transformer_sequence_length_with_start_symbol
        self.B, self.d = batch_size, d
        self.Cd, self.Ed = model.target_num_channels, model.target_num_events_with_start_symbol
        self.Ce, self.Ee = model.source_num_channels, model.source_num_events_with_start_symbol
        memory = memory.contiguous()
        self.S_src = memory.shape[0]
        self.memory_kv: List[torch.Tensor] = [l.multihead_attn.project_kv(memory) for l in self.layers]
        self.cache: List[torch.Tensor] = [
            torch.zeros(S_t, batch_size, 2 * d, dtype=torch.float32, device=memory.device) for _ in self.layers]
        self._lin = _ops.linear_rows if batch_size <= 8 else None

    def _linear(self, x, weight, bias, relu=False, residual=None, out=None):
        if self._lin is not None:
            return self._lin(x, weight, bias, relu=relu, residual=residual, out=out)
        raise NotImplementedError("incremental decoding supports batch sizes up to 8")

    @torch.no_grad()
    def step(self, p: int, x: torch.Tensor) -> torch.Tensor:
        """Decoder output row at sequence position p for the input row x [B, d]."""
        d = self.d
        for l, layer in enumerate(self.layers):
            sa, ca = layer.self_attn, layer.multihead_attn
            W, b = sa.in_proj_weight, sa.in_proj_bias
            q = self._linear(x, W[:d], b[:d])
            self._linear(x, W[d:], b[d:], out=self.cache[l][p])           # k|v of this row -> cache slot p
            kc = self.cache[l]
            a = _ops.rel_attention_decode(q, kc[..., :d], kc[..., d:], sa.rel_embeddings, sa.nhead,
                                          n_keys=p + 1, q_pos=p, Cq=self.Cd, Ck=self.Cd, Ek=self.Ed)
            x1 = layer.norm1.run(self._linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x))
            Wc, bc = ca.in_proj_weight, ca.in_proj_bias
            qc = self._linear(x1, Wc[:d], bc[:d])
            mkv = self.memory_kv[l]
            c = _ops.rel_attention_decode(qc, mkv[..., :d], mkv[..., d:], ca.rel_embeddings, ca.nhead,
                                          n_keys=self.S_src, q_pos=p, Cq=self.Cd, Ck=self.Ce, Ek=self.Ee)
            x2 = layer.norm2.run(self._linear(c, ca.out_proj.weight, ca.out_proj.bias, residual=x1))
            h = self._linear(x2, layer.linear1.weight, layer.linear1.bias, relu=True)
            x = layer.norm3.run(self._linear(h, layer.linear2.weight, layer.linear2.bias, residual=x2))
        return x

    @torch.no_grad()
    def logits(self, out_row: torch.Tensor) -> torch.Tensor:
        head = self.model.project_transformer_outputs_to_logits
        return self._linear(out_row, head.weight, head.bias)
