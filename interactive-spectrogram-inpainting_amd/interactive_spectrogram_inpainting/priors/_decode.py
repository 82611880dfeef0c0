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


def _refuse_aligned(layers) -> None:
    """The key/value-cached loops attend over ALL memory rows; the aligned decoder layer restricts them per target event."""
    from VQCPCB.transformer.transformer_custom import TransformerAlignedDecoderLayerCustom
    if any(isinstance(l, TransformerAlignedDecoderLayerCustom) for l in layers):
        raise NotImplementedError("KV-cached sampling is not built for use_aligned_decoder=True (full passes work)")


class IncrementalDecoder:
    def __init__(self, model, memory: torch.Tensor, batch_size: int):
        """memory: [S_src, B, d] encoder output."""
        dec = model.transformer.decoder
        self.layers = list(dec.layers)
        _refuse_aligned(self.layers)
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
        if batch_size > 32:
            raise NotImplementedError("incremental decoding supports batch sizes up to 32 (NativeSampler: 256)")

    def _linear(self, x, weight, bias, relu=False, residual=None, out=None):
        """Row GEMV of up to 8 rows per launch (`isi_linear_rows_f32`); larger batches go in groups of 8 rows -- a row's
        result does not depend on the rows it shares a launch with."""
        M = x.shape[0]
        if M <= 8:
            return _ops.linear_rows(x, weight, bias, relu=relu, residual=residual, out=out)
        if out is None:
            out = torch.empty(M, weight.shape[0], dtype=torch.float32, device=x.device)
        for lo in range(0, M, 8):
            hi = min(M, lo + 8)
            _ops.linear_rows(x[lo:hi], weight, bias, relu=relu, residual=residual[lo:hi] if residual is not None else None,
                             out=out[lo:hi])
        return out

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


class NativeSampler:
    """The whole sampling loop in one native call (`isi_prior_sample_run`): all
    per-position launches are enqueued without returning to Python and without
    host synchronisation; sampled indices stay on the device."""

    def __init__(self, model, memory: torch.Tensor, x_seq: torch.Tensor, codes: torch.Tensor,
                 mask_seq, uniforms: torch.Tensor):
        import ctypes as C
        import numpy as np
        from .. import _hip
        from .transformer import Seq2SeqInputKind
        self._C, self._hip = C, _hip
        dec_layers = list(model.transformer.decoder.layers)
        _refuse_aligned(dec_layers)
        if len(dec_layers) > _hip.ISI_MAX_LAYERS:
            raise NotImplementedError(f"more than {_hip.ISI_MAX_LAYERS} decoder layers")
        dev = memory.device
        d, B = model.d_model, x_seq.shape[1]
        if B > 256:
            raise NotImplementedError("native sampling supports batch sizes up to 256")
        self.model = model
        self.keep = []
        w = _hip.isi_prior_w()
        w.d_model, w.nhead = d, model.conditional_model_nhead
        w.dim_feedforward = dec_layers[0].linear1.out_features
        w.n_layers, w.n_class = len(dec_layers), model.n_class_target
        w.Cd, w.Ed = model.target_num_channels, model.target_num_events_with_start_symbol
        w.Ce, w.Ee = model.source_num_channels, model.source_num_events_with_start_symbol

        def ptr(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def attn(m):
            a = _hip.isi_attn_w()
            a.in_proj_weight, a.in_proj_bias = ptr(m.in_proj_weight), ptr(m.in_proj_bias)
            a.out_proj_weight, a.out_proj_bias = ptr(m.out_proj.weight), ptr(m.out_proj.bias)
            if m.rel_embeddings is not None:
                a.rel_embeddings, a.rel_rows = ptr(m.rel_embeddings), m.rel_embeddings.shape[1]
            return a

        for i, layer in enumerate(dec_layers):
            L = w.layers[i]
            L.self_attn, L.cross_attn = attn(layer.self_attn), attn(layer.multihead_attn)
            L.linear1_w, L.linear1_b = ptr(layer.linear1.weight), ptr(layer.linear1.bias)
            L.linear2_w, L.linear2_b = ptr(layer.linear2.weight), ptr(layer.linear2.bias)
            for k in (1, 2, 3):
                norm = getattr(layer, f"norm{k}")
                setattr(L, f"norm{k}_w", ptr(norm.weight))
                setattr(L, f"norm{k}_b", ptr(norm.bias))
        head = model.project_transformer_outputs_to_logits
        w.logits_w, w.logits_b = ptr(head.weight), ptr(head.bias)
        table = model._embedding_table(Seq2SeqInputKind.Target)
        w.embed_table, w.eff_dim = ptr(table), table.shape[1]
        self.w = w

        memory = memory.contiguous()
        self.memory_kv = torch.stack([l.multihead_attn.project_kv(memory) for l in dec_layers]).contiguous()
        S_t = x_seq.shape[0]
        self.kv_cache = torch.zeros(len(dec_layers), S_t, B, 2 * d, dtype=torch.float32, device=dev)
        self.x_seq, self.codes = x_seq, codes
        self.mask_host = np.ascontiguousarray(np.asarray(mask_seq, dtype=np.uint8))
        self.uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        n_scratch = _hip.lib().isi_prior_decode_scratch_floats(C.byref(w), B)
        self.scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        st = _hip.isi_prior_state()
        st.x_seq, st.kv_cache, st.memory_kv = x_seq.data_ptr(), self.kv_cache.data_ptr(), self.memory_kv.data_ptr()
        st.codes, st.mask = codes.data_ptr(), self.mask_host.ctypes.data
        st.uniforms, st.scratch, st.scratch_floats = self.uniforms.data_ptr(), self.scratch.data_ptr(), n_scratch
        st.S_t, st.S_src, st.S, st.B = S_t, memory.shape[0], codes.shape[1], B
        st.start_len = model.target_start_symbol.shape[1]
        self.state = st

    @torch.no_grad()
    def prefill(self, p0: int) -> None:
        """Key / value cache rows [0, p0) of every decoder layer from ONE causal pass over those rows (the
        full-sequence GEMM / attention kernels) instead of p0 single-row steps: what an unmasked prefix of an
        inpainting request needs -- its codes are known, only its keys and values are read later."""
        if p0 <= 0:
            return
        from VQCPCB.transformer.transformer_custom import _add_norm
        d = self.model.d_model
        x = self.x_seq[:p0].contiguous()
        for l, layer in enumerate(self.model.transformer.decoder.layers):
            sa = layer.self_attn
            qkv = sa._project(x, 0)                                  # [p0, B, 3d]
            self.kv_cache[l, :p0] = qkv[..., d:]
            a = _ops.rel_attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], sa.rel_embeddings, sa.nhead,
                                   sa.Cq, sa.Ck, sa.Ek, mask_mode=1)
            x1 = _add_norm(layer, sa.out_proj, a, x, layer.norm1)
            c = layer.multihead_attn(x1, None, None, kv=self.memory_kv[l])
            x2 = _add_norm(layer, layer.multihead_attn.out_proj, c, x1, layer.norm2)
            h = layer.linear1.run(x2, relu=True)
            x = _add_norm(layer, layer.linear2, h, x2, layer.norm3)

    @torch.no_grad()
    def run(self, p_begin: int, p_end: int, temperature: float, top_k: int, top_p: float) -> None:
        C, _hip = self._C, self._hip
        rc = _hip.lib().isi_prior_sample_run(C.byref(self.w), C.byref(self.state), p_begin, p_end,
                                             float(temperature), int(top_k), float(top_p),
                                             C.c_void_p(_hip.stream_ptr(self.x_seq.device)))
        _hip.check(rc, "isi_prior_sample_run")
