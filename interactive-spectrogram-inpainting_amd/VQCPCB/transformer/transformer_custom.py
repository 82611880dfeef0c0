"""Relative-attention transformer layers on MI355X.

Same class names, constructor keywords and call / return conventions as the
module the reference imports (priors/transformer.py:370-417,756-779):

    TransformerEncoderLayerCustom(d_model, nhead, attention_bias_type, num_channels, num_events)
    TransformerDecoderLayerCustom(d_model, nhead, attention_bias_type_self, attention_bias_type_cross,
                                  num_channels_encoder, num_events_encoder,
                                  num_channels_decoder, num_events_decoder)
    TransformerEncoderCustom(encoder_layer, num_layers).forward(src, mask=)            -> (memory, [])
    TransformerDecoderCustom(decoder_layer, num_layers).forward(tgt, memory, tgt_mask=, memory_mask=) -> (out, [])
    TransformerCustom(nhead, custom_encoder, custom_decoder, d_model)   .encoder / .decoder

Inputs are time-major `[S, B, d_model]` fp32.  Masks may be the additive float
matrices the reference builds (`causal_mask`, its transpose; recognised and
replaced by an in-kernel predicate) or the strings 'causal' / 'anticausal'.
Arithmetic: see oracle/prior_oracle.py (specification; parity unpinned because
the original package is not available).

Training: when autograd is recording, every operator runs through the
`torch.autograd.Function`s of priors/_train.py (hand-written backward kernels).
In `.train()` mode dropout(p) is applied where torch's post-norm layers apply it
(after each attention output projection, after the feed-forward activation and
after the second feed-forward linear); attention probabilities are not dropped.
"""
from __future__ import annotations

import copy
import math
import weakref
import os
from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops
from interactive_spectrogram_inpainting.priors import _train

MaskArg = Union[None, str, torch.Tensor]


class _LinearParams(nn.Module):
    """nn.Linear-shaped parameter holder (same default init); GEMM operand cached per version."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)
        self._packed, self._key = None, None
        self._packed_t, self._key_t = None, None
        self._range = _ops.WeightRange()

    def packed(self):
        inference = not (torch.is_grad_enabled() and self.weight.requires_grad)
        key = (_hip.version_of(self.weight), self.weight.data_ptr(), inference)
        if self._key != key:
            ok = self._range.update(self.weight, inference)
            self._packed, self._key = _ops.pack_linear_weight(self.weight, range_check=ok, with_f16=inference and ok), key
        return self._packed

    def packed_t(self):
        """GEMM operand of the input gradient: W^T ([in, out]) packed like a forward weight."""
        key = (_hip.version_of(self.weight), self.weight.data_ptr())
        if self._key_t != key:
            self._packed_t, self._key_t = _ops.pack_linear_weight_t(self.weight), key
        return self._packed_t

    def run(self, x, relu=False, residual=None, rectified_input=False, grad_pre_gated=False, dropout_p=0.0,
            input_keep_scale=1.0, carry=False):
        """`rectified_input` / `grad_pre_gated` / `dropout_p` / `input_keep_scale`: see _train.LinearFn (a feed-forward
        block's pair of layers)."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            through = carry and _CARRY
            out = _train.LinearFn.apply(x, self.weight, self.bias, residual, relu, self.packed(), self.packed_t,
                                        rectified_input, grad_pre_gated, dropout_p, input_keep_scale, through)
            return (out, x) if (carry and not through) else out
        if dropout_p > 0.0:
            raise RuntimeError("fused dropout is a training-path feature (no gradient is being recorded)")
        y = _ops.linear(x, self.packed(), self.bias, self.out_features, relu=relu, residual=residual)
        return (y, x) if carry else y


class _LayerNormParams(nn.Module):
    def __init__(self, d: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.eps = eps

    def run(self, x, residual=None, dropout_p=0.0):
        """LayerNorm(dropout(x) + residual); the dropout (train mode) runs inside the LayerNorm kernels."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return _train.LayerNormFn.apply(x, residual, self.weight, self.bias, self.eps, dropout_p)
        if dropout_p > 0.0:
            return _ops.layernorm(x, self.weight, self.bias, self.eps, residual=residual, dropout_p=dropout_p,
                                  dropout_seed_=_ops.dropout_seed())
        return _ops.layernorm(x, self.weight, self.bias, self.eps, residual=residual)


def _classify_mask(mask: MaskArg, Sq: int, Sk: int, device) -> Tuple[int, Optional[torch.Tensor]]:
    """-> (mask_mode, dense_mask): 0 none, 1 causal, 2 anti-causal."""
    if mask is None:
        return 0, None
    if isinstance(mask, str):
        return {"causal": 1, "anticausal": 2, "none": 0}[mask], None
    m = mask.detach()
    if m.shape != (Sq, Sk):
        raise RuntimeError(f"attention mask of shape {tuple(m.shape)} for a {Sq}x{Sk} score matrix")
    # masks are never touched by an optimizer: the tensor's own version counter is the whole key (the global
    # optimizer-step count of _hip.version_of would miss after every step: a host sync and a D2H copy each time)
    key = (m.data_ptr(), m._version, tuple(m.shape), m.device)
    hit = _classify_mask.cache.get(key)
    if hit is not None:
        # (the key is an address: a verdict only stands for the tensor it was formed on -- another mask allocated where a
        # dead one lived must not inherit it)
        ref, hit = hit
        if ref() is not mask:
            hit = None
    if hit is None:
        allowed = (m.cpu() == 0)
        tril = torch.ones(Sq, Sk, dtype=torch.bool).tril()
        if Sq == Sk and torch.equal(allowed, tril):
            hit = 1
        elif Sq == Sk and torch.equal(allowed, tril.t()):
            hit = 2
        elif bool(allowed.all()):
            hit = 0
        else:
            hit = -1
        if len(_classify_mask.cache) > 64:
            _classify_mask.cache.clear()
        _classify_mask.cache[key] = (weakref.ref(mask), hit)
    if hit >= 0:
        return hit, None
    return 0, m.to(device=device, dtype=torch.float32).contiguous()


_classify_mask.cache = {}


class RelativeMultiheadAttention(nn.Module):
    """q.k + q.e[r] attention; r(i,j) = i//Cq - j//Ck + Ek - 1 (oracle/prior_oracle.py)."""

    def __init__(self, d_model: int, nhead: int, attention_bias_type: str, num_channels_q: int,
                 num_events_q: int, num_channels_k: int, num_events_k: int):
        super().__init__()
        if d_model % nhead:
            raise ValueError("d_model must be divisible by nhead")
        if attention_bias_type not in ("relative_attention", "relative_attention_target_source", "no_bias"):
            raise ValueError(f"unknown attention_bias_type {attention_bias_type}")
        self.d_model, self.nhead, self.head_dim = d_model, nhead, d_model // nhead
        self.attention_bias_type = attention_bias_type
        self.Cq, self.Eq, self.Ck, self.Ek = num_channels_q, num_events_q, num_channels_k, num_events_k
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.out_proj = _LinearParams(d_model, d_model)
        if attention_bias_type != "no_bias":
            self.rel_embeddings = nn.Parameter(
                torch.randn(nhead, num_events_q + num_events_k - 1, self.head_dim) * self.head_dim ** -0.5)
        else:
            self.register_parameter("rel_embeddings", None)
        self._packed, self._key = None, None
        self._packed_t, self._key_t = None, None
        self._range = _ops.WeightRange()

    def _pack_t(self, which: int):
        """W^T operand of an input gradient: which = 0 all three projections, 1 q only, 2 k|v -- each packed when it is
        first asked for (self-attention only ever uses 0, cross-attention 1 and 2: a launch per operand and step)."""
        key = (_hip.version_of(self.in_proj_weight), self.in_proj_weight.data_ptr())
        if self._key_t != key:
            self._packed_t, self._key_t = [None, None, None], key
        if self._packed_t[which] is None:
            d = self.d_model
            W = self.in_proj_weight       # (not a detached alias: the pack group holds the PARAMETER weakly, _ops._WtPackGroup)
            self._packed_t[which] = _ops.pack_linear_weight_t((W, W[:d], W[d:])[which], owner=W)
        return self._packed_t[which]

    def _project(self, x, which: int, carry: bool = False):
        """which: 0 = q|k|v, 1 = q, 2 = k|v of the fused in-projection.  carry: (projection, x) -- _train.LinearFn."""
        d = self.d_model
        lo, hi = ((0, 3 * d), (0, d), (d, 3 * d))[which]
        # (the whole parameter, not a [0:3d] slice of it: a slice's backward materialises a zero-filled copy of the parameter)
        W, b = (self.in_proj_weight, self.in_proj_bias) if which == 0 else (self.in_proj_weight[lo:hi], self.in_proj_bias[lo:hi])
        if torch.is_grad_enabled() and (x.requires_grad or self.in_proj_weight.requires_grad):
            through = carry and _CARRY
            out = _train.LinearFn.apply(x, W, b, None, False, self._packs()[which], lambda: self._pack_t(which),
                                        False, False, 0.0, 1.0, through)
            return (out, x) if (carry and not through) else out
        y = _ops.linear(x, self._packs()[which], b, hi - lo)
        return (y, x) if carry else y

    def _packs(self):
        inference = not (torch.is_grad_enabled() and self.in_proj_weight.requires_grad)
        key = (_hip.version_of(self.in_proj_weight), self.in_proj_weight.data_ptr(), inference)
        if self._key != key:
            d = self.d_model
            W = self.in_proj_weight.detach()
            ok = self._range.update(self.in_proj_weight, inference)      # one check covers the three row ranges
            f16 = inference and ok     # weights that stay put: pair copies for the GEMM kernel
            self._packed = (_ops.pack_linear_weight(W, range_check=ok, with_f16=f16),
                            _ops.pack_linear_weight(W[:d], range_check=ok, with_f16=f16),
                            _ops.pack_linear_weight(W[d:], range_check=ok, with_f16=f16))
            self._key = key
        return self._packed

    def project_kv(self, mem: torch.Tensor) -> torch.Tensor:
        """[Sk,B,d] -> fused [Sk,B,2d] keys|values (cacheable: depends on mem only)."""
        return self._project(mem, 2)

    def forward(self, x: torch.Tensor, mem: Optional[torch.Tensor], mask: MaskArg = None,
                kv: Optional[torch.Tensor] = None, carry: bool = False):
        """Attention output BEFORE the output projection ([Sq,B,d]); `mem is None` = self-attention.  carry: (output, x) with
        x handed on THROUGH the in-projection's autograd node, for the residual add behind the block (_train.LinearFn)."""
        d = self.d_model
        Sq = x.shape[0]
        xc = x
        if mem is None and kv is None:
            a, b = self._project(x, 0, carry), None
            if carry:
                a, xc = a
            Sk = Sq
        else:
            W = self.in_proj_weight
            if (kv is None and torch.is_grad_enabled() and W.requires_grad and d % 32 == 0 and self.in_proj_bias is not None):
                # training: both in-projections as one autograd node (their weight gradients fill one [3d, d] tensor)
                packs = self._packs()
                through = carry and _CARRY
                out = _train.CrossInProjFn.apply(x, mem, W, self.in_proj_bias, packs[1], packs[2],
                                                 lambda: self._pack_t(1), lambda: self._pack_t(2), through)
                a, b = out[0], out[1]
                if through:
                    xc = out[2]
            else:
                a = self._project(x, 1, carry)
                if carry:
                    a, xc = a
                b = kv if kv is not None else self.project_kv(mem)
            Sk = b.shape[0]
        mode, dense = _classify_mask(mask, Sq, Sk, x.device)
        if torch.is_grad_enabled() and a.requires_grad:
            o = _train.RelAttentionFn.apply(a, b, self.rel_embeddings, self.nhead, self.Cq, self.Ck, self.Ek, mode, dense)
            return (o, xc) if carry else o
        if b is None:
            q, k, v = a[..., :d], a[..., d:2 * d], a[..., 2 * d:]
        else:
            q, k, v = a, b[..., :d], b[..., d:]
        o = _ops.rel_attention(q, k, v, self.rel_embeddings, self.nhead, self.Cq, self.Ck, self.Ek,
                               mask_mode=mode, dense_mask=dense)
        return (o, xc) if carry else o




# ISI_FF_GATE=0: the feed-forward block's ReLU backward as its own mask pass (clone + kernel) instead of the gated epilogue
# of linear2's input-gradient GEMM (A/B switch)
_FF_GATE = os.environ.get("ISI_FF_GATE", "1") != "0"


# ISI_FUSED_DROPOUT=0: the layers' dropouts as torch's kernels (A/B switch)
_FUSED_DROPOUT = os.environ.get("ISI_FUSED_DROPOUT", "1") != "0"


# ISI_RESIDUAL_CARRY=0: a residual branch's input read twice by autograd (its two gradients summed by a kernel of torch's)
# instead of being handed on through the branch's first linear node (A/B switch; _train.LinearFn: `carry`)
_CARRY = os.environ.get("ISI_RESIDUAL_CARRY", "1") != "0"


def _drop(layer: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if layer.training and layer.dropout > 0:
        return nn.functional.dropout(x, layer.dropout, True)
    return x


def _add_norm(layer: nn.Module, lin: _LinearParams, x: torch.Tensor, residual: torch.Tensor,
              norm: _LayerNormParams, rectified_input: bool = False, input_keep_scale: float = 1.0) -> torch.Tensor:
    """norm(residual + dropout(lin(x))): the residual add rides in the GEMM epilogue when nothing is
    dropped, in the LayerNorm kernel otherwise.  `rectified_input`: x is a ReLU's output (_train.LinearFn)."""
    if layer.training and layer.dropout > 0:
        y = lin.run(x, rectified_input=rectified_input, input_keep_scale=input_keep_scale)
        if _FUSED_DROPOUT and y.numel() < (1 << 32):      # the dropout inside the LayerNorm kernels (forward and backward)
            return norm.run(y, residual=residual, dropout_p=layer.dropout)
        return norm.run(nn.functional.dropout(y, layer.dropout, True), residual=residual)
    return norm.run(lin.run(x, residual=residual, rectified_input=rectified_input))


def _feed_forward(layer: nn.Module, x: torch.Tensor, norm: _LayerNormParams) -> torch.Tensor:
    """norm(x + dropout(linear2(dropout(relu(linear1(x)))))).  linear2's input gradient is gated by h > 0 in its GEMM
    epilogue and linear1 skips its mask pass (ISI_FF_GATE); in a training step the hidden dropout ([M, 2048]: the
    largest element-wise tensor of a layer) rides in linear1's GEMM epilogue and its backward in linear2's gate."""
    p = layer.dropout if layer.training else 0.0
    M = x.numel() // x.shape[-1]
    fused = (_FUSED_DROPOUT and _FF_GATE and p > 0 and torch.is_grad_enabled()
             and (x.requires_grad or layer.linear1.weight.requires_grad)
             and _ops.fused_tails_ok(M, layer.linear1.out_features, layer.linear1.in_features)
             and _ops.fused_tails_ok(M, layer.linear2.in_features, layer.linear2.out_features))
    if fused:
        h, x = layer.linear1.run(x, relu=True, grad_pre_gated=True, dropout_p=p, carry=True)   # (x: handed on through linear1's node)
        return _add_norm(layer, layer.linear2, h, x, norm, rectified_input=True, input_keep_scale=1.0 / (1.0 - p))
    h, x = layer.linear1.run(x, relu=True, grad_pre_gated=_FF_GATE, carry=True)
    return _add_norm(layer, layer.linear2, _drop(layer, h), x, norm, rectified_input=_FF_GATE)


class TransformerEncoderLayerCustom(nn.Module):
    def __init__(self, d_model: int, nhead: int, attention_bias_type: str = "relative_attention",
                 num_channels: int = 1, num_events: int = 1, dim_feedforward: int = 2048,
                 dropout: float = 0.1):
        super().__init__()
        self.self_attn = RelativeMultiheadAttention(d_model, nhead, attention_bias_type, num_channels,
                                                    num_events, num_channels, num_events)
        self.linear1 = _LinearParams(d_model, dim_feedforward)
        self.linear2 = _LinearParams(dim_feedforward, d_model)
        self.norm1 = _LayerNormParams(d_model)
        self.norm2 = _LayerNormParams(d_model)
        self.dropout = dropout  # identity in eval mode

    def forward(self, src: torch.Tensor, src_mask: MaskArg = None) -> torch.Tensor:
        a, src = self.self_attn(src, None, src_mask, carry=True)
        x = _add_norm(self, self.self_attn.out_proj, a, src, self.norm1)
        return _feed_forward(self, x, self.norm2)


class TransformerDecoderLayerCustom(nn.Module):
    def __init__(self, d_model: int, nhead: int, attention_bias_type_self: str = "relative_attention",
                 attention_bias_type_cross: str = "relative_attention_target_source",
                 num_channels_encoder: int = 1, num_events_encoder: int = 1,
                 num_channels_decoder: int = 1, num_events_decoder: int = 1,
                 dim_feedforward: int = 2048, dropout: float = 0.1):
        super().__init__()
        self.self_attn = RelativeMultiheadAttention(d_model, nhead, attention_bias_type_self,
                                                    num_channels_decoder, num_events_decoder,
                                                    num_channels_decoder, num_events_decoder)
        self.multihead_attn = RelativeMultiheadAttention(d_model, nhead, attention_bias_type_cross,
                                                         num_channels_decoder, num_events_decoder,
                                                         num_channels_encoder, num_events_encoder)
        self.linear1 = _LinearParams(d_model, dim_feedforward)
        self.linear2 = _LinearParams(dim_feedforward, d_model)
        self.norm1 = _LayerNormParams(d_model)
        self.norm2 = _LayerNormParams(d_model)
        self.norm3 = _LayerNormParams(d_model)
        self.dropout = dropout

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, tgt_mask: MaskArg = None,
                memory_mask: MaskArg = None, memory_kv: Optional[torch.Tensor] = None) -> torch.Tensor:
        a, tgt = self.self_attn(tgt, None, tgt_mask, carry=True)
        x = _add_norm(self, self.self_attn.out_proj, a, tgt, self.norm1)
        c, x = self.multihead_attn(x, memory, memory_mask, kv=memory_kv, carry=True)
        x = _add_norm(self, self.multihead_attn.out_proj, c, x, self.norm2)
        return _feed_forward(self, x, self.norm3)


class TransformerAlignedDecoderLayerCustom(TransformerDecoderLayerCustom):
    """The reference's hierarchical 'aligned' decoder layer (priors/transformer.py:388-396, `use_aligned_decoder=True`): "this
    computes cross-attention only with tokens from the source that directly condition underlying tokens in the target".  Its
    definition lives only in the absent package; OUR specification (parity unpinned, like every layer of this file): a
    TransformerDecoderLayerCustom whose cross-attention lets target token i see the source tokens of ITS event only --

        allowed(i, j)  <=>  i // num_channels_decoder == j // num_channels_encoder

    (for the bottom prior: the four codes of a 2 x 2 patch attend to the one top code above them; the start symbols are event 0 on
    both sides) -- as an additive mask handed to the attention kernels, combined with whatever `memory_mask` the caller passes.
    Self-attention, feed-forward and every parameter are the parent's, so state dicts are interchangeable.  KV-cached sampling is
    not built for it (priors/_decode.py raises)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._align = {}

    def alignment_mask(self, St: int, Ss: int, device) -> torch.Tensor:
        key = (St, Ss, str(device))
        m = self._align.get(key)
        if m is None:
            ev_t = torch.arange(St) // self.multihead_attn.Cq
            ev_s = torch.arange(Ss) // self.multihead_attn.Ck
            m = torch.full((St, Ss), float("-inf"))
            m[ev_t[:, None] == ev_s[None, :]] = 0.0
            if bool(torch.isinf(m).all(dim=1).any()):
                raise ValueError(f"aligned decoder: {St} target tokens of {self.multihead_attn.Cq} per event against {Ss} source "
                                 f"tokens of {self.multihead_attn.Ck}: a target event without a source event")
            m = self._align[key] = m.to(device)
        return m

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, tgt_mask: MaskArg = None,
                memory_mask: MaskArg = None, memory_kv: Optional[torch.Tensor] = None) -> torch.Tensor:
        Ss = memory_kv.shape[0] if memory_kv is not None else memory.shape[0]
        m = self.alignment_mask(tgt.shape[0], Ss, tgt.device)
        if memory_mask is not None:
            if isinstance(memory_mask, str):
                raise NotImplementedError("the aligned decoder combines its alignment with additive tensor masks only")
            m = self._combined(m, memory_mask)
        return super().forward(tgt, memory, tgt_mask, m, memory_kv)

    def _combined(self, align: torch.Tensor, memory_mask: torch.Tensor) -> torch.Tensor:
        """alignment + the caller's additive mask, formed ONCE per (mask tensor, version): a fresh sum per forward would be a
        new tensor every time -- the classification cache of `_classify_mask` (keyed on the tensor's identity) would never
        hit, every layer would pay a device-to-host copy per step, and a recorded step could not contain it.
        A target token left without any allowed source column (unequal event counts, or a caller mask that removes the only
        aligned key) would have an all -inf row, i.e. a NaN softmax: rejected here, where the mask is formed."""
        key = (memory_mask.data_ptr(), memory_mask._version, tuple(memory_mask.shape), str(memory_mask.device), tuple(align.shape))
        hit = self._align.get("combined")
        if hit is not None and hit[0] == key and hit[1]() is memory_mask:
            return hit[2]
        m = align + memory_mask.detach().to(align.device)
        if bool(torch.isinf(m).all(dim=1).any()):
            raise ValueError("aligned decoder: a target token has no allowed source token (alignment combined with memory_mask)")
        self._align["combined"] = (key, weakref.ref(memory_mask), m)
        return m


class TransformerEncoderCustom(nn.Module):
    def __init__(self, encoder_layer: nn.Module, num_layers: int):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        for layer in self.layers[1:]:
            _reinit(layer)

    def forward(self, src: torch.Tensor, mask: MaskArg = None) -> Tuple[torch.Tensor, List]:
        x = src.contiguous()
        for layer in self.layers:
            x = layer(x, mask)
        return x, []


class TransformerDecoderCustom(nn.Module):
    def __init__(self, decoder_layer: nn.Module, num_layers: int):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        for layer in self.layers[1:]:
            _reinit(layer)

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, tgt_mask: MaskArg = None,
                memory_mask: MaskArg = None, condition: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, List]:
        if condition is not None:
            raise NotImplementedError("local class conditioning is deprecated in the reference "
                                      "(priors/transformer.py:107-109)")
        x = tgt.contiguous()
        memory = memory.contiguous()
        for layer in self.layers:
            x = layer(x, memory, tgt_mask, memory_mask)
        return x, []


class TransformerCustom(nn.Module):
    def __init__(self, nhead: int, custom_encoder: nn.Module, custom_decoder: nn.Module, d_model: int):
        super().__init__()
        self.nhead, self.d_model = nhead, d_model
        self.encoder = custom_encoder
        self.decoder = custom_decoder

    def forward(self, src, tgt=None, mask: MaskArg = None, **kw):
        raise NotImplementedError("the unconditional (encoder-only) model is not used by the reference's "
                                  "Self-attentive / Upsampling priors and is not built")


def _reinit(layer: nn.Module) -> None:
    """Deep copies share initial values; give every copy its own draw (like torch's
    _get_clones users usually re-initialise with xavier)."""
    for m in layer.modules():
        if isinstance(m, _LinearParams):
            nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
            nn.init.uniform_(m.bias, -1 / math.sqrt(m.in_features), 1 / math.sqrt(m.in_features))
        elif isinstance(m, RelativeMultiheadAttention):
            nn.init.xavier_uniform_(m.in_proj_weight)
            if m.rel_embeddings is not None:
                nn.init.normal_(m.rel_embeddings, std=m.head_dim ** -0.5)
