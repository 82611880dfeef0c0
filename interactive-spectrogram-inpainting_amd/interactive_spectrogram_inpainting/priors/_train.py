"""Differentiable building blocks of the transformer prior on MI355X.

The reference trains the prior with `loss.backward()` through the absent package's
layers (train_autoregressive_model.py:203-257).  Here every heavy operator is a
`torch.autograd.Function` whose forward AND backward are entry points of
libisi_hip.so; torch autograd only chains them and handles the small index / concat
plumbing of `VQNSynthTransformer.prepare_data` (priors/transformer.py:513-680).

  Linear        y = x W^T + b (+relu) (+residual)   fwd: isi_conv2d_f32 (1x1 = GEMM)
                                                    dX : the same GEMM with W^T packed
                                                    dW, db: isi_conv_wgrad_f32 (row-reduction GEMM)
  LayerNorm     LN(x + residual)                    isi_layernorm_f32 / isi_layernorm_bwd_f32
  RelAttention  softmax((q.k + q.e[r]) * s + mask)v isi_rel_attention_f32 (+LSE) / isi_rel_attention_bwd_f32
  LabelSmoothingLoss                                isi_label_smoothing_loss_f32 (loss and gradient in one pass)

There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from .. import _hip
from . import _ops


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


def _rows(t: torch.Tensor) -> torch.Tensor:
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.is_contiguous() else t2.contiguous()


def _wgrad_flags() -> int:
    from ..vqvae._train import WGRAD_FLAGS   # one switch for both models (ISI_WGRAD_PRECISION)
    return WGRAD_FLAGS


def linear_wgrad(x2: torch.Tensor, dy2: torch.Tensor, out_w: Optional[torch.Tensor] = None,
                 out_b: Optional[torch.Tensor] = None):
    """x2 [M,K], dy2 [M,N] dense -> (dW [N,K] torch layout, db [N]).  `out_w` / `out_b`: contiguous destinations (rows of
    a larger gradient tensor), K a multiple of 32."""
    M, K = x2.shape
    N = dy2.shape[1]
    L = _hip.lib()
    Kpad = (K + 31) // 32 * 32
    nws = L.isi_conv_wgrad_workspace_floats(N, K, M, 1)
    ws = torch.empty(nws, dtype=torch.float32, device=x2.device)
    if out_w is not None:
        assert Kpad == K and out_w.shape == (N, K) and out_w.is_contiguous() and out_b.shape == (N,) and out_b.is_contiguous()
    packed = out_w if out_w is not None else torch.empty(N, Kpad, dtype=torch.float32, device=x2.device)
    db = out_b if out_b is not None else torch.empty(N, dtype=torch.float32, device=x2.device)
    s0 = _hip.isi_src(x2.data_ptr(), K, 0, 1, 0, x2.stride(0))
    rc = L.isi_conv_wgrad_f32(C.byref(s0), None, dy2.data_ptr(), packed.data_ptr(), db.data_ptr(), ws.data_ptr(), nws,
                              1, 1, M, N, 1, 1, 1, 0, _wgrad_flags(), _s(x2))
    _hip.check(rc, "isi_conv_wgrad_f32 (linear)")
    return (packed if Kpad == K else packed[:, :K]), db


def _grad_precision() -> str:
    # an exact forward ('f32') keeps exact gradients; otherwise the input-gradient GEMMs use their own default
    return "f32" if _ops.LINEAR_PRECISION == "f32" else _ops.LINEAR_GRAD_PRECISION


class LinearFn(torch.autograd.Function):
    """y[..., N] = x[..., K] W^T + b, optionally rectified, optionally + residual (added before the ReLU)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu: bool, packed, packed_t_fn, rectified_input: bool = False,
                grad_pre_gated: bool = False, dropout_p: float = 0.0, input_keep_scale: float = 1.0, carry: bool = False):
        """`rectified_input`: x is the (possibly dropout-scaled) output of a ReLU -- the input gradient is then zeroed
        where x <= 0 by the GEMM's epilogue, i.e. that ReLU's backward mask is applied HERE; the producing layer is
        built with `grad_pre_gated` and skips its own mask pass (a clone and a three-pass kernel over [M, 2048] per
        feed-forward block).  Where a unit was dropped x is 0 and the gradient is 0 either way.
        `dropout_p` (a rectified, pre-gated layer only): inverted dropout of the output inside the GEMM epilogue -- no mask
        tensor, no dropout kernels; the consumer is built with `input_keep_scale` = 1 / (1 - p), which its input-gradient
        GEMM applies together with the gate (what the dropout's backward would have done: kept units scaled, dropped
        units -- zeros of the rectified tensor -- gated to zero).
        `carry`: also return x itself.  A residual branch reads its input twice -- this layer and the residual add of the
        LayerNorm behind the branch -- and autograd would sum the two gradients with a kernel of its own (42 adds of
        [S, B, d] per training step of the top prior); handed on through this node, the residual path's gradient arrives
        HERE and is added by the input-gradient GEMM's epilogue."""
        ctx.carry = carry
        if carry:
            ctx.set_materialize_grads(False)
        if dropout_p > 0.0:
            assert relu and grad_pre_gated, "fused dropout goes with a rectified output whose consumer gates the gradient"
            y = _ops.linear(x, packed, bias, weight.shape[0], relu=relu, residual=residual, dropout_p=dropout_p,
                            dropout_seed_=_ops.dropout_seed())
        else:
            y = _ops.linear(x, packed, bias, weight.shape[0], relu=relu, residual=residual)
        ctx.input_keep_scale = input_keep_scale
        ctx.relu, ctx.packed_t_fn = relu, packed_t_fn
        ctx.rectified_input, ctx.grad_pre_gated = rectified_input, grad_pre_gated
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return (y, x) if carry else y

    @staticmethod
    def backward(ctx, dy, dcarry=None):
        x, weight, y = ctx.saved_tensors
        N, K = weight.shape
        dy2 = _rows(dy)
        if ctx.relu and not ctx.grad_pre_gated:
            if dy2.data_ptr() == dy.data_ptr():
                dy2 = dy2.clone()
            _hip.check(_hip.lib().isi_relu_bwd_f32(dy2.data_ptr(), _rows(y).data_ptr(), dy2.numel(), _s(dy2)),
                       "isi_relu_bwd_f32")
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            gate = _rows(x) if ctx.rectified_input else None
            dx = _ops.linear(dy2, ctx.packed_t_fn(), None, K, precision=_grad_precision(), gate=gate,
                             gate_scale=ctx.input_keep_scale if gate is not None else 1.0,
                             residual=_rows(dcarry) if dcarry is not None else None).reshape(x.shape)
        elif dcarry is not None:
            dx = dcarry
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = linear_wgrad(_rows(x), dy2)
            if not ctx.has_bias:
                db = None
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy2.reshape(dy.shape)
        return dx, dw, db, dres, None, None, None, None, None, None, None, None


class CrossInProjFn(torch.autograd.Function):
    """The two in-projections of a cross-attention from ONE fused parameter: q = x W[:d]^T + b[:d], k|v = mem W[d:]^T + b[d:].
    As two LinearFn nodes on slices of the parameter, autograd materialised each slice's gradient as a zero-filled
    [3d, d] tensor (+ bias), filled the slice and added the two: ~12 small launches per layer and step.  Here both weight
    gradients are written into the rows of one [3d, d] tensor that is returned as the parameter's gradient."""

    @staticmethod
    def forward(ctx, x, mem, weight, bias, packed_q, packed_kv, packed_t_q_fn, packed_t_kv_fn, carry: bool = False):
        d = weight.shape[1]
        q = _ops.linear(x, packed_q, bias[:d], d)
        kv = _ops.linear(mem, packed_kv, bias[d:], 2 * d)
        ctx.fns = (packed_t_q_fn, packed_t_kv_fn)
        ctx.save_for_backward(x, mem, weight)
        if carry:        # (LinearFn: the residual path's gradient of x arrives here)
            ctx.set_materialize_grads(False)
            return q, kv, x
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv, dcarry=None):
        x, mem, weight = ctx.saved_tensors
        d = weight.shape[1]
        dq2, dkv2 = _rows(dq), _rows(dkv)
        dx = dmem = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _ops.linear(dq2, ctx.fns[0](), None, d, precision=_grad_precision(),
                             residual=_rows(dcarry) if dcarry is not None else None).reshape(x.shape)
        elif dcarry is not None:
            dx = dcarry
        if ctx.needs_input_grad[1]:
            dmem = _ops.linear(dkv2, ctx.fns[1](), None, d, precision=_grad_precision()).reshape(mem.shape)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dw = torch.empty_like(weight)
            db = torch.empty(3 * d, dtype=torch.float32, device=weight.device)
            linear_wgrad(_rows(x), dq2, out_w=dw[:d], out_b=db[:d])
            linear_wgrad(_rows(mem), dkv2, out_w=dw[d:], out_b=db[d:])
        return dx, dmem, dw, db, None, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """LayerNorm(drop(x) + residual) * gamma + beta over the last dimension.  `dropout_p` > 0 (train mode): the inverted
    dropout of x inside the LayerNorm kernels (mask = hash of (seed, index): no mask tensor, no dropout launches); the
    backward returns the masked, rescaled gradient for x and the plain one for the residual."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps: float, dropout_p: float = 0.0):
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        ctx.drop = (float(dropout_p), _ops.dropout_seed() if dropout_p > 0.0 else 0)
        y = _ops.layernorm(x, gamma, beta, eps, residual=residual, dropout_p=ctx.drop[0], dropout_seed_=ctx.drop[1])
        ctx.eps = eps
        ctx.save_for_backward(x, residual, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, residual, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        D = x.shape[-1]
        M = x.numel() // D
        L = _hip.lib()
        dz = torch.empty_like(x)
        dg = torch.empty(D, dtype=torch.float32, device=x.device)
        db = torch.empty(D, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.isi_layernorm_bwd_workspace_floats(M, D), dtype=torch.float32, device=x.device)
        p, seed = ctx.drop
        if p > 0.0:
            dx = torch.empty_like(x)
            rc = L.isi_layernorm_dropout_bwd_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                                 gamma.data_ptr(), dy.data_ptr(), dz.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                                 db.data_ptr(), ws.data_ptr(), M, D, ctx.eps, p, seed, _s(x))
            _hip.check(rc, "isi_layernorm_dropout_bwd_f32")
            return dx, (dz if residual is not None else None), dg, db, None, None
        rc = L.isi_layernorm_bwd_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                     gamma.data_ptr(), dy.data_ptr(), dz.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                     ws.data_ptr(), M, D, ctx.eps, _s(x))
        _hip.check(rc, "isi_layernorm_bwd_f32")
        return dz, (dz if residual is not None else None), dg, db, None, None


class RelAttentionFn(torch.autograd.Function):
    """Attention over fused projection buffers, consumed and differentiated in place:
    self-attention  a = qkv [S,B,3d], b = None;  cross-attention  a = q [Sq,B,d], b = k|v [Sk,B,2d];
    rel [H,R,hd] or None -> [Sq,B,d].  The gradient of a (and b) is ONE buffer whose q / k / v
    slices the backward kernels write directly."""

    @staticmethod
    def _split(a, b):
        if b is None:
            d = a.shape[-1] // 3
            return a[..., :d], a[..., d:2 * d], a[..., 2 * d:]
        d = a.shape[-1]
        return a, b[..., :d], b[..., d:]

    workspace_fill = None  # tests: value the backward's workspace (G, partial sums) is filled with before the call

    @classmethod
    def apply(cls, *args):
        # inside `forward` grad mode is always off: the caller's mode travels as an argument (not as class state: two
        # threads, or a no_grad call between another call's apply and forward, would read each other's)
        return super().apply(*args, torch.is_grad_enabled())

    @staticmethod
    def forward(ctx, a, b, rel, nhead, Cq, Ck, Ek, mask_mode, dense_mask, grad_mode=True):
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        q, k, v = RelAttentionFn._split(a, b)
        Sq, B, _ = q.shape
        lse = torch.empty(B, nhead, Sq, dtype=torch.float32, device=q.device)
        # kept only when a backward will follow (under no_grad nothing needs a gradient)
        keep = grad_mode and any(ctx.needs_input_grad[:3])
        logits = _ops.attention_logits_buffer(B, nhead, Sq, k.shape[0], q.device) if keep else None
        out = _ops.rel_attention(q, k, v, rel, nhead, Cq, Ck, Ek, mask_mode=mask_mode, dense_mask=dense_mask,
                                 lse=lse, logits=logits)
        ctx.cfg = (nhead, Cq, Ck, Ek, mask_mode, _ops.ATTENTION_PRECISION)
        ctx.save_for_backward(a, b, rel, out, lse, dense_mask, logits)
        return out

    @staticmethod
    def backward(ctx, dout):
        a_in, b_in, rel, out, lse, dense_mask, logits = ctx.saved_tensors
        nhead, Cq, Ck, Ek, mask_mode, precision = ctx.cfg
        q, k, v = RelAttentionFn._split(a_in, b_in)
        Sq, B, d = q.shape
        Sk = k.shape[0]
        hd = d // nhead
        dout = dout.contiguous()
        L = _hip.lib()
        da = torch.empty_like(a_in)
        db = torch.empty_like(b_in) if b_in is not None else None
        dq, dk, dv = RelAttentionFn._split(da, db)
        drel = torch.empty_like(rel) if rel is not None else None
        a = _hip.isi_attn_bwd_args()
        a.fwd = _ops._attn_args(q, k, v, rel, out, Sq, Sk, B, nhead, hd, Cq, Ck, Ek, mask_mode, dense_mask)
        f = a.fwd
        f.q_ss, f.q_sb, f.q_sh = q.stride(0), q.stride(1), hd
        f.k_ss, f.k_sb, f.k_sh = k.stride(0), k.stride(1), hd
        f.v_ss, f.v_sb, f.v_sh = v.stride(0), v.stride(1), hd
        f.o_ss, f.o_sb, f.o_sh = out.stride(0), out.stride(1), hd
        f.lse = lse.data_ptr()
        f.precision = _ops._ATTN_PREC[precision]       # the mode the forward ran (and wrote its kept logits) in
        if logits is not None and f.precision >= 1:      # (a backward in the exact-fp32 mode recomputes)
            f.logits, f.logits_ld = logits.data_ptr(), logits.stride(2)
        a.d_out = dout.data_ptr()
        a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.d_rel = drel.data_ptr() if drel is not None else None
        nws = L.isi_rel_attention_bwd_workspace_floats(C.byref(a.fwd))
        ws = torch.empty(nws, dtype=torch.float32, device=q.device)
        if RelAttentionFn.workspace_fill is not None:      # tests: poison what the kernels must not read before writing
            ws.fill_(RelAttentionFn.workspace_fill)
        a.workspace, a.workspace_floats = ws.data_ptr(), nws
        _hip.check(L.isi_rel_attention_bwd_f32(C.byref(a), _s(q)), "isi_rel_attention_bwd_f32")
        return da, db, drel, None, None, None, None, None, None, None


class EmbeddingRowsFn(torch.autograd.Function):
    """table[idx] with a dense, deterministic table gradient (nn.Embedding's backward):
    dTable = onehot(idx)^T dy, the one-hot GEMM of the codebook statistics
    (isi_vq_embed_sum_f32: the one-hot operand is generated on the fly, never stored)."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.rows = table.shape[0]
        return table[idx]

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        D = dy.shape[-1]
        dy2 = _rows(dy)
        N = dy2.shape[0]
        Vp = (ctx.rows + 31) // 32 * 32
        L = _hip.lib()
        out = torch.empty(D, Vp, dtype=torch.float32, device=dy.device)
        nws = L.isi_vq_embed_sum_workspace_floats(D, Vp, N)
        ws = torch.empty(nws, dtype=torch.float32, device=dy.device)
        rc = L.isi_vq_embed_sum_f32(dy2.data_ptr(), idx.reshape(-1).contiguous().data_ptr(), out.data_ptr(),
                                    ws.data_ptr(), nws, N, D, Vp, _s(dy))
        _hip.check(rc, "isi_vq_embed_sum_f32 (embedding backward)")
        return out.t()[:ctx.rows], None


class LabelSmoothingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2, target1, num_classes: int, smoothing: float):
        M, K = logits2.shape
        row_loss = torch.empty(M, dtype=torch.float32, device=logits2.device)
        grad = torch.empty_like(logits2) if logits2.requires_grad else None
        rc = _hip.lib().isi_label_smoothing_loss_f32(
            logits2.data_ptr(), target1.data_ptr(), row_loss.data_ptr(),
            grad.data_ptr() if grad is not None else None, M, K, num_classes, smoothing, 1.0 / M, _s(logits2))
        _hip.check(rc, "isi_label_smoothing_loss_f32")
        ctx.save_for_backward(grad)
        return row_loss.mean()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def label_smoothing_loss(pred: torch.Tensor, target: torch.Tensor, num_classes: int, smoothing: float,
                         dim: int = 1) -> torch.Tensor:
    """`LabelSmoothingLoss(num_classes, smoothing, dim)(pred, target)` (utils/losses/prediction.py:14-20):
    class scores along `dim` of pred, int64 targets of pred's shape without that dim."""
    _hip.require_gpu(pred, "logits")
    dim = dim % pred.dim()
    logits2 = pred.movedim(dim, -1)
    K = logits2.shape[-1]
    logits2 = logits2.reshape(-1, K)
    if not logits2.is_contiguous():
        logits2 = logits2.contiguous()
    target1 = target.reshape(-1).contiguous()
    if target1.numel() != logits2.shape[0]:
        raise RuntimeError(f"target of shape {tuple(target.shape)} for predictions of shape {tuple(pred.shape)}")
    return LabelSmoothingFn.apply(logits2, target1, int(num_classes), float(smoothing))
