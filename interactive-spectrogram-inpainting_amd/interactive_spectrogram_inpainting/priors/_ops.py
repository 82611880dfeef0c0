"""Python wrappers over the transformer-prior entry points of libisi_hip.so.
Sequences are time-major `[S, B, d]` fp32 like at the reference's call sites
(priors/transformer.py:736-738)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from .. import _hip
from ..vqvae._ops import pack_conv_weight


import os
import weakref


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


# Products of the whole-sequence GEMMs (data and accumulation are fp32): 'bf16x6' (default) = six-term split
# on the bf16 matrix pipe -- x = hi + mid + lo exactly, every term above 2^-24 of a product kept: error against
# fp64 at or below the fp32 pipe's own, 1.4x its speed; 'f32' = the fp32 matrix pipe.
# Shapes the split kernel does not cover (N <= 32 or K < 128) run on the fp32 pipe either way.
# 'f16x3' (default since round 2) = the VQ-VAE path's three-term split-f16 products: fp32-grade (error against fp64 at
# or below the fp32 pipe's) at half the matrix work of 'bf16x6', but limited to f16's RANGE
# (|activation| < 16384, |weight| < 64): a weight matrix beyond it is recognised when it is packed and runs 'bf16x6';
# an activation beyond it shows as Inf / NaN in the output.
LINEAR_PRECISION = os.environ.get("ISI_LINEAR_PRECISION", "f16x3")
# products of the input-gradient GEMMs of the training path (dX = dY W): three terms (relative error ~2^-16 per
# product) are far below the noise the ReLU / dropout masks and the optimiser see
LINEAR_GRAD_PRECISION = os.environ.get("ISI_LINEAR_GRAD_PRECISION", "bf16x3")
# products of the attention contractions (q k^T, q e^T, p v): 'f32' = fp32 matrix pipe, 'bf16x3' = three-term
# split-bf16 on the bf16 pipe (relative error of a product ~2^-16, fp32 accumulation, logits / softmax fp32),
# 'bf16' = single-term bf16 (operands rounded to 8 significand bits: north_star's "MFMA bf16" mode; measured error
# in bench.py's attention leg and in tests/test_prior_gpu.py)
ATTENTION_PRECISION = os.environ.get("ISI_ATTENTION_PRECISION", "bf16x3")
# 'f16' = single-term f16 (operands rounded to 11 significand bits: the same matrix rate as 'bf16' at an eighth of its
# rounding error -- inside north_star's 1e-3; limited to f16's range, |q k v e| < 65504, like every f16 mixed-precision
# attention; its backward runs the three-term kernels)
_ATTN_PREC = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16": 3}
ATTENTION_PRECISIONS = tuple(_ATTN_PREC)                 # modes bench.py times
ATTENTION_TERMS = {"f32": 0, "bf16x3": 3, "bf16": 1, "f16": 1}    # 16-bit MFMA terms per product (0: fp32 matrix pipe)
# training: the forward keeps its logits ([B,H,Sq,Sk rounded up to 32] fp32: 277 MB per attention at B 8, H 8, S 1025) and
# the backward reads them instead of forming Q K^T, the band product Q E^T and its skew twice more -- 60 % of the
# backward's time at that shape; 0 = recompute (no extra memory)
SAVE_ATTENTION_LOGITS = os.environ.get("ISI_ATTN_SAVE_LOGITS", "1") != "0"
# byte cap of ONE kept-logits buffer (B H Sq ceil32(Sk) floats: quadratic in the sequence length, 22 of them alive per step
# of the top prior); an attention beyond it -- or one whose buffer the allocator cannot provide -- recomputes in its backward
ATTENTION_LOGITS_MAX_MB = float(os.environ.get("ISI_ATTN_LOGITS_MAX_MB", "4096"))
_PREC_FLAG = {"f32": 0, "bf16x3": 2, "bf16x6": 4, "f16x3": 8}
_F16_WEIGHT_LIMIT = 63.98   # 65520 / 1024 and above rounds to inf in the f16 pieces


class WeightRange:
    """Whether a weight tensor is inside the operand range of the split-f16 products (|w| < 64), decided without a
    device read-back per optimizer step.  Inference weights are checked exactly whenever their version changes.  A
    weight under training is checked every ~256 versions against HALF the limit: between two checks it would have
    to double to leave the range (an update of lr = 3e-4 moves it by ~1e-4 per step), and if it ever did, the f16
    pieces overflow to Inf and the step's outputs are non-finite -- loud, not silently wrong."""
    PERIOD = 256

    def __init__(self):
        self.ok, self.next_check, self.last_version = False, -1, -1

    def update(self, weight: torch.Tensor, inference: bool) -> bool:
        v = weight._version + _hip.optimizer_steps()      # either counter moves when the values may have changed
        if capturing() and self.next_check >= 0:          # (decided by the eager steps in front of a graph capture)
            return self.ok
        if inference or self.next_check < 0 or v >= self.next_check or v < self.last_version:
            m = float(weight.detach().abs().max()) if weight.numel() else 0.0          # reads one scalar back
            self.ok = m < (_F16_WEIGHT_LIMIT if inference else 0.5 * _F16_WEIGHT_LIMIT)
            self.next_check = -1 if inference else v + self.PERIOD + (id(self) >> 4) % 64   # (staggered over modules)
        self.last_version = v
        return self.ok


def pack_linear_weight(weight: torch.Tensor, range_check=False, with_f16: bool = False) -> torch.Tensor:
    """nn.Linear weight [N,K] -> packed GEMM operand (a 1x1 convolution weight: [N][K padded to 32]).  A dense fp32
    weight whose K is a multiple of 32 IS that operand: it is returned as a view, no copy and no launch (the prior's
    ~120 linears were re-packed every training step: 265 launches of 5 us).  `range_check`: "now" = compare against
    the split-f16 operand range here (reads one scalar back: constant operands, packed once); True / False = the
    caller's knowledge (WeightRange.update); False also stands for an unknown range (the GEMM runs 'bf16x6')."""
    w = weight.detach()
    if with_f16:
        # a weight that stays put (inference): its split-f16 pair copy behind it (ISI_CONV_W16) -- the GEMM kernel then
        # stages the weight tile by plain copies instead of converting it once per 128-row tile
        packed = pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1), with_f16=True)
        packed.isi_w16 = True
    elif (w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float32 and w.shape[1] % 32 == 0
            and w.data_ptr() % 16 == 0 and w.is_cuda):
        packed = w.view(w.shape[0], w.shape[1])          # a fresh tensor object (it carries an attribute below)
    else:
        packed = pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1))
    ok = range_check
    if isinstance(range_check, str):
        if range_check != "now":
            raise ValueError("range_check: True, False or 'now'")
        ok = weight.numel() == 0 or bool(w.abs().max() < _F16_WEIGHT_LIMIT)
    packed.isi_f16_ok = bool(ok) and LINEAR_PRECISION == "f16x3"
    return packed


BATCHED_WT_PACK = os.environ.get("ISI_BATCHED_WT_PACK", "1") != "0"


class _WtPackGroup:
    """The W^T operands of every linear layer under training, re-packed by ONE launch per optimizer step
    (isi_pack_linear_wT_bf16_multi) instead of one launch per weight when its backward first asks (~80 launches of ~5 us per
    step of the top prior).  An entry = (weak reference to the tensor that OWNS the weight's storage -- the Parameter, or
    the Parameter a slice was taken from --, its persistent output buffer, the version it was packed at); a request that
    finds its entry stale re-packs every stale entry of its device in that one launch.  Nothing here keeps a weight alive:
    when the owner is collected the entry (and its 2 N K output floats) goes with the next request or `purge()`."""

    def __init__(self):
        self.entries = {}          # (device index, data_ptr, N, K) -> [weakref of the owner, out, packed version]
        self.tables = {}           # device index -> (keys in table order, device table)

    def _alive(self, e) -> bool:
        return e[0]() is not None

    def purge(self) -> int:
        """Drops the entries whose weight is gone; returns how many are left."""
        dead = [k for k, e in self.entries.items() if not self._alive(e)]
        for k in dead:
            del self.entries[k]
            self.tables.pop(k[0], None)
        return len(self.entries)

    def get(self, weight: torch.Tensor, owner: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`weight`: the Parameter itself or a view of it; `owner`: the long-lived tensor whose death retires the entry
        (default: the view's base, else `weight` itself -- NOT a detached alias: `detach()` has no `_base`, a weak reference
        to it would die with the temporary)."""
        N, K = weight.shape
        dev = weight.device.index or 0
        key = (dev, weight.data_ptr(), N, K)
        e = self.entries.get(key)
        if e is None or not self._alive(e):
            out = torch.empty(2 * N * K, dtype=torch.float32, device=weight.device)
            out.isi_f16_ok = False
            out.isi_w16_bf16 = True
            if owner is None:
                owner = weight._base if weight._base is not None else weight
            e = self.entries[key] = [weakref.ref(owner), out, None]
            self.tables.pop(dev, None)
        stamp = _hip.version_of(weight)
        if e[2] != stamp:
            self._repack(dev, weight.device)
        return e[1]

    def _repack(self, dev: int, device: torch.device) -> None:
        # Under stream capture nothing may be dropped or rebuilt: the table is a host-to-device copy (not capturable), and an
        # entry that died since the warm-up steps -- a model of an earlier test / run collected by the gc.collect() in front
        # of the capture -- used to send the recorded step through exactly that rebuild (flaky NaN gradients of a replayed
        # step).  `prepare_for_capture()` purges and rebuilds in front of the capture; a dead entry met here stays in the
        # table for this launch (its output buffer is still owned by the entry, its weight's memory is still mapped).
        cap = capturing()
        if not cap:
            self.purge()
        hit = self.tables.get(dev)
        if hit is None and cap:
            raise RuntimeError("a linear layer's W^T operand was first requested while a step was being recorded: run the step "
                               "eagerly once (GraphedTrainingStep's warm-up) before recording it")
        if hit is None:
            keys = [k for k in self.entries if k[0] == dev]
            rows = [[k[1], self.entries[k][1].data_ptr(), k[2], k[3]] for k in keys]
            hit = self.tables[dev] = (keys, torch.tensor(rows, dtype=torch.int64).to(device))
        keys, table = hit
        # (every live entry of the device: after an optimizer step they are all stale, and a stale one left out would
        # need its own launch later)
        _hip.check(_hip.lib().isi_pack_linear_wT_bf16_multi(table.data_ptr(), len(keys), 64,
                                                            C.c_void_p(_hip.stream_ptr(device))),
                   "isi_pack_linear_wT_bf16_multi")
        for k in keys:
            e = self.entries.get(k)
            owner = e[0]() if e is not None else None
            if owner is not None:
                e[2] = _hip.version_of(owner)

    def prepare_for_capture(self) -> None:
        """Called right before a step is recorded (after the warm-up steps and the garbage collection in front of the
        capture): dead entries dropped and every device's table rebuilt NOW, so that the recorded step's one re-pack launch
        finds its table and drops nothing."""
        self.purge()
        for dev in sorted({k[0] for k in self.entries}):
            if dev not in self.tables:
                keys = [k for k in self.entries if k[0] == dev]
                rows = [[k[1], self.entries[k][1].data_ptr(), k[2], k[3]] for k in keys]
                device = self.entries[keys[0]][1].device
                self.tables[dev] = (keys, torch.tensor(rows, dtype=torch.int64).to(device))


_WT_GROUP = _WtPackGroup()


def pack_linear_weight_t(weight: torch.Tensor, owner: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear weight [N,K] -> operand of the input-gradient GEMM dX = dY W: W^T as a packed [K][N] weight followed
    by its split-bf16 pair copy (isi_pack_linear_wT_bf16[_multi]: the GEMM kernel stages the weight tile by plain copies);
    other shapes: the transposing copy.  The returned buffer belongs to the weight's entry of `_WT_GROUP` and is rewritten
    by the next re-pack."""
    w = weight.detach()
    N, K = w.shape
    if (N % 32 == 0 and K % 32 == 0 and w.is_contiguous() and w.dtype == torch.float32 and w.is_cuda
            and w.data_ptr() % 16 == 0 and LINEAR_GRAD_PRECISION == "bf16x3"):
        if BATCHED_WT_PACK:
            return _WT_GROUP.get(weight, owner)       # (the group keeps a WEAK reference to the weight's owner)
        out = torch.empty(2 * N * K, dtype=torch.float32, device=w.device)
        _hip.check(_hip.lib().isi_pack_linear_wT_bf16(w.data_ptr(), out.data_ptr(), N, K, _s(w)), "isi_pack_linear_wT_bf16")
        out.isi_f16_ok = False
        out.isi_w16_bf16 = True
        return out
    return pack_linear_weight(w.t().contiguous())


def fused_tails_ok(M: int, N: int, K: int) -> bool:
    """Shapes `isi_linear_f32` takes (the split-product GEMM kernel): where a layer's dropout / scaled gate can ride in the
    GEMM epilogue instead of torch kernels of their own."""
    return (K % 32 == 0 and K >= 128 and N > 32 and M >= 256 and LINEAR_PRECISION in ("f16x3", "bf16x3")
            and LINEAR_GRAD_PRECISION == "bf16x3")


_SEED_BASE = None      # the tensor behind isi_set_dropout_seed_base (kept alive here)


def set_dropout_seed_base(counter: Optional[torch.Tensor]) -> None:
    """A device-resident int64 scalar added to every fused dropout's seed (include/isi_hip.h: isi_set_dropout_seed_base), or
    None.  The owner of a replayed HIP graph advances it between replays (utils/training/graphed_step.py)."""
    global _SEED_BASE
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
        raise ValueError("the seed base is one int64 on the GPU")
    _hip.check(_hip.lib().isi_set_dropout_seed_base(counter.data_ptr() if counter is not None else None),
               "isi_set_dropout_seed_base")
    _SEED_BASE = counter


def capturing() -> bool:
    """Whether the current stream is being captured into a HIP graph: nothing may be read back, checks that would are
    skipped (they ran in the eager warm-up steps in front of the capture)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def dropout_seed() -> int:
    """A fresh 62-bit seed from torch's CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.empty((), dtype=torch.int64).random_(0, 1 << 62).item())


def linear(x: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor], n_out: int,
           relu: bool = False, residual: Optional[torch.Tensor] = None,
           precision: Optional[str] = None, gate: Optional[torch.Tensor] = None, gate_scale: float = 1.0,
           dropout_p: float = 0.0, dropout_seed_: int = 0) -> torch.Tensor:
    """y[..., n_out] = x[..., K] W^T + b (+ residual): the implicit-GEMM convolution kernel with a
    1x1 window (rows = "pixels"); `precision` overrides LINEAR_PRECISION.  `gate` ([..., n_out], contiguous):
    y is zeroed where gate <= 0 (a ReLU's backward mask applied by the input-gradient GEMM's epilogue) and multiplied by
    `gate_scale` elsewhere.  `dropout_p` > 0: inverted dropout of the output inside the GEMM epilogue (isi_linear_f32: the
    mask is a hash of (seed, element index); shapes outside `fused_tails_ok` raise)."""
    _hip.require_gpu(x, "linear input")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    prec = precision or LINEAR_PRECISION
    if prec == "f16x3" and not getattr(packed_w, "isi_f16_ok", False):
        prec = "bf16x6"                      # weights beyond the f16 pieces' range (or of unknown range)
    out = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    s0 = _hip.isi_src(x2.data_ptr(), K, 0, 1, 0, x2.stride(0))
    dst = _hip.isi_dst(out.data_ptr(), 0, 1, 0, n_out)
    res = None
    if residual is not None:
        r2 = residual.reshape(M, n_out)
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
        res = _hip.isi_src(r2.data_ptr(), n_out, 0, 1, 0, r2.stride(0))
    w16 = (16 if (prec == "f16x3" and getattr(packed_w, "isi_w16", False)) else                    # ISI_CONV_W16
           256 if (prec == "bf16x3" and getattr(packed_w, "isi_w16_bf16", False)) else 0)         # ISI_CONV_W16_BF16
    if dropout_p > 0.0 or gate_scale != 1.0:
        a = _hip.isi_linear_args()
        a.x, a.ldx, a.packed_w = x2.data_ptr(), x2.stride(0), packed_w.data_ptr()
        a.bias = bias.data_ptr() if bias is not None else None
        if residual is not None:
            r2 = residual.reshape(M, n_out)
            r2 = r2 if r2.stride(1) == 1 else r2.contiguous()
            a.residual, a.ldr = r2.data_ptr(), r2.stride(0)
        if gate is not None:
            g2 = gate.reshape(M, n_out)
            if not g2.is_contiguous() or g2.dtype != torch.float32 or g2.device != x.device:
                raise ValueError("linear: gate must be a contiguous fp32 tensor of the output's shape on the same device")
            a.gate, a.ldg, a.gate_scale = g2.data_ptr(), n_out, gate_scale
        a.out, a.ldo, a.M, a.N, a.K = out.data_ptr(), n_out, M, n_out, K
        a.flags = int(relu) | _PREC_FLAG[prec] | w16
        a.drop_p, a.drop_seed = float(dropout_p), int(dropout_seed_)
        _hip.check(_hip.lib().isi_linear_f32(C.byref(a), _s(x)), "isi_linear_f32")
        return out.reshape(*x.shape[:-1], n_out)
    if gate is not None:
        g2 = gate.reshape(M, n_out)
        if not g2.is_contiguous() or g2.dtype != torch.float32 or g2.device != x.device:
            raise ValueError("linear: gate must be a contiguous fp32 tensor of the output's shape on the same device")
        rc = _hip.lib().isi_conv2d_gated_f32(C.byref(s0), None, packed_w.data_ptr(),
                                             bias.data_ptr() if bias is not None else None,
                                             C.byref(res) if res is not None else None, g2.data_ptr(), C.byref(dst),
                                             1, 1, M, n_out, 1, 1, 1, 0, int(relu) | _PREC_FLAG[prec] | w16, _s(x))
        _hip.check(rc, "isi_conv2d_gated_f32 (linear)")
        return out.reshape(*x.shape[:-1], n_out)
    rc = _hip.lib().isi_conv2d_f32(C.byref(s0), None, packed_w.data_ptr(),
                                   bias.data_ptr() if bias is not None else None,
                                   C.byref(res) if res is not None else None, C.byref(dst),
                                   1, 1, M, n_out, 1, 1, 1, 0, int(relu) | _PREC_FLAG[prec] | w16, _s(x))
    _hip.check(rc, "isi_conv2d_f32 (linear)")
    return out.reshape(*x.shape[:-1], n_out)


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False,
                residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Single-token path: x [M<=8, K] against the unpacked torch weight [N, K];
    `out` may be a row-strided view (e.g. a slot of a key/value cache)."""
    _hip.require_gpu(x, "linear_rows input")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    rc = _hip.lib().isi_linear_rows_f32(
        x.data_ptr(), x.stride(0), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
        residual.data_ptr() if residual is not None else None,
        residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), M, N, K, int(relu),
        _s(x))
    _hip.check(rc, "isi_linear_rows_f32")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              residual: Optional[torch.Tensor] = None, dropout_p: float = 0.0, dropout_seed_: int = 0) -> torch.Tensor:
    """LayerNorm(drop(x) + residual); `dropout_p` > 0: the inverted dropout of x inside the kernel (mask = hash of (seed,
    element index), isi_layernorm_dropout_f32)."""
    _hip.require_gpu(x, "layernorm input")
    x = x.contiguous()
    D = x.shape[-1]
    out = torch.empty_like(x)
    if residual is not None:
        residual = residual.contiguous()
    if dropout_p > 0.0:
        rc = _hip.lib().isi_layernorm_dropout_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                                  gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), x.numel() // D, D,
                                                  eps, float(dropout_p), int(dropout_seed_), _s(x))
        _hip.check(rc, "isi_layernorm_dropout_f32")
        return out
    rc = _hip.lib().isi_layernorm_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                      gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), x.numel() // D, D,
                                      eps, _s(x))
    _hip.check(rc, "isi_layernorm_f32")
    return out


def _attn_args(q, k, v, rel, out, Sq, Sk, B, H, hd, Cq, Ck, Ek, mask_mode, dense_mask):
    a = _hip.isi_attn_args()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.rel_embeddings = rel.data_ptr() if rel is not None else None
    a.dense_mask = dense_mask.data_ptr() if dense_mask is not None else None
    a.Sq, a.Sk, a.B, a.H, a.head_dim = Sq, Sk, B, H, hd
    a.Cq, a.Ck, a.Ek = Cq, Ck, Ek
    a.rel_rows = rel.shape[1] if rel is not None else 0
    a.mask_mode = mask_mode
    a.scale = 1.0 / math.sqrt(hd)
    a.precision = _ATTN_PREC[ATTENTION_PRECISION]
    return a


def attention_logits_buffer(B: int, nhead: int, Sq: int, Sk: int, device) -> Optional[torch.Tensor]:
    """[B,H,Sq,ld] buffer for the logits a training forward keeps (None when they are recomputed: exact-fp32 mode,
    ISI_ATTN_SAVE_LOGITS=0, or the round-3 forward kernels selected)."""
    if not SAVE_ATTENTION_LOGITS or ATTENTION_PRECISION == "f32":
        return None
    old = C.c_int()
    _hip.check(_hip.lib().isi_knob_get(b"ISI_ATTN_OLD_FWD", C.byref(old)), "isi_knob_get")
    if old.value and ATTENTION_PRECISION != "f16":
        return None
    ld = (Sk + 31) // 32 * 32
    if B * nhead * Sq * ld * 4 > ATTENTION_LOGITS_MAX_MB * (1 << 20):
        return None                      # beyond the byte cap: this attention's backward recomputes
    try:
        return torch.empty(B, nhead, Sq, ld, dtype=torch.float32, device=device)
    except torch.cuda.OutOfMemoryError:
        return None                      # no room for the logits: recompute instead of failing the step


def rel_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel: Optional[torch.Tensor], nhead: int,
                  Cq: int, Ck: int, Ek: int, mask_mode: int = 0,
                  dense_mask: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                  logits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [Sq,B,d], k/v [Sk,B,d] (any row strides, last dim contiguous; views into a
    fused qkv buffer are consumed in place) -> [Sq,B,d]."""
    _hip.require_gpu(q, "attention input")
    Sq, B, d = q.shape
    Sk = k.shape[0]
    hd = d // nhead
    out = torch.empty(Sq, B, d, dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, v, rel, out, Sq, Sk, B, nhead, hd, Cq, Ck, Ek, mask_mode, dense_mask)
    a.q_ss, a.q_sb, a.q_sh = q.stride(0), q.stride(1), hd
    a.k_ss, a.k_sb, a.k_sh = k.stride(0), k.stride(1), hd
    a.v_ss, a.v_sb, a.v_sh = v.stride(0), v.stride(1), hd
    a.o_ss, a.o_sb, a.o_sh = out.stride(0), out.stride(1), hd
    if lse is not None:  # [B,H,Sq] log-sum-exp per query, kept for the backward
        a.lse = lse.data_ptr()
    if logits is not None:  # [B,H,Sq,ld] logits of the allowed pairs, kept for the backward (attention_logits_buffer)
        a.logits, a.logits_ld = logits.data_ptr(), logits.stride(2)
    L = _hip.lib()
    ws_bytes = L.isi_rel_attention_workspace_bytes(C.byref(a))
    if ws_bytes:    # 16-bit planes of K, V and the table for the LDS-DMA-staged kernel (rel_attention_fwd3.hip)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    _hip.check(L.isi_rel_attention_f32(C.byref(a), _s(q)), "isi_rel_attention_f32")
    return out


def rel_attention_decode(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel: Optional[torch.Tensor],
                         nhead: int, n_keys: int, q_pos: int, Cq: int, Ck: int, Ek: int) -> torch.Tensor:
    """q [B,d] (one position), k/v caches [S_max,B,d]; attends to keys 0..n_keys-1."""
    _hip.require_gpu(q, "attention input")
    B, d = q.shape
    hd = d // nhead
    out = torch.empty(B, d, dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, v, rel, out, 1, n_keys, B, nhead, hd, Cq, Ck, Ek, 0, None)
    a.q_ss, a.q_sb, a.q_sh = 0, q.stride(0), hd
    a.k_ss, a.k_sb, a.k_sh = k.stride(0), k.stride(1), hd
    a.v_ss, a.v_sb, a.v_sh = v.stride(0), v.stride(1), hd
    a.o_ss, a.o_sb, a.o_sh = 0, out.stride(0), hd
    L = _hip.lib()
    ws = torch.empty(L.isi_rel_attention_decode_workspace_floats(B, nhead, hd), dtype=torch.float32,
                     device=q.device)
    _hip.check(L.isi_rel_attention_decode_f32(C.byref(a), q_pos, ws.data_ptr(), _s(q)),
               "isi_rel_attention_decode_f32")
    return out


def sample_rows(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, u: torch.Tensor,
                return_filtered: bool = False):
    """logits [rows, n] -> int64 [rows] (and the filtered logits when asked)."""
    _hip.require_gpu(logits, "logits")
    rows, n = logits.shape
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    filt = torch.empty(rows, n, dtype=torch.float32, device=logits.device) if return_filtered else None
    u = u.to(device=logits.device, dtype=torch.float32).contiguous()
    rc = _hip.lib().isi_sample_row_f32(logits.data_ptr(), logits.stride(0), rows, n, float(temperature),
                                       int(top_k), float(top_p), u.data_ptr(), out.data_ptr(),
                                       filt.data_ptr() if filt is not None else None, _s(logits))
    _hip.check(rc, "isi_sample_row_f32")
    return (out, filt) if return_filtered else out
