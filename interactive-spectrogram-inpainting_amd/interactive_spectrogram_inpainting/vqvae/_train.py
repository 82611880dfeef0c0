"""Training step of the VQ-VAE on MI355X: train-mode forward that keeps what the
backward needs, the EMA codebook update, and a hand-written backward.

Replaces what the reference gets from autograd + DDP in `train_vqvae.train`
(train_vqvae.py:168-192): `out, latent_loss, ... = model(img)` stays differentiable
for the caller's `loss = criterion(out, img) + 0.25 * latent_loss.mean();
loss.backward()` through one `torch.autograd.Function` whose backward enqueues

  * input gradients  = the forward convolution kernels with re-laid-out weights
  * weight gradients = `isi_conv_wgrad_f32` (implicit GEMM over pixels)
  * bias gradients   = column sums, ReLU masks, the straight-through / commitment
                       gradient of the quantiser (bottleneck.py:94-95)

Every ReLU of the forward is folded into its producer, so the tape stores
rectified outputs `y` and the backward masks with `y > 0`.

Data parallelism (one process per GPU, torch.distributed "nccl" = RCCL): the
EMA statistics (counts, embed_sum) are all-reduced before the codebook update so
all ranks keep identical codebooks, and parameter gradients live in one flat
buffer whose buckets are all-reduced asynchronously as soon as the backward has
produced them (decoder first), overlapping the remaining backward.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import _hip
from ..utils.training.graphed_step import host_boundary
from . import _ops
from .encoder_decoder import RosinalityDecoder, RosinalityEncoder, RosinalityResBlock, _ConvParams


import os


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


# Products of the training step's forward and input-gradient convolutions: 2 = six-term split-bf16
# (fp32-grade; default), 0 = fp32 matrix pipe.  Weight gradients always use the fp32 pipe.
TRAIN_PRECISION = {"f32": 0, "bf16x6": 2}[os.environ.get("ISI_TRAIN_PRECISION", "bf16x6")]
# Products of the training step's FORWARD convolutions: 'f16x3' = the eval path's three-term split-f16 products
# (fp32-grade, half the matrix work of the six-term bf16 split; operands are activations and weights, whose range the
# f16 pieces cover -- gradients are not: the input-gradient convolutions use DGRAD_PRECISION below), reading the weights'
# pieces prepared at pack time (ISI_CONV_W16: no per-tile weight conversion in the kernel, 9.3 -> 9.0 ms per step);
# 'same' = TRAIN_PRECISION.
FWD_PRECISION = {"same": TRAIN_PRECISION, "f16x3": 4 if TRAIN_PRECISION else 0}[os.environ.get("ISI_TRAIN_FWD_PRECISION", "f16x3")]
# Products of the INPUT-gradient convolutions (dX = dY * W^T): three-term split-bf16 by default, like the weight
# gradients and the prior's input-gradient GEMMs (priors/_ops.py LINEAR_GRAD_PRECISION): a relative error of ~2^-16
# per product, measured against the fp64 gradients in tests/test_train_gpu.py (bar 2e-4), at half the matrix work of
# the six-term split ('same' = TRAIN_PRECISION).  bf16 pieces: gradients are not range-limited like the f16 split.
DGRAD_PRECISION = {"same": TRAIN_PRECISION, "bf16x3": 1 if TRAIN_PRECISION else 0}[os.environ.get("ISI_TRAIN_DGRAD_PRECISION", "bf16x3")]
# Products of the weight-gradient GEMMs (flag bits of isi_conv_wgrad_f32's `transposed` word): three-term split by
# default (relative error ~4e-6 of the gradient's maximum, far below the step-to-step noise of training and 50x
# inside the parity tests' 2e-4), 'bf16x6' = fp32-grade, 'f32' = fp32 matrix pipe.
FORCE_COLLECTIVES = os.environ.get("ISI_FORCE_COLLECTIVES", "0") == "1"
WGRAD_FLAGS = {"f32": 0, "bf16x3": 2, "bf16x6": 4}[os.environ.get("ISI_WGRAD_PRECISION", "bf16x3")]


class PairOnly:
    """A tape entry the training forward kept in the split-f16 pair format ONLY (no fp32 twin was written): every
    backward consumer of such a tensor reads pairs -- the halo-staged weight-gradient kernel as its source operand
    (ISI_CONV_IN*_PAIR), the input-gradient convolutions as their ReLU mask (ISI_CONV_GATE_PAIR).  `pair`: [B,C,H,W] view of
    dense channels-last pair storage."""
    __slots__ = ("pair",)

    def __init__(self, pair: torch.Tensor):
        self.pair = pair

    @property
    def shape(self):
        return self.pair.shape

    @property
    def device(self):
        return self.pair.device


GATE_PAIR = 512       # ISI_CONV_GATE_PAIR


def _gate_of(t):
    """(dense NHWC gate tensor, extra flag) of a tape entry used as the ReLU mask of an input-gradient convolution."""
    if t is None:
        return None, 0
    if isinstance(t, PairOnly):
        return t.pair.permute(0, 2, 3, 1), GATE_PAIR
    return _nhwc(t), 0


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W]-shaped view -> dense channels-last storage (copy only if needed)."""
    p = t.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


def _as_bchw(nhwc: torch.Tensor) -> torch.Tensor:
    return nhwc.permute(0, 3, 1, 2)


def relu_bwd_(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dy, y: dense tensors of identical layout; dy *= (y > 0) in place."""
    _hip.check(_hip.lib().isi_relu_bwd_f32(dy.data_ptr(), y.data_ptr(), dy.numel(), _s(dy)), "isi_relu_bwd_f32")
    return dy


def axpy_(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    _hip.check(_hip.lib().isi_axpy_f32(a.data_ptr(), b.data_ptr(), alpha, a.numel(), _s(a)), "isi_axpy_f32")
    return a


def _rows(t_bchw: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] view -> NHWC-shaped tensor whose pixels are uniformly strided rows of C contiguous channels: dense
    channels-last storage, or a CHANNEL SLICE of it (the two halves of a concatenated input's gradient) -- the kernels
    that take a row stride read such a slice in place; anything else is copied dense."""
    p = t_bchw.permute(0, 2, 3, 1)
    B, H, W, Cc = p.shape
    sb, sh, sw, sc = p.stride()
    if sc == 1 and (W == 1 or sh == W * sw) and (B == 1 or sb == H * sh) and sw % 4 == 0 and p.data_ptr() % 16 == 0:
        return p
    return p.contiguous()


def _rows2d(p_nhwc: torch.Tensor) -> torch.Tensor:
    """The [M, C] matrix of a `_rows` tensor (row stride = its pixel stride)."""
    B, H, W, Cc = p_nhwc.shape
    return p_nhwc.as_strided((B * H * W, Cc), (p_nhwc.stride(2), 1), p_nhwc.storage_offset())


def add_gate_rows(a_nhwc: torch.Tensor, b_nhwc: Optional[torch.Tensor], y_nhwc: Optional[torch.Tensor]) -> torch.Tensor:
    """(y > 0) * (a + b) in one pass; a: `_rows` tensor (possibly a channel slice), b / y dense or None."""
    B, H, W, Cc = a_nhwc.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=a_nhwc.device)
    for t in (b_nhwc, y_nhwc):
        if t is not None and not t.is_contiguous():
            raise ValueError("add_gate_rows: the second term and the gate must be dense channels-last")
    _hip.check(_hip.lib().isi_add_gate_rows_f32(out.data_ptr(), a_nhwc.data_ptr(), a_nhwc.stride(2),
                                                b_nhwc.data_ptr() if b_nhwc is not None else None,
                                                y_nhwc.data_ptr() if y_nhwc is not None else None, B * H * W, Cc, _s(out)),
               "isi_add_gate_rows_f32")
    return out


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M, C] (row stride arbitrary, unit column stride) -> [C] (`out`: written in place, e.g. a bias-gradient slot)."""
    M, Cc = x2d.shape
    L = _hip.lib()
    ws = torch.empty(L.isi_colsum_num_partials(M) * Cc, dtype=torch.float32, device=x2d.device)
    if out is None or not out.is_contiguous():
        out = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    _hip.check(L.isi_colsum_f32(x2d.data_ptr(), x2d.stride(0), out.data_ptr(), ws.data_ptr(), M, Cc, _s(x2d)),
               "isi_colsum_f32")
    return out


DEFER_REDUCTIONS = os.environ.get("ISI_TRAIN_DEFER_REDUCTIONS", "0") != "0"      # measured: no gain; opt-in
# partial sums waiting for their reduction, in MB, beyond which the jobs collected so far are run: deferring ALL of a step's
# reductions to its end (0.7 GB of partials at B = 64) lost what the fewer launches gained -- by then the partials have left
# the 256 MB last-level cache they were written through and come back from HBM (measured: 8.04 vs 8.00 ms per step)
DEFER_LIMIT_MB = float(os.environ.get("ISI_TRAIN_DEFER_LIMIT_MB", "96"))


class ReduceJobs:
    """Split reductions of the weight-gradient GEMMs that have not been launched yet (isi_conv_wgrad_deferred_f32): the
    jobs, and the workspaces their partial sums sit in.  `flush()` runs them all in one launch per 48 jobs; it is called
    before anything reads the gradients -- a data-parallel bucket's all-reduce, the end of the backward."""

    def __init__(self):
        self.jobs, self.keep, self.stream, self.bytes = [], [], None, 0

    def add(self, jobs, n, workspace):
        self.jobs.extend(jobs[i] for i in range(n))
        self.keep.append(workspace)
        self.stream = _s(workspace)
        self.bytes += workspace.numel() * 4
        if self.bytes > DEFER_LIMIT_MB * (1 << 20):
            self.flush()

    def flush(self) -> None:
        if not self.jobs:
            return
        arr = (_hip.isi_reduce_job * len(self.jobs))()
        for i, j in enumerate(self.jobs):
            C.memmove(C.byref(arr, i * C.sizeof(_hip.isi_reduce_job)), C.byref(j), C.sizeof(_hip.isi_reduce_job))
        _hip.check(_hip.lib().isi_reduce_jobs_f32(arr, len(self.jobs), self.stream), "isi_reduce_jobs_f32")
        self.jobs, self.keep, self.bytes = [], [], 0


def conv_wgrad(layer: _ConvParams, x: torch.Tensor, dy_nhwc: torch.Tensor,
               x2: Optional[torch.Tensor] = None,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               defer: Optional[ReduceJobs] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(weight gradient in torch layout, bias gradient).  x (and x2): layer input(s) as
    [B,C,H,W] views; dy_nhwc: dense [B,OH,OW,Cout].  `out` = (weight-shaped, bias-shaped) dense tensors that receive
    the gradients directly (the parameters' slots of the flat gradient buffer: the split reduction writes torch's
    layout itself, no packed intermediate and no permuting copy; ungrouped layers).

    A transposed convolution is the adjoint of the stride-2 convolution with the SAME
    weight tensor ([Cin_T, Cout_T, 4, 4] read as [out, in, 4, 4]), so its weight gradient
    is that convolution's with the roles swapped: "input" = dy, "output gradient" = x.
    No 4-phase decomposition, and for the 2-channel last layer the GEMM becomes
    [Cin x 16*Cout] instead of a [Cout = 2 x ...] sliver."""
    L = _hip.lib()
    tr = layer.transposed
    k = layer.kernel_size
    pair_flags = 0
    if isinstance(x, PairOnly) or isinstance(x2, PairOnly):
        if tr or out is None or layer.groups != 1:
            raise ValueError("pair-format sources go to the halo-staged weight-gradient kernel (plain ungrouped convolutions, "
                             "gradients written in place)")
        if isinstance(x, PairOnly):
            x, pair_flags = x.pair, pair_flags | _ops.PAIR_IN0
        if isinstance(x2, PairOnly):
            x2, pair_flags = x2.pair, pair_flags | _ops.PAIR_IN1
    if tr:
        # roles swapped: dy is the SOURCE operand here (any strides: a channel slice of a wider gradient is read in place)
        src, grad = _as_bchw(dy_nhwc), _nhwc(x)
        rows, cin_role = layer.in_channels, layer.out_channels
    else:
        src, grad = x, dy_nhwc
        rows, cin_role = layer.out_channels, layer.in_channels
    B, _, H, W = src.shape
    cin_true = cin_role
    if cin_role < 4 and x2 is None:
        # 2-channel spectrogram side (first encoder / last decoder layer): zero-pad the channels to 4 in a
        # channels-last copy, so that the vectorised split-product kernel applies (K = 16 * 4 instead of a
        # scalar-gather fp32 launch: 630-700 us -> ~100 us); the padded taps are dropped from dW below
        src4 = torch.empty(B, H, W, 4, dtype=torch.float32, device=src.device)
        sv = _hip.src_nchw_view(src)
        _hip.check(L.isi_pad_channels4_f32(C.byref(sv), src4.data_ptr(), B, H, W, _s(src)), "isi_pad_channels4_f32")
        src, cin_role = src4.permute(0, 3, 1, 2), 4
    K = k * k * cin_role
    Kpad = (K + 31) // 32 * 32
    OH = (H + 2 * layer.padding - k) // layer.stride + 1
    OW = (W + 2 * layer.padding - k) // layer.stride + 1
    M = B * OH * OW
    nws = L.isi_conv_wgrad_workspace_floats(rows, K, M, 1)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    s0 = _hip.src_nchw_view(src)
    s1 = _hip.src_nchw_view(x2) if (x2 is not None and not tr) else None
    if out is not None and layer.groups == 1 and out[0].is_contiguous() and out[1].is_contiguous():
        dw, db = out
        assert dw.shape == layer.weight.shape and db.shape == layer.bias.shape
        if defer is not None:
            jobs, n_jobs = (_hip.isi_reduce_job * 4)(), C.c_int(0)
            rc = L.isi_conv_wgrad_deferred_f32(C.byref(s0), C.byref(s1) if s1 is not None else None, grad.data_ptr(),
                                               dw.data_ptr(), cin_true, None if tr else db.data_ptr(), ws.data_ptr(), nws,
                                               B, H, W, rows, k, k, layer.stride, layer.padding, WGRAD_FLAGS | pair_flags, _s(x),
                                               jobs, C.byref(n_jobs))
            _hip.check(rc, "isi_conv_wgrad_deferred_f32")
            defer.add(jobs, n_jobs.value, ws)
        else:
            rc = L.isi_conv_wgrad_torch_f32(C.byref(s0), C.byref(s1) if s1 is not None else None, grad.data_ptr(),
                                            dw.data_ptr(), cin_true, None if tr else db.data_ptr(), ws.data_ptr(), nws,
                                            B, H, W, rows, k, k, layer.stride, layer.padding, WGRAD_FLAGS | pair_flags, _s(x))
            _hip.check(rc, "isi_conv_wgrad_torch_f32")
        if tr:
            colsum(_rows2d(dy_nhwc), out=db)
        return dw, db
    packed = torch.empty(rows, Kpad, dtype=torch.float32, device=x.device)
    db = None if tr else torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = L.isi_conv_wgrad_f32(C.byref(s0), C.byref(s1) if s1 is not None else None, grad.data_ptr(),
                              packed.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), nws,
                              B, H, W, rows, k, k, layer.stride, layer.padding, WGRAD_FLAGS, _s(x))
    _hip.check(rc, "isi_conv_wgrad_f32")
    # [rows][kh][kw][cin_role] -> [rows, cin_role, kh, kw]  (= torch layout for both layer kinds)
    dw = layer.grouped(packed[:, :K].reshape(rows, k, k, cin_role)[..., :cin_true].permute(0, 3, 1, 2))
    if tr:
        db = colsum(_rows2d(dy_nhwc))
    return dw, db


class _DgradWeights:
    """Packed weights of the input-gradient convolutions, cached per weight version."""

    def __init__(self):
        self.cache: Dict[int, Tuple[tuple, torch.Tensor]] = {}

    def get(self, layer: _ConvParams) -> torch.Tensor:
        key = (_hip.version_of(layer.weight), layer.weight.data_ptr())
        hit = self.cache.get(id(layer))
        if hit is None or hit[0] != key:
            w = layer.dense_weight()
            if layer.transposed:
                # d/dx ConvT(k4,s2,p1) = Conv(k4,s2,p1) with weight [out=Cin_T, in=Cout_T]: same tensor
                packed = _ops.pack_conv_weight(w)
            elif layer.stride == 2:
                # d/dx Conv(k4,s2,p1) = ConvT(k4,s2,p1) with weight [in=Cout, out=Cin]: same tensor
                packed = _ops.pack_convT_weight(w)
            else:
                # d/dx Conv(k, s=1) = Conv(k, s=1, p=k-1-p) with the 180-degree rotated, transposed weight
                packed = _ops.pack_conv_dgrad_weight(w)
            hit = (key, packed)
            self.cache[id(layer)] = hit
        return hit[1]


BATCHED_PACK = os.environ.get("ISI_TRAIN_BATCHED_PACK", "1") != "0"


class PackGroup:
    """Every weight layout of one model's training step -- forward operands (packed + split-f16 pair copy), the operands of
    the input-gradient convolutions, both codebooks -- re-packed by ONE launch per step (isi_pack_multi) into persistent
    buffers, instead of one launch per layer and role when first asked for (63 launches of ~5 us per step of the default
    model in round 4).  `refresh()` runs at the head of every train-mode forward: when the parameters have moved since the
    last launch it re-packs everything and stamps the per-layer caches (`_ConvParams.packed`, `_DgradWeights`,
    `Quantize.packed`) so that their own version checks hit.  Grouped convolutions keep the per-layer path."""

    def __init__(self, model):
        self.model = model
        self.rows, self.stamps = [], []      # table rows; (setter of the per-layer cache) per row
        self.key = None
        self.table = None
        L = _hip.lib()
        dev = next(model.parameters()).device
        dw = model._dgrad_weights

        def buf(n):
            return torch.empty(n, dtype=torch.float32, device=dev)
        for layer in [m for m in model.modules() if isinstance(m, _ConvParams)]:
            if layer.groups != 1:
                continue
            w = layer.weight
            k = layer.kernel_size
            if layer.transposed:
                cin, cout = w.shape[0], w.shape[1]
                if (k, layer.stride, layer.padding) != (4, 2, 1):
                    continue
                n = L.isi_packed_convT_k4s2_weight_floats(cin, cout)
                small = n == 16 * cin * cout and cout <= 4 and cin in (32, 64)
                out = buf(2 * n)
                self._add(5 if small else 4, w, out, cin, cout, 4, 4, 0, self._fwd_setter(layer, out))
                # d/dx ConvT(k4,s2,p1) = Conv(k4,s2,p1) with weight [out = Cin_T, in = Cout_T]: the same tensor
                outd = buf(L.isi_packed_conv_weight_floats(cin, cout, 4, 4))
                self._add(0, w, outd, cin, cout, 4, 4, 0, self._dgrad_setter(dw, layer, outd))
            else:
                cout, cin = w.shape[0], w.shape[1]
                n = L.isi_packed_conv_weight_floats(cout, cin, k, k)
                out = buf(2 * n)
                self._add(1, w, out, cout, cin, k, k, 0, self._fwd_setter(layer, out))
                if layer.stride == 2 and k == 4 and layer.padding == 1:
                    # d/dx Conv(k4,s2,p1) = ConvT(k4,s2,p1) with weight [in = Cout, out = Cin]: the same tensor
                    nt = L.isi_packed_convT_k4s2_weight_floats(cout, cin)
                    small = nt == 16 * cout * cin and cin <= 4 and cout in (32, 64)
                    if small:
                        continue       # (the 2-channel first layer: its input gradient is never asked for; per-layer path if it is)
                    outd = buf(nt)
                    self._add(3, w, outd, cout, cin, 4, 4, 0, self._dgrad_setter(dw, layer, outd))
                elif layer.stride == 1:
                    outd = buf(L.isi_packed_conv_weight_floats(cin, cout, k, k))
                    self._add(2, w, outd, cout, cin, k, k, 0, self._dgrad_setter(dw, layer, outd))
        self.quantizers = []
        if not model.disable_quantization:
            for q in (model.quantize_t, model.quantize_b):
                D, K = q.embed.shape
                codes, e2 = torch.empty(K, D, dtype=torch.float32, device=dev), buf(K)
                self._add(6, q.embed, codes, D, K, 0, 0, e2.data_ptr(), self._codebook_setter(q, codes, e2))
                self.quantizers.append(q)
        self.table = torch.tensor(self.rows, dtype=torch.int64).to(dev)
        self.ptrs = tuple(r[1] for r in self.rows)

    def _add(self, kind, src, dst, d0, d1, kh, kw, aux, setter):
        self.rows.append([kind, src.data_ptr(), dst.data_ptr(), d0, d1, kh, kw, aux])
        self.stamps.append((src, setter))

    @staticmethod
    def _fwd_setter(layer, out):
        def f():
            layer._packed = out
            layer._packed_key = (_hip.version_of(layer.weight), layer.weight.data_ptr(), layer.weight.device)
        return f

    @staticmethod
    def _dgrad_setter(dw, layer, out):
        def f():
            dw.cache[id(layer)] = ((_hip.version_of(layer.weight), layer.weight.data_ptr()), out)
        return f

    @staticmethod
    def _codebook_setter(q, codes, e2):
        def f():
            q._packed = (codes, e2)
            q._packed_key = (_hip.version_of(q.embed), q.embed.data_ptr(), q.embed.device)
        return f

    def valid(self) -> bool:
        """The table holds raw pointers: a parameter that moved (`.to()`, a loaded checkpoint re-allocating) retires it."""
        return all(src.data_ptr() == p for (src, _), p in zip(self.stamps, self.ptrs))

    def refresh(self) -> None:
        key = tuple(_hip.version_of(src) for src, _ in self.stamps)
        if key == self.key and all(q._packed_key is not None for q in self.quantizers):
            return
        _hip.check(_hip.lib().isi_pack_multi(self.table.data_ptr(), len(self.rows), 16, _s(self.table)), "isi_pack_multi")
        for _, setter in self.stamps:
            setter()
        self.key = key


def refresh_packs(model) -> None:
    """Head of a train-mode forward: one launch re-packs whatever the optimizer (or the EMA update) has changed."""
    if not BATCHED_PACK:
        return
    grp = getattr(model, "_pack_group", None)
    if grp is None or not grp.valid():
        grp = model._pack_group = PackGroup(model)
    grp.refresh()


def conv_dgrad(dw: _DgradWeights, layer: _ConvParams, dy: torch.Tensor,
               residual: Optional[torch.Tensor] = None, gate=None) -> torch.Tensor:
    """Gradient w.r.t. the layer input ([B,Cin,H,W] view of channels-last storage);
    dy: [B,Cout,OH,OW] view (any strides).  `gate` (dense NHWC, the rectified activation the layer consumed): the
    backward of that ReLU -- zero where the activation is zero -- applied in the convolution's epilogue (after the
    residual is added) instead of a separate pass over the gradient."""
    packed = dw.get(layer)
    cin = layer.in_channels
    gflag = 0
    if isinstance(gate, PairOnly):      # a tape entry kept as pairs: its hi pieces are the mask
        gate, gflag = _gate_of(gate)
    if gate is not None and not gate.is_contiguous():
        raise ValueError("gate must be dense channels-last")
    if layer.transposed:
        return _ops.conv2d(dy, packed, None, cin, 4, 2, 1, relu=False, bf16x3=DGRAD_PRECISION, gate_nhwc=gate, extra_flags=gflag)
    if layer.stride == 2:
        return _ops.conv_transpose2d_k4s2(dy, packed, None, cin, relu=False, bf16x3=DGRAD_PRECISION, gate_nhwc=gate,
                                          extra_flags=gflag)
    k = layer.kernel_size
    return _ops.conv2d(dy, packed, None, cin, k, 1, k - 1 - layer.padding, relu=False, residual_bchw=residual,
                       bf16x3=DGRAD_PRECISION, gate_nhwc=gate, extra_flags=gflag)


def _g(entry):
    """Gate argument of conv_dgrad from a tape entry: pair-only entries travel as they are, fp32 views as dense NHWC."""
    return entry if isinstance(entry, PairOnly) else _nhwc(entry)


def _set_wb(grads, layer: _ConvParams, wb) -> None:
    grads.set(layer.weight, wb[0])
    grads.set(layer.bias, wb[1])


def _wgrad_into(grads, layer: _ConvParams, x, dy_nhwc, x2=None) -> None:
    """Weight and bias gradient of `layer`, written by the kernels straight into the flat gradient buffer."""
    _set_wb(grads, layer, conv_wgrad(layer, x, dy_nhwc, x2=x2, out=(grads.view(layer.weight), grads.view(layer.bias)),
                                     defer=grads.reductions))


class Tape:
    """Activations kept by the train-mode forward."""

    def __init__(self):
        self.t: Dict[str, torch.Tensor] = {}

    def __setitem__(self, k, v):
        self.t[k] = v

    def __getitem__(self, k):
        return self.t[k]


class Grads:
    """Parameter gradients in one flat buffer (views per parameter).  With data
    parallelism the buffer is cut into buckets of consecutive parameters; the backward
    fills it from the last layers to the first, and a bucket is all-reduced
    asynchronously (RCCL, its own stream) as soon as all of its gradients are written,
    overlapping the rest of the backward."""

    def __init__(self, model, n_buckets: int = 4):
        self.params = list(model.parameters())
        self.index = {id(p): i for i, p in enumerate(self.params)}
        sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.params[0].device)
        self.views: List[torch.Tensor] = []
        offsets = [0]
        for p, n in zip(self.params, sizes):
            self.views.append(self.flat[offsets[-1]:offsets[-1] + n].view_as(p))
            offsets.append(offsets[-1] + n)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # ISI_FORCE_COLLECTIVES=1 (tests / single-GPU measurements): issue the collectives in a 1-rank group too
        self.collectives = self.world > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())
        # buckets = contiguous parameter ranges of roughly equal size
        self.bucket_of = [min(n_buckets - 1, offsets[i] * n_buckets // max(1, offsets[-1])) for i in range(len(sizes))]
        self.bucket_span = [[None, None] for _ in range(n_buckets)]
        self.bucket_left = [0] * n_buckets
        for i, b in enumerate(self.bucket_of):
            span = self.bucket_span[b]
            span[0] = offsets[i] if span[0] is None else span[0]
            span[1] = offsets[i + 1]
            self.bucket_left[b] += 1
        self.handles = []
        self.n_collectives = 0      # bucket all-reduces issued (or, in a recorded step, planned) so far
        # split reductions of the weight-gradient GEMMs, launched together (None: each behind its GEMM)
        self.reductions = ReduceJobs() if (DEFER_REDUCTIONS and self.flat.is_cuda) else None

    def view(self, p: torch.nn.Parameter) -> torch.Tensor:
        """The parameter's slot of the flat buffer (kernels may write the gradient there themselves; `set` with
        that very tensor then only does the bucket accounting)."""
        return self.views[self.index[id(p)]]

    def set(self, p: torch.nn.Parameter, g: torch.Tensor) -> None:
        i = self.index[id(p)]
        if g is not self.views[i]:
            self.views[i].copy_(g.reshape(self.views[i].shape))
        b = self.bucket_of[i]
        self.bucket_left[b] -= 1
        if self.collectives and self.bucket_left[b] == 0:
            if self.reductions is not None:
                self.reductions.flush()       # the bucket's gradients must be complete before they travel
            a, e = self.bucket_span[b]
            bucket, handles = self.flat[a:e], self.handles
            self.n_collectives += 1

            def go():
                handles.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True))
            # a host call into RCCL: eagerly now, while the backward goes on; in a recorded step (utils/training/
            # graphed_step.py) the graph segment ends here and every replay makes the call between two segments
            host_boundary(go)

    def finish(self) -> List[torch.Tensor]:
        missing = [i for i, left in enumerate(self.bucket_left) if left > 0]
        assert not missing, "backward did not produce every parameter gradient"
        if self.reductions is not None:
            self.reductions.flush()
        if self.collectives:
            handles = self.handles

            def wait_all():
                for h in handles:
                    h.wait()
                handles.clear()
            host_boundary(wait_all)
            self.flat.div_(self.world)   # DDP semantics: average over ranks
        return self.views


# ------------------------------------------------------------------ forward (train mode)
# Round 5: the train-mode forward runs on the eval path's LDS-DMA kernels (conv_pair / convT_pair / resblock_pair,
# DESIGN.md section 4) wherever they apply: a producer writes its output in the split-f16 pair format -- which its
# consumer stages by plain LDS-DMA copies, no conversion per tap -- AND once more as fp32 (`twin`, the tape the backward's
# weight-gradient kernels and ReLU masks read); the fused residual block also hands out its hidden activation.  Layers
# outside those kernels' shapes (1x1 quantiser convolutions, the 2-channel last layer, grouped convolutions, product
# modes other than split-f16) keep the fp32-activation kernels.  ISI_TRAIN_PAIR_FORWARD=0: the round-4 path.
PAIR_FORWARD = os.environ.get("ISI_TRAIN_PAIR_FORWARD", "1") != "0"
_PAIR_FLAGS = 8 | 16      # ISI_CONV_F16X3 | ISI_CONV_W16


DROP_TWINS = os.environ.get("ISI_TRAIN_DROP_TWINS", "1") != "0"
# quantize_conv + codebook search of the training forward in ONE launch (the eval path's vq_conv1x1_nearest kernel, which
# also writes z for the backward and the EMA sums, and q in the pair format for the decoders); 0: a 1x1 convolution
# launch, the stand-alone search and a pair-encode pass (bit-identical results)
FUSED_QUANTIZER = os.environ.get("ISI_TRAIN_FUSED_QUANTIZER", "1") != "0"


class _Act:
    """An activation of the training forward: `f32` = [B,C,H,W] view of dense channels-last fp32 storage (the tape's
    usual form), `pair` = the same tensor in the pair format (same shape / strides) where a pair-route consumer follows.
    At least one of the two exists; `f32` is None where every consumer -- forward and backward -- reads pairs."""
    __slots__ = ("f32", "pair")

    def __init__(self, f32, pair=None):
        self.f32, self.pair = f32, pair

    @property
    def any(self):
        return self.f32 if self.f32 is not None else self.pair

    def want_pair(self):
        """The pair-format twin, encoded on the spot when the producer did not write one (quantiser outputs)."""
        if self.pair is None:
            self.pair = _ops.pair_encode(self.f32)
        return self.pair

    def tape(self):
        return self.f32 if self.f32 is not None else PairOnly(self.pair)


def _pair_mode() -> bool:
    return PAIR_FORWARD and FWD_PRECISION == 4


def _dense_cl(t: torch.Tensor) -> bool:
    return t.permute(0, 2, 3, 1).is_contiguous()


def _pairs_suffice(nxt, shape, c1: int = 0) -> bool:
    """Can the tensor of `shape` [B,C,H,W] that feeds `nxt` (a convolution layer, or a residual block) be kept as pairs
    ONLY?  Yes when (a) the layer's forward takes the LDS-DMA pair route, (b) its weight gradient runs the halo-staged
    kernel, which reads pair sources, and (c) the ReLU mask the tensor provides is applied by an implicit-GEMM input-gradient
    convolution (always the case for these layers).  `c1`: channels of a second, concatenated source."""
    if not (DROP_TWINS and _pair_mode() and WGRAD_FLAGS == 2 and DGRAD_PRECISION == 1):
        return False
    L = _hip.lib()
    B, Cc, H, W = shape
    if isinstance(nxt, RosinalityResBlock):
        c3, c1l = nxt.conv[1], nxt.conv[3]
        return bool(c3.groups == 1 and c1l.groups == 1 and _ops.resblock_fusable(Cc, c3.out_channels)
                    and L.isi_resblock_pair_route(B, H, W, Cc, c3.out_channels)
                    and L.isi_conv_wgrad_halo_route(c3.out_channels, Cc, 0, 3, 3, 1, 1, H, W))
    if nxt is None or nxt.transposed or nxt.groups != 1:
        return False
    k, st, pd = nxt.kernel_size, nxt.stride, nxt.padding
    OH, OW = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    return bool(L.isi_conv2d_pair_route(Cc, c1, nxt.out_channels, k, k)
                and L.isi_conv_wgrad_halo_route(nxt.out_channels, Cc, c1, k, k, st, pd, OH, OW))


def _conv_fwd(layer: _ConvParams, x: _Act, relu, x2: Optional[_Act] = None, out_nchw=False, keep_pair=True,
              need_f32=True) -> _Act:
    """One convolution / transposed convolution of the training forward.  `keep_pair` False: the output's consumers read
    fp32 (the 2-channel last layer, the 1x1 quantiser convolutions), so only fp32 is written.  `need_f32` False
    (`_pairs_suffice` for the consumer): no fp32 twin next to a pair-format output."""
    L = _hip.lib()
    k, cin, cout = layer.kernel_size, layer.in_channels, layer.out_channels
    xa = x.any
    first = (not layer.transposed and x2 is None and cin == 2 and (k, layer.stride, layer.padding) == (4, 2, 1)
             and cout in (32, 64) and x.f32 is not None and x.f32.is_contiguous())   # the NCHW spectrogram: conv_first_f32.hip
    pair_ok = (_pair_mode() and layer.groups == 1 and not out_nchw and (first or _dense_cl(xa))
               and (x2 is None or _dense_cl(x2.any)))
    if pair_ok and layer.transposed and x2 is None and (k, layer.stride, layer.padding) == (4, 2, 1) \
            and L.isi_conv_transpose2d_pair_route(cin, cout) and cin % 8 == 0:
        xp = x.want_pair()
        B, _, H, W = xp.shape
        s0 = _hip.src_nchw_view(xp)
        if not keep_pair:       # pair in, fp32 out: the plain entry point
            return _Act(_ops.conv_transpose2d_k4s2(xp, layer.packed(), layer.bias, cout, relu, bf16x3=4,
                                                   extra_flags=_ops.PAIR_IN0))
        if not need_f32:        # pair in, pair out
            return _Act(None, _ops.conv_transpose2d_k4s2(xp, layer.packed(), layer.bias, cout, relu, bf16x3=4,
                                                         extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT))
        out = torch.empty(B, 2 * H, 2 * W, cout, dtype=torch.float32, device=xp.device)
        twin = torch.empty_like(out)
        dst = _hip.dst_nchw_view(out.permute(0, 3, 1, 2))
        rc = L.isi_conv_transpose2d_k4s2_twin_f32(C.byref(s0), layer.packed().data_ptr(), layer.bias.data_ptr(), C.byref(dst),
                                                  twin.data_ptr(), B, H, W, cout,
                                                  int(relu) | _PAIR_FLAGS | _ops.PAIR_IN0 | _ops.PAIR_OUT, _s(xp))
        _hip.check(rc, "isi_conv_transpose2d_k4s2_twin_f32")
        return _Act(twin.permute(0, 3, 1, 2), out.permute(0, 3, 1, 2))
    c0 = xa.shape[1]
    c1 = x2.any.shape[1] if x2 is not None else 0
    if pair_ok and not layer.transposed and keep_pair and (first or L.isi_conv2d_pair_route(c0, c1, cout, k, k)):
        src = x.f32 if first else x.want_pair()
        src2 = x2.want_pair() if x2 is not None else None
        flags = _ops.PAIR_OUT | (0 if first else _ops.PAIR_IN0) | (_ops.PAIR_IN1 if src2 is not None else 0)
        if not need_f32:
            return _Act(None, _ops.conv2d(src, layer.packed(), layer.bias, cout, k, layer.stride, layer.padding, relu,
                                          x2_bchw=src2, bf16x3=4, extra_flags=flags))
        B, _, H, W = src.shape
        OH = (H + 2 * layer.padding - k) // layer.stride + 1
        OW = (W + 2 * layer.padding - k) // layer.stride + 1
        out = torch.empty(B, OH, OW, cout, dtype=torch.float32, device=src.device)
        twin = torch.empty_like(out)
        s0 = _hip.src_nchw_view(src)
        s1 = _hip.src_nchw_view(src2) if src2 is not None else None
        dst = _hip.dst_nchw_view(out.permute(0, 3, 1, 2))
        rc = L.isi_conv2d_twin_f32(C.byref(s0), C.byref(s1) if s1 is not None else None, layer.packed().data_ptr(),
                                   layer.bias.data_ptr(), C.byref(dst), twin.data_ptr(), B, H, W, cout, k, k,
                                   layer.stride, layer.padding, int(relu) | _PAIR_FLAGS | flags, _s(src))
        _hip.check(rc, "isi_conv2d_twin_f32")
        return _Act(twin.permute(0, 3, 1, 2), out.permute(0, 3, 1, 2))
    if pair_ok and not layer.transposed and not keep_pair and L.isi_conv2d_pair_route(c0, c1, cout, k, k):
        # pair sources, fp32 output only
        extra = _ops.PAIR_IN0 | (_ops.PAIR_IN1 if x2 is not None else 0)
        return _Act(_ops.conv2d(x.want_pair(), layer.packed(), layer.bias, cout, k, layer.stride, layer.padding, relu,
                                x2_bchw=x2.want_pair() if x2 is not None else None, bf16x3=4, extra_flags=extra))
    if x.f32 is None or (x2 is not None and x2.f32 is None):
        raise RuntimeError("training forward: a tensor was kept as pairs only, but its consumer reads fp32")
    return _Act(layer.run(x.f32, relu=relu, x2=x2.f32 if x2 is not None else None, out_nchw=out_nchw, bf16x3=FWD_PRECISION))


def _res_block_fwd(blk: RosinalityResBlock, x: _Act, tape: "Tape", key: str, need_f32: bool = True) -> _Act:
    """relu(r + conv1x1(relu(conv3x3(r)))) on a rectified r; tape[key.h] = the hidden activation, tape[key.y] = the output."""
    c3, c1 = blk.conv[1], blk.conv[3]
    L = _hip.lib()
    B, Cc, H, W = x.any.shape
    R = c3.out_channels
    if (_pair_mode() and c3.groups == 1 and c1.groups == 1 and _dense_cl(x.any) and _ops.resblock_fusable(Cc, R)
            and L.isi_resblock_pair_route(B, H, W, Cc, R)):
        xp = x.want_pair().permute(0, 2, 3, 1)
        out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=xp.device)
        twin = torch.empty_like(out) if need_f32 else None
        hid = torch.empty(B, H, W, R, dtype=torch.float32, device=xp.device)
        rc = L.isi_resblock_tape_f32(xp.data_ptr(), c3.packed().data_ptr(), c3.bias.data_ptr(), c1.packed().data_ptr(),
                                     c1.bias.data_ptr(), out.data_ptr(), twin.data_ptr() if twin is not None else None,
                                     hid.data_ptr(), B, H, W, Cc, R,
                                     1 | _PAIR_FLAGS | _ops.PAIR_IN0 | _ops.PAIR_OUT, _s(xp))
        _hip.check(rc, "isi_resblock_tape_f32")
        y = _Act(twin.permute(0, 3, 1, 2) if twin is not None else None, out.permute(0, 3, 1, 2))
        tape[f"{key}.h"], tape[f"{key}.y"] = hid.permute(0, 3, 1, 2), y.tape()
        return y
    if x.f32 is None:
        raise RuntimeError("training forward: a tensor was kept as pairs only, but its consumer reads fp32")
    h = c3.run(x.f32, relu=True, bf16x3=FWD_PRECISION)
    yv = c1.run(h, relu=True, residual=x.f32, bf16x3=FWD_PRECISION)
    tape[f"{key}.h"], tape[f"{key}.y"] = h, yv
    return _Act(yv)


def _as_act(x) -> _Act:
    return x if isinstance(x, _Act) else _Act(x)


def _res_stack_fwd(blocks, idxs, x: _Act, tape: "Tape", tag: str) -> _Act:
    """The residual stack; a block's output is kept as pairs only where the NEXT block reads nothing else (the stack's
    own output always keeps its fp32 form: other modules and the stack's top-level ReLU mask read it)."""
    for j, i in enumerate(idxs):
        nxt = blocks[idxs[j + 1]] if j + 1 < len(idxs) else None
        need = not (nxt is not None and _pairs_suffice(nxt, x.any.shape))
        x = _res_block_fwd(blocks[i], x, tape, f"{tag}.res{j}", need_f32=need)
    return x


def encoder_forward(m: RosinalityEncoder, x, tape: Tape, tag: str) -> _Act:
    x = _as_act(x)
    tape[f"{tag}.in"] = x.tape()
    seq = [m.blocks[i] for i in m._down] + [m.blocks[m._conv3]]
    for j, layer in enumerate(seq):
        last = j == len(seq) - 1
        nxt = (m.blocks[m._res[0]] if m._res else None) if last else seq[j + 1]
        B, _, H, W = x.any.shape
        k, st, pd = layer.kernel_size, layer.stride, layer.padding
        oshape = (B, layer.out_channels, (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1)
        x = _conv_fwd(layer, x, True, need_f32=not _pairs_suffice(nxt, oshape))
        tape[f"{tag}.c3" if last else f"{tag}.down{j}"] = x.tape()
    return _res_stack_fwd(m.blocks, m._res, x, tape, tag)


def decoder_forward(m: RosinalityDecoder, x, x2, tape: Tape, tag: str, out_nchw_last: bool, last_pair: bool = False) -> _Act:
    """`last_pair`: the decoder's output is read by a pair-route consumer (dec_t feeds nothing of the kind: its consumer
    is the 1x1 quantiser convolution)."""
    x = _as_act(x)
    x2 = _as_act(x2) if x2 is not None else None
    tape[f"{tag}.in"], tape[f"{tag}.in2"] = x.tape(), (x2.tape() if x2 is not None else None)
    c3 = m.blocks[0]
    B, _, H, W = x.any.shape
    nxt = m.blocks[m._res[0]] if m._res else None
    x = _conv_fwd(c3, x, True, x2=x2, need_f32=not _pairs_suffice(nxt, (B, c3.out_channels, H, W)))
    tape[f"{tag}.c3"] = x.tape()
    x = _res_stack_fwd(m.blocks, m._res, x, tape, tag)
    for j, i in enumerate(m._up):
        last = j == len(m._up) - 1
        nxt = m.blocks[m._up[j + 1]] if not last else None
        # the few-channel last layer (Cout <= 4) and whatever follows the decoder read fp32
        keep = last_pair if last else nxt.out_channels > 4
        x = _conv_fwd(m.blocks[i], x, not last, out_nchw=(last and out_nchw_last), keep_pair=keep)
        tape[f"{tag}.up{j}"] = x.tape()
    return x


def _res_stack_backward(blocks, idxs, tape, tag, d_y, x_in_key, dw, grads: Grads, gated: bool = False):
    """d_y: dense NHWC gradient w.r.t. the (rectified) stack output -- `gated`: already zeroed where that output is
    zero.  Returns the NHWC gradient w.r.t. the stack input (the rectified conv3 output), gated by that input: every
    ReLU backward below the top one is applied in the epilogue of the input-gradient convolution that produces the
    gradient (conv_dgrad's `gate`), not as a pass of its own."""
    for j in reversed(range(len(idxs))):
        blk = blocks[idxs[j]]
        y, h = tape[f"{tag}.res{j}.y"], tape[f"{tag}.res{j}.h"]
        r = tape[f"{tag}.res{j - 1}.y"] if j > 0 else tape[x_in_key]
        g = d_y if gated else relu_bwd_(d_y, _nhwc(y))                 # through relu(r + conv1(h)) (a stack's output keeps fp32)
        c1, c3 = blk.conv[3], blk.conv[1]
        _wgrad_into(grads, c1, h, g)
        dh = _nhwc(conv_dgrad(dw, c1, _as_bchw(g), gate=_g(h)))        # through relu(conv3(r))
        _wgrad_into(grads, c3, r, dh)
        # + skip connection, then through the ReLU that produced r
        d_y = _nhwc(conv_dgrad(dw, c3, _as_bchw(dh), residual=_as_bchw(g), gate=_g(r)))
        gated = True
    return d_y if gated else relu_bwd_(d_y, _nhwc(tape[x_in_key]))


def encoder_backward(m: RosinalityEncoder, tape: Tape, tag: str, d_out, dw, grads: Grads, need_input_grad: bool,
                     gated: bool = False):
    """d_out: dense NHWC gradient w.r.t. the encoder output (`gated`: already zeroed where that rectified output is zero)."""
    g = _res_stack_backward(m.blocks, m._res, tape, tag, d_out, f"{tag}.c3", dw, grads, gated=gated)   # gated by the conv3 output
    c3 = m.blocks[m._conv3]
    prev = tape[f"{tag}.down{len(m._down) - 1}"]
    _wgrad_into(grads, c3, prev, g)
    g = _nhwc(conv_dgrad(dw, c3, _as_bchw(g), gate=_g(prev)))
    for j in reversed(range(len(m._down))):
        layer = m.blocks[m._down[j]]
        prev = tape[f"{tag}.down{j - 1}"] if j > 0 else tape[f"{tag}.in"]
        _wgrad_into(grads, layer, prev, g)
        if j > 0:
            g = _nhwc(conv_dgrad(dw, layer, _as_bchw(g), gate=_g(prev)))
        elif need_input_grad:
            g = _nhwc(conv_dgrad(dw, layer, _as_bchw(g)))              # the encoder input is not a ReLU output
    return g if need_input_grad else None


def decoder_backward(m: RosinalityDecoder, tape: Tape, tag: str, d_out_bchw, dw, grads: Grads):
    """d_out_bchw: gradient w.r.t. the decoder output as a [B,C,H,W] view (any layout).
    Returns the dense NHWC gradient w.r.t. cat(in, in2)."""
    d_view = d_out_bchw
    for j in reversed(range(len(m._up))):
        layer = m.blocks[m._up[j]]
        # below the last layer: our own input-gradient output, gated by this layer's output.  A transposed layer reads its
        # output gradient as a strided SOURCE (weight gradient with the roles swapped, input gradient = a convolution of
        # it): a channel slice needs no dense copy
        g = _rows(d_view) if layer.transposed else _nhwc(d_view)
        prev = tape[f"{tag}.up{j - 1}"] if j > 0 else tape[f"{tag}.res{len(m._res) - 1}.y" if m._res else f"{tag}.c3"]
        _wgrad_into(grads, layer, prev, g)
        d_view = conv_dgrad(dw, layer, _as_bchw(g), gate=_g(prev))
    g = _res_stack_backward(m.blocks, m._res, tape, tag, _nhwc(d_view), f"{tag}.c3", dw, grads, gated=True)
    c3 = m.blocks[0]
    _wgrad_into(grads, c3, tape[f"{tag}.in"], g, x2=tape[f"{tag}.in2"])
    return _nhwc(conv_dgrad(dw, c3, _as_bchw(g)))


# ------------------------------------------------------------------ quantiser (train mode)
def _ema_collectives() -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


def exchange_ema_statistics(counts: torch.Tensor, embed_sum: torch.Tensor):
    """Data-parallel exchange of the EMA statistics of one quantiser: the per-code usage counts `onehot.sum(0)` [K]
    and the per-code vector sums `flatten^T @ onehot` [D, K] (bottleneck.py:80-84) of every rank's batch shard are
    summed in ONE all-reduce message ([K] + [D*K] floats), so that N ranks x B/N samples update the codebook exactly
    like one process with B samples (the reference lets DDP broadcast rank 0's buffers instead).
    Identity when not distributed.  Host logic only (any device: RCCL on the GPU, gloo in the CPU tests).
    Blocking form (tests, stand-alone callers); the training step uses `PendingEma` below."""
    if not _ema_collectives():
        return counts, embed_sum
    K = counts.numel()
    packed = torch.cat([counts.reshape(-1).float(), embed_sum.reshape(-1).float()])
    dist.all_reduce(packed)
    return packed[:K], packed[K:].reshape(embed_sum.shape)


class PendingEma:
    """EMA codebook updates whose batch statistics are still travelling (round 5).

    The updated codebook is first needed by the NEXT search of its quantiser, so nothing in the rest of the forward has
    to wait for the statistics' all-reduce: `submit` launches it asynchronously (RCCL's own stream; one message of
    [K] + [D K] floats per quantiser) and the forward goes on -- decoder, bottom quantiser, ... -- ; `flush`, called once
    at the end of the train-mode forward, waits for the handles and only then launches the `isi_vq_ema_update_f32` kernels
    that write the codebooks.  The blocking form stalled the stream against RCCL's latency twice per step on every rank.
    Both host calls sit behind `host_boundary`, so a recorded step replays them between graph segments."""

    def __init__(self):
        self.items = []        # (quantiser, packed statistics [K + D K], D, K)
        self.handles = []

    def submit(self, q, counts_f: torch.Tensor, embed_sum: torch.Tensor) -> None:
        D, K = embed_sum.shape
        packed = torch.cat([counts_f.reshape(-1), embed_sum.reshape(-1)])
        self.items.append((q, packed, D, K))
        if _ema_collectives():
            handles = self.handles

            def go():
                handles.append(dist.all_reduce(packed, async_op=True))
            host_boundary(go)

    def flush(self) -> None:
        if not self.items:
            return
        if _ema_collectives():
            handles = self.handles

            def wait_all():
                for h in handles:
                    h.wait()
                handles.clear()
            host_boundary(wait_all)
        for q, packed, D, K in self.items:
            self._update(q, packed, D, K)
        self.items = []

    @staticmethod
    def _update(q, packed: torch.Tensor, D: int, K: int) -> None:
        """bottleneck.py:80-92 on the [D,K] buffers from the (summed) statistics `packed` = counts [K] | embed_sum [D K]."""
        _hip.check(_hip.lib().isi_vq_ema_update_f32(q.embed.data_ptr(), q.cluster_size.data_ptr(), q.embed_avg.data_ptr(),
                                                    packed.data_ptr(), packed.data_ptr() + 4 * K, D, K, q.decay, q.eps,
                                                    _s(packed)), "isi_vq_ema_update_f32")
        # the buffers were written through raw pointers (no torch version bump): invalidate caches
        q._packed_key = None
        q._ema_steps = getattr(q, "_ema_steps", 0) + 1


def fused_quantizer_applies(q, a: "_Act", a2: Optional["_Act"] = None) -> bool:
    """Can quantize_conv on cat(a, a2) and the search of `q` run as the one fused launch?  (pair-format sources of the
    kernel's shapes, no index corruption: the corrupted codes are drawn on the host between search and statistics)"""
    if not (FUSED_QUANTIZER and _pair_mode()) or q.corruption_weights is not None or q.dim != 64:
        return False
    if a.pair is None or (a2 is not None and a2.pair is None) or not _dense_cl(a.pair) or (a2 is not None and not _dense_cl(a2.pair)):
        return False
    c1 = a2.pair.shape[1] if a2 is not None else 0
    return bool(_hip.lib().isi_vq_conv1x1_fusable(a.pair.shape[1], c1, q.dim, q.n_embed))


def quantize_conv_train(qconv, q, a: "_Act", a2: Optional["_Act"], pending: Optional[PendingEma] = None):
    """quantize_conv_* (vqvae.py:260,272) + quantize_train in one launch: returns (z [B,H,W,D], q as an _Act with its
    pair twin, diff, idx, perplexity); z and every output are bit-identical to the two-launch path."""
    L = _hip.lib()
    D, K = q.dim, q.n_embed
    codes, e2 = q.packed()
    B, C0, H, W = a.pair.shape
    C1 = a2.pair.shape[1] if a2 is not None else 0
    dev = a.pair.device

    def src(t):      # [B,C,H,W] view of dense channels-last storage
        return _hip.isi_src(t.data_ptr(), t.shape[1], t.stride(0), 1, t.stride(2), t.stride(3))
    s0, s1 = src(a.pair), (src(a2.pair) if a2 is not None else None)
    N = B * H * W
    z = torch.empty(B, H, W, D, dtype=torch.float32, device=dev)
    q_st, q_pair = torch.empty_like(z), torch.empty_like(z)
    idx = torch.empty(B, H, W, dtype=torch.int64, device=dev)
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    n_part = L.isi_vq_num_partials(N)
    part = torch.empty(n_part, dtype=torch.float32, device=dev)
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    ws = torch.empty(max(4, L.isi_vq_conv1x1_workspace_floats(C0, C1, D)), dtype=torch.float32, device=dev)
    packed = qconv.packed()
    _hip.check(L.isi_vq_conv1x1_nearest_tape_f32(C.byref(s0), C.byref(s1) if s1 is not None else None,
                                                 packed.data_ptr() + 2 * packed.numel(), qconv.bias.data_ptr(),
                                                 codes.data_ptr(), e2.data_ptr(), idx.data_ptr(), q_st.data_ptr(),
                                                 q_pair.data_ptr(), z.data_ptr(), counts.data_ptr(), part.data_ptr(),
                                                 ws.data_ptr(), B, H, W, D, K, _s(z)),
               "isi_vq_conv1x1_nearest_tape_f32")
    _hip.check(L.isi_vq_finalize_f32(part.data_ptr(), n_part, counts.data_ptr(), K, N, D, out2.data_ptr(), _s(z)),
               "isi_vq_finalize_f32")
    _ema_statistics(q, z, idx, counts, pending)
    return z, _Act(q_st.permute(0, 3, 1, 2), q_pair.permute(0, 3, 1, 2)), out2[0], idx, out2[1]


def _ema_statistics(q, z_nhwc, idx, counts, pending: Optional[PendingEma]) -> None:
    """Cluster sizes and per-code sums of z (bottleneck.py:75-83) handed to the EMA update."""
    L = _hip.lib()
    D, K = q.dim, q.n_embed
    N = z_nhwc.numel() // D
    embed_sum = torch.empty(D, K, dtype=torch.float32, device=z_nhwc.device)
    nws = L.isi_vq_embed_sum_workspace_floats(D, K, N)
    ws = torch.empty(nws, dtype=torch.float32, device=z_nhwc.device)
    _hip.check(L.isi_vq_embed_sum_f32(z_nhwc.data_ptr(), idx.data_ptr(), embed_sum.data_ptr(), ws.data_ptr(), nws,
                                      N, D, K, _s(z_nhwc)), "isi_vq_embed_sum_f32")
    own = pending is None
    if own:
        pending = PendingEma()
    pending.submit(q, counts.float(), embed_sum)
    if own:
        pending.flush()


def quantize_train(q, z_nhwc: torch.Tensor, pending: Optional[PendingEma] = None):
    """Eval-identical search with the CURRENT codebook, optional index corruption (bottleneck.py:63-73),
    then the EMA update of the buffers (bottleneck.py:75-92) from statistics all-reduced over the
    data-parallel ranks.  `pending`: the update is handed to it (asynchronous exchange, codebook written at its
    `flush()`); None: exchanged and written before this returns."""
    codes, e2 = q.packed()
    L = _hip.lib()
    D, K = q.dim, q.n_embed
    N = z_nhwc.numel() // D
    idx = torch.empty(z_nhwc.shape[:-1], dtype=torch.int64, device=z_nhwc.device)
    q_st = torch.empty_like(z_nhwc)
    counts = torch.zeros(K, dtype=torch.int32, device=z_nhwc.device)
    n_part = L.isi_vq_num_partials(N)
    part = torch.empty(n_part, dtype=torch.float32, device=z_nhwc.device)
    out2 = torch.empty(2, dtype=torch.float32, device=z_nhwc.device)
    # candidates on the f16 pipe + fp32 decision (the eval path's search: the fp32 formula's indices at a third of the
    # time).  Code vectors beyond the f16 operand range -- the EMA update divides an unused code by a vanishing cluster
    # size: 4e5 after the FIRST step of a random model -- are handled exactly by the kernel (vq_decide_f32).
    flags = 8 if (FWD_PRECISION in (3, 4) and D == 64) else 0
    _hip.check(L.isi_vq_nearest_flags_f32(z_nhwc.data_ptr(), codes.data_ptr(), e2.data_ptr(), idx.data_ptr(),
                                          q_st.data_ptr(), counts.data_ptr(), part.data_ptr(), N, D, K, flags,
                                          _s(z_nhwc)),
               "isi_vq_nearest_flags_f32")
    _hip.check(L.isi_vq_finalize_f32(part.data_ptr(), n_part, counts.data_ptr(), K, N, D, out2.data_ptr(),
                                     _s(z_nhwc)), "isi_vq_finalize_f32")
    diff, perplexity = out2[0], out2[1]
    if q.corruption_weights is not None:
        if z_nhwc.is_cuda and torch.cuda.is_current_stream_capturing():
            raise NotImplementedError("index corruption draws on the host (torch.multinomial, like the reference): such a "
                                      "step cannot be recorded into a HIP graph")
        # offsets in {-1, 0, +1} drawn exactly like the reference: torch.multinomial on the CPU default
        # generator (same seed -> same offsets), moved to the device and added modulo K; every quantity
        # downstream (codes, diff, usage statistics, EMA sums) is then taken from the corrupted indices
        offsets = torch.multinomial(torch.Tensor(q.corruption_weights), idx.numel(), replacement=True) - 1
        idx = (idx + offsets.reshape(idx.shape).to(idx.device)) % K
        qv = _ops.embed_code(idx, codes)
        delta = qv - z_nhwc
        q_st = z_nhwc + delta
        diff = delta.pow(2).mean()
        counts = torch.bincount(idx.reshape(-1), minlength=K).to(torch.int32)
        p = counts.float() / N
        perplexity = torch.exp(-torch.sum(p * torch.log(p.clamp(min=1e-7))))
    _ema_statistics(q, z_nhwc, idx, counts, pending)
    return q_st, diff, idx, perplexity


class QuantizeTrainFunction(torch.autograd.Function):
    """Stand-alone train-mode QuantizedBottleneck.forward (bottleneck.py:53-101): straight-through
    gradient for the quantised output plus the commitment gradient 2 (z - q) / numel through `diff`."""

    @staticmethod
    def forward(ctx, q, z):
        z = z.contiguous()
        q_st, diff, idx, perplexity = quantize_train(q, z)
        ctx.save_for_backward(z, q_st)
        ctx.mark_non_differentiable(idx, perplexity)
        return q_st, diff, idx, perplexity

    @staticmethod
    def backward(ctx, dq, ddiff, _di, _dp):
        z, q_st = ctx.saved_tensors
        dq = torch.zeros_like(z) if dq is None else dq.contiguous()
        g = torch.zeros((), device=z.device) if ddiff is None else ddiff.reshape(())
        return None, vq_backward(dq, z, q_st, g.contiguous())


@torch.no_grad()
def encode_train(model, x: torch.Tensor):
    """VQVAE.encode under model.train() (vqvae.py:251-278 with the quantisers in train mode): same layer sequence as
    the train-mode forward up to the bottom quantiser, EMA buffers updated in-forward; detached outputs."""
    tape = Tape()
    x = x.contiguous()
    if hasattr(model, "_dgrad_weights"):
        refresh_packs(model)
    enc_b = encoder_forward(model.enc_b, x, tape, "enc_b")
    enc_t = encoder_forward(model.enc_t, enc_b, tape, "enc_t").f32
    enc_b = enc_b.f32
    z_t = _nhwc(model.quantize_conv_t.run(enc_t, relu=False, bf16x3=FWD_PRECISION))
    pending = PendingEma()
    q_t, diff_t, id_t, perp_t = quantize_train(model.quantize_t, z_t, pending)
    dec_t = decoder_forward(model.dec_t, _as_bchw(q_t), None, tape, "dec_t", out_nchw_last=False).f32
    if dec_t.shape[-1] != enc_b.shape[-1]:
        if not model.adapt_quantized_durations:
            raise RuntimeError("Sizes of tensors must match except in dimension 1")
        w = min(dec_t.shape[-1], enc_b.shape[-1])            # vqvae.py:266-269
        dec_t, enc_b = dec_t[..., :w], enc_b[..., :w]
    z_b = _nhwc(model.quantize_conv_b.run(dec_t, relu=False, x2=enc_b, bf16x3=FWD_PRECISION))
    q_b, diff_b, id_b, perp_b = quantize_train(model.quantize_b, z_b, pending)
    pending.flush()
    return _as_bchw(q_t), _as_bchw(q_b), (diff_t + diff_b).reshape(1), id_t, id_b, perp_t, perp_b


def vq_backward(dq_nhwc, z_nhwc, q_st_nhwc, g_diff):
    """dq_nhwc: dense, or a `_rows` tensor (uniformly strided pixels: a channel slice of a wider gradient, read in place)."""
    dz = torch.empty_like(z_nhwc)
    if dq_nhwc.is_contiguous():
        _hip.check(_hip.lib().isi_vq_bwd_f32(dz.data_ptr(), dq_nhwc.data_ptr(), z_nhwc.data_ptr(),
                                             q_st_nhwc.data_ptr(), g_diff.data_ptr(), z_nhwc.numel(), _s(z_nhwc)),
                   "isi_vq_bwd_f32")
        return dz
    D = dq_nhwc.shape[-1]
    _hip.check(_hip.lib().isi_vq_bwd_rows_f32(dz.data_ptr(), dq_nhwc.data_ptr(), dq_nhwc.stride(-2), z_nhwc.data_ptr(),
                                              q_st_nhwc.data_ptr(), g_diff.data_ptr(), z_nhwc.numel() // D, D, _s(z_nhwc)),
               "isi_vq_bwd_rows_f32")
    return dz


# ------------------------------------------------------------------ whole model
class VQVAETrainFunction(torch.autograd.Function):
    """forward(x, *params) -> (dec, diff); backward -> parameter gradients."""

    @staticmethod
    def forward(ctx, model, x, *params):
        tape = Tape()
        D = model.embed_dim
        x = x.contiguous()
        refresh_packs(model)
        enc_b_act = encoder_forward(model.enc_b, x, tape, "enc_b")
        enc_t_act = encoder_forward(model.enc_t, enc_b_act, tape, "enc_t")
        enc_t, enc_b = enc_t_act.f32, enc_b_act.f32
        unq = model.disable_quantization      # UnquantizedBottleneck (bottleneck.py:107-119): identity, diff 0

        def _identity(z):
            dev = z.device
            return (z, torch.zeros((), device=dev), torch.zeros(0, dtype=torch.int64, device=dev),
                    torch.full((), float("inf"), device=dev))
        pending = PendingEma()
        # quantize_conv + search: one launch where the fused kernel applies (q then arrives with its pair twin)
        fuse_t = not unq and fused_quantizer_applies(model.quantize_t, enc_t_act)
        if fuse_t:
            z_t, q_t_act, diff_t, id_t, perp_t = quantize_conv_train(model.quantize_conv_t, model.quantize_t, enc_t_act, None, pending)
            q_t = _nhwc(q_t_act.f32)
        else:
            z_t = _nhwc(model.quantize_conv_t.run(enc_t, relu=False, bf16x3=FWD_PRECISION))
            q_t, diff_t, id_t, perp_t = _identity(z_t) if unq else quantize_train(model.quantize_t, z_t, pending)
            q_t_act = _Act(_as_bchw(q_t))
        tape["z_t"], tape["q_t"] = z_t, q_t
        fuse_b = (not unq and FUSED_QUANTIZER and _pair_mode() and model.quantize_b.corruption_weights is None
                  and enc_b_act.pair is not None)
        dec_t_act = decoder_forward(model.dec_t, q_t_act, None, tape, "dec_t", out_nchw_last=False, last_pair=fuse_b)
        dec_t = dec_t_act.f32
        if dec_t.shape[-1] != enc_b.shape[-1]:
            raise NotImplementedError("training needs input sizes divisible by the total down-sampling factor")
        if fuse_b and fused_quantizer_applies(model.quantize_b, dec_t_act, enc_b_act):
            z_b, q_b_act, diff_b, id_b, perp_b = quantize_conv_train(model.quantize_conv_b, model.quantize_b, dec_t_act,
                                                                     enc_b_act, pending)
            q_b = _nhwc(q_b_act.f32)
        else:
            z_b = _nhwc(model.quantize_conv_b.run(dec_t, relu=False, x2=enc_b, bf16x3=FWD_PRECISION))
            q_b, diff_b, id_b, perp_b = _identity(z_b) if unq else quantize_train(model.quantize_b, z_b, pending)
            q_b_act = _Act(_as_bchw(q_b))
        tape["z_b"], tape["q_b"], tape["dec_t"], tape["enc_b"], tape["enc_t"] = z_b, q_b, dec_t, enc_b, enc_t
        up = q_t_act
        n_up = len(model.upsample_top_to_bottom)
        for j, layer in enumerate(model.upsample_top_to_bottom):
            tape[f"up.in{j}"] = up.tape()
            # (intermediate outputs feed another transposed layer -- fp32 for its weight gradient --; the last one feeds the
            # bottom decoder's two-source 3x3, which reads pairs throughout)
            B_, _, H_, W_ = up.any.shape
            oshape = (B_, layer.out_channels, 2 * H_, 2 * W_)
            need = not (j == n_up - 1 and _pairs_suffice(model.dec.blocks[0], oshape, c1=model.embed_dim))
            up = _conv_fwd(layer, up, False, need_f32=need)
        dec = decoder_forward(model.dec, up, q_b_act, tape, "dec", out_nchw_last=True).f32
        diff = (diff_t + diff_b).reshape(1)
        pending.flush()       # the statistics' all-reduces have had the rest of the forward to arrive; codebooks written here
        ctx.model, ctx.tape = model, tape
        ctx.mark_non_differentiable(id_t, id_b, perp_t, perp_b)
        return dec, diff, perp_t, perp_b, id_t, id_b

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_dec, g_diff, *_unused):
        model, tape = ctx.model, ctx.tape
        dev = tape["z_t"].device
        if g_dec is None:
            g_dec = torch.zeros_like(tape[f"dec.up{len(model.dec._up) - 1}"])
        if g_diff is None:
            g_diff = torch.zeros(1, device=dev)
        g_diff = g_diff.reshape(1).contiguous().float()
        dw = model._dgrad_weights
        grads = Grads(model)
        D = model.embed_dim
        # decoder (bottom): d cat(up, q_b)
        d_cat = decoder_backward(model.dec, tape, "dec", g_dec, dw, grads)
        d_up = _as_bchw(d_cat)[:, :D]
        d_qb = _rows(_as_bchw(d_cat)[:, D:])       # (a channel slice, read in place by the quantiser backward)
        # upsample_top_to_bottom: plain transposed convs
        d_view = d_up
        for j in reversed(range(len(model.upsample_top_to_bottom))):
            layer = model.upsample_top_to_bottom[j]
            g = _rows(d_view) if layer.transposed else _nhwc(d_view)
            _wgrad_into(grads, layer, tape[f"up.in{j}"], g)
            d_view = conv_dgrad(dw, layer, _as_bchw(g))
        d_qt = _nhwc(d_view)          # (a fresh tensor: the input-gradient convolution's own output, or a dense copy of a slice)
        if len(model.upsample_top_to_bottom) == 0:
            d_qt = d_qt.clone()
        # bottom quantiser and its 1x1 conv on cat(dec_t, enc_b)
        unq = model.disable_quantization
        d_zb = d_qb.contiguous() if unq else vq_backward(d_qb, tape["z_b"], tape["q_b"], g_diff)
        qcb = model.quantize_conv_b
        _wgrad_into(grads, qcb, tape["dec_t"], d_zb, x2=tape["enc_b"])
        d_cat2 = conv_dgrad(dw, qcb, _as_bchw(d_zb))
        Cd = tape["dec_t"].shape[1]
        # two channel slices of one tensor: d_dect is read in place by dec_t's last transposed convolution, the enc_b half
        # waits (in place) for enc_t's contribution
        d_dect, d_encb_part = d_cat2[:, :Cd], _rows(d_cat2[:, Cd:])
        # dec_t
        d_qt2 = decoder_backward(model.dec_t, tape, "dec_t", d_dect, dw, grads)
        axpy_(d_qt, d_qt2)
        # top quantiser and its 1x1 conv
        d_zt = d_qt if unq else vq_backward(d_qt, tape["z_t"], tape["q_t"], g_diff)
        qct = model.quantize_conv_t
        _wgrad_into(grads, qct, tape["enc_t"], d_zt)
        # the ReLU masks of the two encoders' outputs ride in the producers of their gradients: the 1x1 input-gradient
        # convolution's gated epilogue (enc_t) and the sum of enc_b's two contributions (one pass instead of a slice copy,
        # an axpy and a mask pass over 134 MB each)
        d_enct = _nhwc(conv_dgrad(dw, qct, _as_bchw(d_zt), gate=_nhwc(tape["enc_t"])))
        d_encb2 = encoder_backward(model.enc_t, tape, "enc_t", d_enct, dw, grads, need_input_grad=True, gated=True)
        d_encb = add_gate_rows(d_encb_part, d_encb2, _nhwc(tape["enc_b"]))
        encoder_backward(model.enc_b, tape, "enc_b", d_encb, dw, grads, need_input_grad=False, gated=True)
        views = grads.finish()
        ctx.tape = None
        return (None, None) + tuple(views)
