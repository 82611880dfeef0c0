"""Two-level VQ-VAE-2 over GANSynth-style spectrograms, MI355X-native.

Drop-in for the reference's `vqvae/vqvae.py:36-342` (`class VQVAE`): same
constructor keywords and JSON persistence, same `state_dict` keys, same method
names and return tuples:

    forward(x)        -> (dec, diff, perplexity_t, perplexity_b, id_t, id_b)   (:245-249)
    encode(x)         -> (quant_t, quant_b, diff, id_t, id_b, perp_t, perp_b)   (:251-278)
    decode(q_t, q_b)  -> dec                                                    (:280-286)
    decode_code(t, b) -> dec                                                    (:288-295)

`encode`/`decode`/`forward` run as ONE call into libisi_hip.so
(`isi_vqvae_run`): the whole launch sequence is enqueued natively on the
current HIP stream, activations stay channels-last in a caller-owned workspace.
`quant_t` / `quant_b` are returned as [B,D,H,W]-shaped views of channels-last
storage, exactly like the reference's `.permute(0, 3, 1, 2)` results.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import pathlib
import warnings
from typing import Iterable, List, Mapping, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn, Tensor

from .. import _hip
from .encoder_decoder import RosinalityEncoder, RosinalityDecoder, _ConvParams
from .bottleneck import QuantizedBottleneck, UnquantizedBottleneck

# ISI_CONV_F16X3 scales weights and code vectors by 2^10 before the f16 split (include/isi_hip.h): 65520 / 1024 and
# above rounds to inf
_F16_WEIGHT_LIMIT = 63.98


class VQVAE(nn.Module):
    n_embed_t: int
    n_embed_b: int

    def __init__(
        self,
        encoders: Optional[Mapping[str, nn.Module]] = None,
        decoders: Optional[Mapping[str, nn.Module]] = None,
        in_channel: int = 3,
        num_hidden_channels: int = 128,
        n_res_block: int = 2,
        num_residual_channels: int = 32,
        embed_dim: int = 64,
        num_embeddings: Union[int, Iterable[int]] = 512,
        decay: float = 0.99,
        groups: int = 1,
        use_local_kernels: bool = False,
        output_activation_type: Optional[str] = None,
        output_spectrogram_min_magnitude: Optional[float] = None,
        resolution_factors: Mapping[str, int] = {'bottom': 4, 'top': 2},
        embeddings_initial_variance: float = 1,
        decoder_output_activation: Optional[nn.Module] = None,
        normalizer_statistics: Optional[Mapping[str, float]] = None,
        corruption_weights: Mapping[str, Optional[List[float]]] = {'top': None, 'bottom': None},
        adapt_quantized_durations: bool = True,
        disable_quantization: bool = False,
        restarts_usage_threshold: float = 1.,
    ):
        if decoder_output_activation is not None:
            raise NotImplementedError("TODO")
        if encoders is not None or decoders is not None:
            raise NotImplementedError("custom encoder/decoder modules (the fastai XResNet variant) "
                                      "are outside the MI355X hot path")
        if restarts_usage_threshold != 1.:
            raise NotImplementedError("QuantizedBottleneckWithRestarts needs the absent `discretization` "
                                      "package and is not built")
        if output_activation_type is not None:
            raise NotImplementedError("decoder output activations ('threshold_gelu', vqvae.py:229-236) are not built")

        # instantiation parameters, JSON round-tripped like vqvae.py:98-122
        self.in_channel = in_channel
        self.num_hidden_channels = num_hidden_channels
        self.n_res_block = n_res_block
        self.num_residual_channels = num_residual_channels
        self.embed_dim = embed_dim
        self.use_local_kernels = use_local_kernels
        self.num_embeddings = num_embeddings
        self.decay = decay
        self.groups = groups
        self.resolution_factors = resolution_factors
        self.embeddings_initial_variance = embeddings_initial_variance
        self.output_activation_type = output_activation_type
        self.decoder_output_activation = decoder_output_activation
        self.corruption_weights = corruption_weights
        self.output_spectrogram_min_magnitude = output_spectrogram_min_magnitude
        self.restarts_usage_threshold = restarts_usage_threshold
        self.normalizer_statistics = normalizer_statistics
        self._instantiation_parameters = self.__dict__.copy()

        super().__init__()

        C_, R, D = num_hidden_channels, num_residual_channels, embed_dim
        fb, ft = resolution_factors['bottom'], resolution_factors['top']
        self.enc_b = RosinalityEncoder(in_channel, C_, n_res_block, R, resolution_factor=fb,
                                       groups=groups, use_local_kernels=use_local_kernels)
        self.enc_t = RosinalityEncoder(C_, C_, n_res_block, R, resolution_factor=ft,
                                       groups=groups, use_local_kernels=use_local_kernels)
        if isinstance(num_embeddings, int):
            self.n_embed_t, self.n_embed_b = [num_embeddings] * 2
        else:
            self.n_embed_t, self.n_embed_b = num_embeddings

        self.quantize_conv_t = _ConvParams(C_, D, 1)
        bottleneck = UnquantizedBottleneck if disable_quantization else QuantizedBottleneck
        self.disable_quantization = disable_quantization
        self.quantize_t = bottleneck(D, self.n_embed_t, decay=decay,
                                     corruption_weights=corruption_weights['top'],
                                     embeddings_initial_variance=embeddings_initial_variance)
        self.dec_t = RosinalityDecoder(D, D, C_, n_res_block, R, groups=groups, resolution_factor=ft,
                                       use_local_kernels=use_local_kernels)
        self.quantize_conv_b = _ConvParams(D + C_, D, 1)
        self.quantize_b = bottleneck(D, self.n_embed_b, decay=decay,
                                     corruption_weights=corruption_weights['bottom'],
                                     embeddings_initial_variance=embeddings_initial_variance)
        n_up = int(np.log2(ft))
        self.upsample_top_to_bottom = nn.ModuleList(
            [_ConvParams(D, D, 4, stride=2, padding=1, transposed=True) for _ in range(n_up)])
        self.dec = RosinalityDecoder(D + D, in_channel, C_, n_res_block, R, resolution_factor=fb,
                                     groups=groups, use_local_kernels=use_local_kernels)

        # GANSynth range normalisation of the input / de-normalisation of the output and the masked-phase
        # output transform (vqvae.py:218-241): one fused HBM pass each (GANsynth_pytorch/normalizer.py)
        self.use_gansynth_normalization = normalizer_statistics is not None
        self.data_normalizer = None
        self.output_transform = None
        if self.use_gansynth_normalization:
            from GANsynth_pytorch.normalizer import DataNormalizer, DataNormalizerStatistics
            self.data_normalizer = DataNormalizer(DataNormalizerStatistics(**normalizer_statistics))
        if output_spectrogram_min_magnitude is not None:
            from GANsynth_pytorch.loader import make_masked_phase_transform
            self.output_transform = make_masked_phase_transform(output_spectrogram_min_magnitude)
        self.adapt_quantized_durations = adapt_quantized_durations
        self._plan = None
        self._plan_key = None
        # Arithmetic of the convolutions' products (data and accumulation are always fp32):
        #   'split_f16' (default)  every product as a sum of f16 pieces on the f16 matrix pipe (16x the fp32 pipe's
        #        rate): x = hi + lo, two 11-bit pieces of the operand scaled by a power of two, THREE terms
        #        (hi.hi + hi.lo + lo.hi).  Per-product relative error ~2^-23; measured against fp64 on the 3x3
        #        128->128 layer: 0.9e-6 of the maximum (fp32 pipe 1.2e-6, six-term bf16 split 1.2e-6, torch-CPU
        #        2.8e-7), code indices agree with the CPU reference as often as the fp32 pipe's do.  Inside f16's
        #        range: activations |x| < 16384, weights |w| < 64.  The weights are checked when the plan is
        #        built (a model beyond the range runs in 'split_bf16'); an activation beyond it turns into NaN in
        #        `dec` / `diff` and into code index -1, never into a silently wrong value.
        #   'split_bf16'  products as sums of bf16 pieces (fp32's exponent range, no operand limits):
        #        * every layer that feeds a code index (encoders, quantiser 1x1s, top decoder): SIX-term split
        #          (x = hi + mid + lo exactly; hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid = every term above
        #          2^-24 of a product).  Measured against fp64 its error is at or below the fp32 pipe's own
        #          (1.2e-6 vs 1.4e-6 of the maximum on the 3x3 128->128 layer; torch-CPU: 2.8e-7), and its code
        #          indices agree with the CPU reference as often as the fp32 pipe's do (near-ties only);
        #        * `upsample_top_to_bottom` and the final decoder `dec` (no index depends on them): THREE-term
        #          split (hi.hi + hi.lo + lo.hi): reconstruction within 1e-5 of its maximum (north_star: 1e-3)
        #   'bf16x3_decoder'  index-feeding layers on the exact-fp32 matrix pipe, decoder three-term split
        #   'f32'             everything on the exact-fp32 matrix pipe
        #   'bf16x3'          every convolution three-term split: near-tie code indices move (not parity-safe)
        self.conv_precision = os.environ.get("ISI_CONV_PRECISION", "split_f16")

    # ------------------------------------------------------------ native plan
    def invalidate_plan(self) -> None:
        """Drop every cached packed weight, codebook and the native plan.  The caches are keyed on each tensor's
        autograd version and address, which in-place writes through `.data` (`p.data.copy_`, weight averaging) do
        not change: call this after such a write.  `load_state_dict`, optimiser steps, `.to()` and the EMA update
        bump the version (or reset the keys) themselves."""
        self._plan, self._plan_key = None, None
        for m in self.modules():
            if hasattr(m, "_packed_key"):
                m._packed_key = None
                if hasattr(m, "_packed"):
                    m._packed = None

    def _plan_fingerprint(self):
        key = []
        for t in list(self.parameters()) + [self.quantize_t.embed, self.quantize_b.embed]:
            key.append((_hip.version_of(t), t.data_ptr()))
        key.append((getattr(self.quantize_t, "_ema_steps", 0), getattr(self.quantize_b, "_ema_steps", 0)))
        key.append(self.conv_precision)
        return tuple(key)

    def _native_weights(self) -> _hip.isi_vqvae_w:
        """isi_vqvae_w describing the packed weights (rebuilt when any parameter changes)."""
        key = self._plan_fingerprint()
        if self._plan is not None and self._plan_key == key:
            return self._plan[0]
        keep = []

        wmax = []

        def conv(m: _ConvParams) -> _hip.isi_conv_w:
            p = m.packed()
            keep.append(p)
            wmax.append(m.weight.detach().abs().max())
            return _hip.isi_conv_w(p.data_ptr(), m.bias.data_ptr(), m.in_channels, m.out_channels)

        def conv1x1_with_fragments(m: _ConvParams) -> _hip.isi_conv_w:
            """quantize_conv_*: {packed | pair copy | fragment-major copy} (isi_vqvae_w.w16 = 2): the fused search then needs no
            pre-kernel per call (csrc/vq_nearest.hip: the fragments are a pack-time job)."""
            p2 = m.packed()
            n = p2.numel() // 2
            kpad = n // m.out_channels
            p3 = torch.empty(3 * n, dtype=torch.float32, device=p2.device)
            p3[:2 * n].copy_(p2)
            _hip.check(_hip.lib().isi_vq_pack_fragments_f32(p3.data_ptr() + 4 * n, p3.data_ptr() + 8 * n, kpad,
                                                            C.c_void_p(_hip.stream_ptr(p2.device))), "isi_vq_pack_fragments_f32")
            keep.append(p3)
            wmax.append(m.weight.detach().abs().max())
            return _hip.isi_conv_w(p3.data_ptr(), m.bias.data_ptr(), m.in_channels, m.out_channels)

        def res(stack, blocks, idxs):
            if len(idxs) > _hip.ISI_MAX_RES:
                raise NotImplementedError(f"more than {_hip.ISI_MAX_RES} residual blocks")
            stack.n_res = len(idxs)
            for j, i in enumerate(idxs):
                stack.res3[j] = conv(blocks[i].conv[1])
                stack.res1[j] = conv(blocks[i].conv[3])

        def enc(m: RosinalityEncoder) -> _hip.isi_encoder_w:
            e = _hip.isi_encoder_w()
            e.n_down = len(m._down)
            for j, i in enumerate(m._down):
                e.down[j] = conv(m.blocks[i])
            e.conv3 = conv(m.blocks[m._conv3])
            res(e, m.blocks, m._res)
            return e

        def dec(m: RosinalityDecoder) -> _hip.isi_decoder_w:
            d = _hip.isi_decoder_w()
            d.conv3 = conv(m.blocks[0])
            res(d, m.blocks, m._res)
            d.n_up = len(m._up)
            for j, i in enumerate(m._up):
                d.up[j] = conv(m.blocks[i])
            return d

        def book(q: QuantizedBottleneck) -> _hip.isi_codebook_w:
            if self.disable_quantization:     # UnquantizedBottleneck: no search, the codebook is never read
                return _hip.isi_codebook_w(None, None, q.dim, q.n_embed)
            codes, e2 = q.packed()
            keep.extend([codes, e2])
            # The search splits the code vectors like weights, but a FINITE code beyond that range (every trained
            # codebook has them: the EMA update divides an unused code by a vanishing cluster size) is handled
            # exactly by the kernel (vq_decide_f32: certificate or fp32 scan); only a non-finite code disqualifies.
            e = q.embed.detach()
            wmax.append(torch.where(torch.isfinite(e).all(), e.new_zeros(()), e.new_full((), float("inf"))))
            return _hip.isi_codebook_w(codes.data_ptr(), e2.data_ptr(), q.dim, q.n_embed)

        w = _hip.isi_vqvae_w()
        w.in_channel = self.in_channel
        w.enc_b, w.enc_t = enc(self.enc_b), enc(self.enc_t)
        frag = all(m.kernel_size == 1 and m.groups == 1 and m.out_channels == 64
                   and (m.packed().numel() // 2 // m.out_channels) % 16 == 0 for m in (self.quantize_conv_t, self.quantize_conv_b))
        qconv = conv1x1_with_fragments if frag else conv
        w.quantize_conv_t, w.quantize_conv_b = qconv(self.quantize_conv_t), qconv(self.quantize_conv_b)
        w.quantize_t, w.quantize_b = book(self.quantize_t), book(self.quantize_b)
        w.dec_t, w.dec = dec(self.dec_t), dec(self.dec)
        w.precision = {"f32": 0, "bf16x3_decoder": 1, "bf16x3": 2, "split_bf16": 3, "split_f16": 4}[self.conv_precision]
        w.w16 = 2 if frag else 1          # _ConvParams.packed() carries the split-f16 pair copies (2: + the quantiser weights' fragments)
        w.no_quantize = 1 if self.disable_quantization else 0
        w.n_upsample = len(self.upsample_top_to_bottom)
        for j, m in enumerate(self.upsample_top_to_bottom):
            w.upsample[j] = conv(m)
        if w.precision == 4 and not float(torch.stack(wmax).max()) < _F16_WEIGHT_LIMIT:
            warnings.warn(f"a convolution weight reaches {_F16_WEIGHT_LIMIT:g} in magnitude (or a weight / code vector is not finite): "
                          "beyond the operand range of conv_precision='split_f16', running this model in 'split_bf16'")
            w.precision = 3
        self._plan, self._plan_key = (w, keep), key
        return w

    def _total_factor(self) -> Tuple[int, int]:
        return self.resolution_factors['bottom'], self.resolution_factors['top']

    def _latent_shapes(self, H: int, W: int):
        def down(x, n):
            for _ in range(n):
                x = (x + 2 - 4) // 2 + 1
            return x
        nb, nt = len(self.enc_b._down), len(self.enc_t._down)
        Hb, Wb = down(H, nb), down(W, nb)
        Ht, Wt = down(Hb, nt), down(Wb, nt)
        Wq = min(Wt * 2 ** nt, Wb) if self.adapt_quantized_durations else Wb
        return Hb, Wb, Ht, Wt, Wq

    def _run(self, mode: int, x: Optional[Tensor], B: int, H: int, W: int, out: _hip.isi_vqvae_out,
             device: torch.device):
        if not self.adapt_quantized_durations:
            Hb, Wb, Ht, Wt, Wq = self._latent_shapes(H, W)
            if Wq != Wt * 2 ** len(self.enc_t._down):
                raise RuntimeError("Sizes of tensors must match except in dimension 1 "
                                   "(adapt_quantized_durations=False with an odd bottom width)")
        w = self._native_weights()
        L = _hip.lib()
        nbytes = L.isi_vqvae_workspace_bytes(C.byref(w), B, H, W)
        if nbytes == 0:
            raise _hip.HipLibraryError(f"unsupported input shape [B={B}, H={H}, W={W}] for this VQVAE")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        rc = L.isi_vqvae_run(C.byref(w), mode, x.data_ptr() if x is not None else None, B, H, W,
                             C.byref(out), ws.data_ptr(), nbytes, C.c_void_p(_hip.stream_ptr(device)))
        _hip.check(rc, "isi_vqvae_run")
        # `ws` may be recycled by the caching allocator once this returns: it is
        # only reused by work enqueued later on the same stream.

    # ---------------------------------------------------------------- API
    @torch.no_grad()
    def _encode_impl(self, input: Tensor, with_decode: bool, need_quant: bool = True):
        _hip.require_gpu(input, "input")
        if input.dim() != 4 or input.shape[1] != self.in_channel:
            raise RuntimeError(f"expected input [B, {self.in_channel}, H, W], got {tuple(input.shape)}")
        x = input.contiguous()
        if self.data_normalizer is not None:
            x = self.data_normalizer.normalize(x)          # vqvae.py:254-255
        B, _, H, W = x.shape
        Hb, Wb, Ht, Wt, Wq = self._latent_shapes(H, W)
        dev, D = x.device, self.embed_dim
        f32 = dict(dtype=torch.float32, device=dev)
        # `need_quant` False (forward(): the caller takes dec, diff, ids, perplexities): the library then keeps the quantised
        # maps in the pair format its decoders read and writes no fp32 copy (the fused search's stores are what bounds it)
        need_quant = need_quant or not with_decode or self.disable_quantization
        quant_t = torch.empty(B, Ht, Wt, D, **f32) if need_quant else None
        quant_b = torch.empty(B, Hb, Wq, D, **f32) if need_quant else None
        id_t = torch.empty(B, Ht, Wt, dtype=torch.int64, device=dev)
        id_b = torch.empty(B, Hb, Wq, dtype=torch.int64, device=dev)
        scalars = torch.empty(5, **f32)
        dec = None
        if with_decode:
            fb = 2 ** len(self.dec._up)
            dec = torch.empty(B, self.in_channel, Hb * fb, Wq * fb, **f32)
        out = _hip.isi_vqvae_out(dec.data_ptr() if dec is not None else None, quant_t.data_ptr() if need_quant else None,
                                 quant_b.data_ptr() if need_quant else None, id_t.data_ptr(), id_b.data_ptr(), scalars.data_ptr())
        self._run(_hip.MODE_FORWARD if with_decode else _hip.MODE_ENCODE, x, B, H, W, out, dev)
        if self.disable_quantization:
            # UnquantizedBottleneck.forward (bottleneck.py:107-119): diff zeros(1) each -> unsqueeze(0) and summed
            # (vqvae.py:263,275,277): [1, 1]; no indices; perplexity tensor([inf])
            return (dec, quant_t.permute(0, 3, 1, 2), quant_b.permute(0, 3, 1, 2), scalars[4].reshape(1, 1),
                    None, None, scalars[1:2], scalars[3:4])
        diff = scalars[4].reshape(1)  # diff_t.unsqueeze(0) + diff_b.unsqueeze(0), vqvae.py:263,275,277
        return (dec, quant_t.permute(0, 3, 1, 2) if need_quant else None, quant_b.permute(0, 3, 1, 2) if need_quant else None,
                diff, id_t, id_b, scalars[1], scalars[3])

    def forward(self, input: Tensor):
        if self.training:
            return self._forward_train(input)
        dec, _, _, diff, id_t, id_b, perp_t, perp_b = self._encode_impl(input, with_decode=True, need_quant=False)
        return self.post_process(dec), diff, perp_t, perp_b, id_t, id_b

    def _forward_train(self, input: Tensor):
        """Train-mode forward (EMA codebook update in-forward, bottleneck.py:79-92) whose
        `dec` / `diff` outputs are differentiable w.r.t. every parameter through a
        hand-written backward (`vqvae/_train.py`)."""
        from ._train import VQVAETrainFunction, _DgradWeights
        _hip.require_gpu(input, "input")
        if not hasattr(self, "_dgrad_weights"):
            self._dgrad_weights = _DgradWeights()
        if self.data_normalizer is not None:
            input = self.data_normalizer.normalize(input)
        dec, diff, perp_t, perp_b, id_t, id_b = VQVAETrainFunction.apply(self, input, *self.parameters())
        if self.disable_quantization:   # UnquantizedBottleneck: diff [1, 1] zeros, perplexities tensor([inf]), no indices
            diff, perp_t, perp_b, id_t, id_b = diff.reshape(1, 1), perp_t.reshape(1), perp_b.reshape(1), None, None
        return self.post_process(dec), diff, perp_t, perp_b, id_t, id_b      # differentiable (SpecAffineMaskFunction)

    def encode(self, input: Tensor):
        if self.training and not self.disable_quantization:
            # train mode: the quantisers update their EMA buffers in-forward (bottleneck.py:79-92), like the
            # reference's encode under model.train(); outputs are returned detached (gradients flow through
            # forward(), the path train_vqvae.py uses)
            from ._train import encode_train
            _hip.require_gpu(input, "input")
            if self.data_normalizer is not None:
                input = self.data_normalizer.normalize(input)
            return encode_train(self, input)
        _, q_t, q_b, diff, id_t, id_b, perp_t, perp_b = self._encode_impl(input, with_decode=False)
        return q_t, q_b, diff, id_t, id_b, perp_t, perp_b

    @torch.no_grad()
    def decode(self, quant_t: Tensor, quant_b: Tensor) -> Tensor:
        """quant_t [B,D,Ht,Wt], quant_b [B,D,Hb,Wb] (any strides; channels-last
        storage, as returned by `encode`, is consumed without a copy)."""
        _hip.require_gpu(quant_t, "quant_t")
        _hip.require_gpu(quant_b, "quant_b")
        qt = quant_t.permute(0, 2, 3, 1).contiguous()
        qb = quant_b.permute(0, 2, 3, 1).contiguous()
        B, Ht, Wt, D = qt.shape
        _, Hb, Wb, _ = qb.shape
        fb = 2 ** len(self.dec._up)
        H, W = Hb * fb, Wb * fb
        if self._latent_shapes(H, W) != (Hb, Wb, Ht, Wt, Wb):
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1: top {Ht}x{Wt} "
                               f"does not upsample to bottom {Hb}x{Wb}")
        dec = torch.empty(B, self.in_channel, H, W, dtype=torch.float32, device=qt.device)
        out = _hip.isi_vqvae_out(dec.data_ptr(), qt.data_ptr(), qb.data_ptr(), None, None, None)
        self._run(_hip.MODE_DECODE, None, B, H, W, out, qt.device)
        return self.post_process(dec)

    def decode_code(self, code_t: Tensor, code_b: Tensor) -> Tensor:
        quant_t = self.quantize_t.embed_code(code_t).permute(0, 3, 1, 2)
        quant_b = self.quantize_b.embed_code(code_b).permute(0, 3, 1, 2)
        return self.decode(quant_t, quant_b)

    def post_process(self, dec: Tensor) -> Tensor:
        """vqvae.py:297-302: de-normalise, then zero the phase of bins at or below the magnitude floor
        (one fused pass when both are configured)."""
        thr = self.output_spectrogram_min_magnitude
        if self.data_normalizer is not None:
            return self.data_normalizer.denormalize(dec, threshold=thr)
        if self.output_transform is not None:
            return self.output_transform(dec)
        return dec

    # ------------------------------------------------------------ persistence
    @classmethod
    def from_parameters_and_weights(cls, parameters_json_path: pathlib.Path,
                                    model_weights_checkpoint_path: pathlib.Path,
                                    device: Union[str, torch.device] = 'cpu',
                                    encoders=None, decoders=None) -> 'VQVAE':
        """vqvae.py:304-337.  Unlike the reference (strict=False, silently a
        no-op on DDP-saved checkpoints), a `module.` prefix is stripped and
        missing / unexpected keys are reported."""
        with open(parameters_json_path, 'r') as f:
            parameters = json.load(f)
        vqvae = cls(**parameters, encoders=encoders, decoders=decoders)
        sd = torch.load(model_weights_checkpoint_path, map_location=device)
        if 'model' in sd.keys():
            sd = sd['model']
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        result = vqvae.load_state_dict(sd, strict=False)
        if result.missing_keys or result.unexpected_keys:
            warnings.warn(f"VQVAE checkpoint mismatch: missing {result.missing_keys}, "
                          f"unexpected {result.unexpected_keys}")
        return vqvae

    def store_instantiation_parameters(self, path: pathlib.Path) -> None:
        with open(path, 'w') as f:
            json.dump(self._instantiation_parameters, f, indent=4)
