"""Rosinality-style convolutional encoder / decoder stacks on MI355X.

Drop-in for the reference's `vqvae/encoder_decoder.py:18-227`
(`RosinalityResBlock`, `RosinalityEncoder`, `RosinalityDecoder`): same
constructor signatures, same `state_dict` keys (`blocks.<i>.weight`,
`blocks.<i>.conv.{1,3}.weight`, ...), same weight layouts and default
initialisation.  The modules own parameters only; `forward` enqueues the HIP
kernels of libisi_hip.so (implicit-GEMM convolutions on the exact-fp32 matrix
pipe).  The fastai XResNet/U-Net variant (`:230-447`) is out of scope.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import nn

from .. import _hip
from . import _ops


class _Slot(nn.Module):
    """Parameter-less placeholder keeping the reference's nn.Sequential
    numbering (its nn.ReLU entries), so state_dict keys line up."""

    def forward(self, *a, **k):  # pragma: no cover - never part of a compute path
        raise RuntimeError("placeholder module; the owning stack fuses this ReLU into its kernels")


class _ConvParams(nn.Module):
    """Weights of one (transposed) convolution, initialised like torch.nn's
    Conv2d / ConvTranspose2d (kaiming_uniform(a=sqrt(5)) + fan-in bias bound).
    Holds parameters only; packing for the HIP kernels is cached per version."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int,
                 stride: int = 1, padding: int = 0, transposed: bool = False, groups: int = 1):
        super().__init__()
        if groups < 1 or in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")  # torch.nn's check
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.transposed = transposed
        self.groups = groups
        # torch.nn layouts: Conv2d [Cout, Cin/groups, k, k], ConvTranspose2d [Cin, Cout/groups, k, k]
        shape = ((in_channels, out_channels // groups) if transposed else (out_channels, in_channels // groups))
        self.weight = nn.Parameter(torch.empty(*shape, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.weight.shape[1] * kernel_size * kernel_size
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)
        self._packed = None
        self._packed_key = None

    def dense_weight(self) -> torch.Tensor:
        """The weight as the kernels consume it (detached): a grouped convolution (encoder_decoder.py:58-215,
        `groups=groups`) is run as the dense convolution with the block-diagonal weight -- group g maps
        its slice of the input channels to its slice of the output channels, every other product is an
        exact zero -- so the implicit-GEMM kernels need no grouped variant."""
        w = self.weight.detach()
        if self.groups == 1:
            return w
        rows = w.shape[0] // self.groups          # Conv2d: output channels / group; ConvTranspose2d: input channels
        cols = w.shape[1]
        dense = w.new_zeros(w.shape[0], cols * self.groups, *w.shape[2:])
        for g in range(self.groups):
            dense[g * rows:(g + 1) * rows, g * cols:(g + 1) * cols] = w[g * rows:(g + 1) * rows]
        return dense

    def grouped(self, dense: torch.Tensor) -> torch.Tensor:
        """Block-diagonal part of a dense-layout tensor (a weight gradient) in the parameter's layout."""
        if self.groups == 1:
            return dense
        rows, cols = dense.shape[0] // self.groups, dense.shape[1] // self.groups
        return torch.cat([dense[g * rows:(g + 1) * rows, g * cols:(g + 1) * cols] for g in range(self.groups)], 0)

    def packed(self) -> torch.Tensor:
        """Packed weight followed by its split-f16 pair copy (the layout ISI_CONV_W16 / isi_vqvae_w.w16 expect;
        entry points called without that flag read the first half only)."""
        key = (_hip.version_of(self.weight), self.weight.data_ptr(), self.weight.device)
        if self._packed is None or self._packed_key != key:
            if self.transposed:
                if (self.kernel_size, self.stride, self.padding) != (4, 2, 1):
                    raise NotImplementedError("only ConvTranspose2d(k=4, s=2, p=1) is built "
                                              "(use_local_kernels=True is not)")
                self._packed = _ops.pack_convT_weight(self.dense_weight(), with_f16=True)
            else:
                self._packed = _ops.pack_conv_weight(self.dense_weight(), with_f16=True)
            self._packed_key = key
        return self._packed

    def extra_repr(self):
        kind = "ConvTranspose2d" if self.transposed else "Conv2d"
        return (f"{kind}({self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, padding={self.padding}" + (f", groups={self.groups})" if self.groups > 1 else ")"))

    def run(self, x, relu: bool, x2=None, residual=None, out_nchw: bool = False, bf16x3: bool = False):
        if self.transposed:
            return _ops.conv_transpose2d_k4s2(x, self.packed(), self.bias, self.out_channels, relu,
                                              out_nchw=out_nchw, bf16x3=bf16x3)
        return _ops.conv2d(x, self.packed(), self.bias, self.out_channels, self.kernel_size,
                           self.stride, self.padding, relu, x2_bchw=x2, residual_bchw=residual, bf16x3=bf16x3)


class RosinalityResBlock(nn.Module):
    """reference encoder_decoder.py:18-35.  The reference's first ReLU is
    in-place, so the block computes r + conv1x1(relu(conv3x3(r))) with
    r = relu(input).  Inside a stack the producer of `input` already rectified
    it; `forward_rectified` therefore takes r and can also rectify its output."""

    def __init__(self, in_channel: int, channel: int):
        super().__init__()
        self.conv = nn.ModuleList([
            _Slot(), _ConvParams(in_channel, channel, 3, padding=1),
            _Slot(), _ConvParams(channel, in_channel, 1)])

    def forward_rectified(self, r: torch.Tensor, relu_out: bool, fused: bool = True,
                          bf16x3: bool = False) -> torch.Tensor:
        c3, c1 = self.conv[1], self.conv[3]
        if fused and _ops.resblock_fusable(c3.in_channels, c3.out_channels):
            return _ops.resblock(r, c3.packed(), c3.bias, c1.packed(), c1.bias, c3.out_channels, relu_out,
                                 bf16x3=bf16x3)
        h = self.conv[1].run(r, relu=True)
        return self.conv[3].run(h, relu=relu_out, residual=r)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # the in-place ReLU of the reference is reproduced on the caller's tensor
        _ops.relu_(input)
        return self.forward_rectified(input, relu_out=False)


# The reference keeps padding=1 with its k=2 "local" kernels (encoder_decoder.py:46-51,146-151), so a stage maps
# H -> H/2 + 1 going down and H -> 2H - 2 going up: VQVAE.encode then fails on torch.cat([dec_t, enc_b])
# (vqvae.py:264-271; probed: "Sizes of tensors must match ... Expected size 8 but got size 9").  The option
# cannot run in the reference's own model, so there is no behaviour to reproduce.
_LOCAL_KERNELS = ("use_local_kernels=True is not built: with the reference's padding=1 the k=2 stages give "
                  "H/2+1 and 2H-2 sized maps and the reference VQVAE.encode itself fails on the top/bottom concat")


def _down_channels(in_channel: int, channel: int, factor: int):
    table = {16: [in_channel, channel // 4, channel // 2, 3 * channel // 4, channel],
             8: [in_channel, channel // 2, channel // 2, channel],
             4: [in_channel, channel // 2, channel],
             2: [in_channel, channel // 2]}
    if factor not in table:
        raise ValueError(f"Unexpected resolution factor {factor}")
    chans = table[factor]
    return list(zip(chans[:-1], chans[1:]))


def _up_channels(channel: int, out_channel: int, factor: int):
    table = {16: [channel, 3 * channel // 4, channel // 2, channel // 4, out_channel],
             8: [channel, channel // 2, channel // 2, out_channel],
             4: [channel, channel // 2, out_channel],
             2: [channel, out_channel]}
    if factor not in table:
        raise ValueError(f"Unexpected resolution factor {factor}")
    chans = table[factor]
    return list(zip(chans[:-1], chans[1:]))


class RosinalityEncoder(nn.Module):
    """reference encoder_decoder.py:38-126."""

    def __init__(self, in_channel: int, channel: int, n_res_block: int, n_res_channel: int,
                 resolution_factor: int, groups: int = 1, use_local_kernels: bool = False):
        super().__init__()
        if use_local_kernels:
            raise NotImplementedError(_LOCAL_KERNELS)
        self.use_local_kernels = use_local_kernels
        self.resolution_factor = resolution_factor
        blocks: List[nn.Module] = []
        self._down: List[int] = []
        last = in_channel
        for (a, b) in _down_channels(in_channel, channel, resolution_factor):
            self._down.append(len(blocks))
            blocks += [_ConvParams(a, b, 4, stride=2, padding=1, groups=groups), _Slot()]
            last = b
        self._conv3 = len(blocks)
        blocks.append(_ConvParams(last, channel, 3, padding=1, groups=groups))
        self._res: List[int] = []
        for _ in range(n_res_block):
            self._res.append(len(blocks))
            blocks.append(RosinalityResBlock(channel, n_res_channel))
        blocks.append(_Slot())
        self.blocks = nn.ModuleList(blocks)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        x = input
        for i in self._down:
            x = self.blocks[i].run(x, relu=True)
        x = self.blocks[self._conv3].run(x, relu=True)
        for i in self._res:
            x = self.blocks[i].forward_rectified(x, relu_out=True)
        return x


class RosinalityDecoder(nn.Module):
    """reference encoder_decoder.py:129-227."""

    def __init__(self, in_channel: int, out_channel: int, channel: int, n_res_block: int,
                 n_res_channel: int, resolution_factor: int, groups: int = 1,
                 use_local_kernels: bool = False, output_activation: Optional[nn.Module] = None):
        super().__init__()
        if use_local_kernels:
            raise NotImplementedError(_LOCAL_KERNELS)
        if output_activation is not None:
            raise NotImplementedError("decoder output activations are not built (always None in the reference, "
                                      "vqvae.py:95-96)")
        self.use_local_kernels = use_local_kernels
        self.resolution_factor = resolution_factor
        blocks: List[nn.Module] = [_ConvParams(in_channel, channel, 3, padding=1)]
        self._res: List[int] = []
        for _ in range(n_res_block):
            self._res.append(len(blocks))
            blocks.append(RosinalityResBlock(channel, n_res_channel))
        blocks.append(_Slot())
        self._up: List[int] = []
        ups = _up_channels(channel, out_channel, resolution_factor)
        for j, (a, b) in enumerate(ups):
            self._up.append(len(blocks))
            blocks.append(_ConvParams(a, b, 4, stride=2, padding=1, transposed=True, groups=groups))
            if j != len(ups) - 1:
                blocks.append(_Slot())
        self.blocks = nn.ModuleList(blocks)

    def forward(self, input: torch.Tensor, input2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`input2`, when given, is concatenated after `input` on channels
        (the torch.cat of vqvae.py:282) without materialising the concat."""
        x = self.blocks[0].run(input, relu=True, x2=input2)
        for i in self._res:
            x = self.blocks[i].forward_rectified(x, relu_out=True)
        for j, i in enumerate(self._up):
            x = self.blocks[i].run(x, relu=(j != len(self._up) - 1))
        return x
