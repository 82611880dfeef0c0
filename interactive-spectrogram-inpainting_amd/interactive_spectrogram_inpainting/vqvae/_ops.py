"""Thin Python wrappers over the per-operator C-ABI entry points.

Tensors are channels-last `[B,H,W,C]` fp32 unless a name says otherwise.  These
helpers only allocate outputs (torch = device memory owner) and marshal
pointers; all arithmetic happens in libisi_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _hip


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


def _with_f16_copy(out: torch.Tensor, n: int) -> torch.Tensor:
    """Fill out[n:2n] with the split-f16 pair copy of the packed weight out[:n] (ISI_CONV_W16)."""
    _hip.check(_hip.lib().isi_split_conv_weight_f16(out.data_ptr(), out.data_ptr() + 4 * n, n, _s(out)),
               "isi_split_conv_weight_f16")
    return out


def pack_conv_weight(weight: torch.Tensor, with_f16: bool = False) -> torch.Tensor:
    """torch Conv2d weight [Cout,Cin,KH,KW] -> packed [Cout, Kpad] (`with_f16`: followed by its split-f16 pair
    copy, the layout ISI_CONV_W16 expects)."""
    _hip.require_gpu(weight, "conv weight")
    w = weight.detach().contiguous()
    cout, cin, kh, kw = w.shape
    n = _hip.lib().isi_packed_conv_weight_floats(cout, cin, kh, kw)
    out = torch.empty(2 * n if with_f16 else n, dtype=torch.float32, device=w.device)
    if with_f16:      # packed weight and its pair copy in one launch
        _hip.check(_hip.lib().isi_pack_conv_weight_w16_f32(w.data_ptr(), out.data_ptr(), cout, cin, kh, kw, _s(w)),
                   "isi_pack_conv_weight_w16_f32")
        return out
    _hip.check(_hip.lib().isi_pack_conv_weight_f32(w.data_ptr(), out.data_ptr(), cout, cin, kh, kw, _s(w)),
               "isi_pack_conv_weight_f32")
    return out


def pack_conv_dgrad_weight(weight: torch.Tensor) -> torch.Tensor:
    """torch Conv2d (stride 1) weight [Cout,Cin,KH,KW] -> packed weight [Cin, Kpad] of its input-gradient convolution
    (= pack_conv_weight(weight.flip(2, 3).transpose(0, 1)) in one launch)."""
    _hip.require_gpu(weight, "conv weight")
    w = weight.detach().contiguous()
    cout, cin, kh, kw = w.shape
    out = torch.empty(_hip.lib().isi_packed_conv_weight_floats(cin, cout, kh, kw), dtype=torch.float32, device=w.device)
    _hip.check(_hip.lib().isi_pack_conv_dgrad_weight_f32(w.data_ptr(), out.data_ptr(), cout, cin, kh, kw, _s(w)),
               "isi_pack_conv_dgrad_weight_f32")
    return out


def pack_convT_weight(weight: torch.Tensor, with_f16: bool = False) -> torch.Tensor:
    """torch ConvTranspose2d(k4,s2,p1) weight [Cin,Cout,4,4] -> 4 packed phase matrices."""
    _hip.require_gpu(weight, "convT weight")
    w = weight.detach().contiguous()
    cin, cout, kh, kw = w.shape
    if (kh, kw) != (4, 4):
        raise NotImplementedError("only ConvTranspose2d(kernel 4, stride 2, padding 1) is built")
    n = _hip.lib().isi_packed_convT_k4s2_weight_floats(cin, cout)
    out = torch.empty(2 * n if with_f16 else n, dtype=torch.float32, device=w.device)
    _hip.check(_hip.lib().isi_pack_convT_k4s2_weight_f32(w.data_ptr(), out.data_ptr(), cin, cout, _s(w)),
               "isi_pack_convT_k4s2_weight_f32")
    return _with_f16_copy(out, n) if with_f16 else out


def pack_codebook(embed: torch.Tensor):
    """`embed` buffer [D,K] -> (codes [K,D], e2 [K])."""
    _hip.require_gpu(embed, "embed")
    e = embed.detach().contiguous()
    d, k = e.shape
    codes = torch.empty(k, d, dtype=torch.float32, device=e.device)
    e2 = torch.empty(k, dtype=torch.float32, device=e.device)
    _hip.check(_hip.lib().isi_pack_codebook_f32(e.data_ptr(), codes.data_ptr(), e2.data_ptr(), d, k, _s(e)),
               "isi_pack_codebook_f32")
    return codes, e2


def _pair_storage(x: torch.Tensor):
    """The pair format groups 8 consecutive CHANNELS in storage order (csrc/split_f16.h): a [B,C,H,W]-shaped view of
    channels-last storage (what the convolutions return) is processed as [B,H,W,C]; anything else as a dense tensor
    whose last dimension is the channel axis.  Returns (dense tensor, function mapping a result back to x's view)."""
    if x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1), lambda y: y.permute(0, 3, 1, 2)
    return x.contiguous(), lambda y: y


def pair_encode(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> split-f16 pair format (ISI_CONV_IN*_PAIR / OUT_PAIR): {hi[8] | lo[8]} per group of 8 channels."""
    _hip.require_gpu(x, "pair_encode input")
    d, back = _pair_storage(x)
    if d.shape[-1] % 8:
        raise ValueError("pair format needs a multiple of 8 channels")
    out = torch.empty_like(d)
    _hip.check(_hip.lib().isi_pair_encode_f32(d.data_ptr(), out.data_ptr(), d.numel(), _s(d)), "isi_pair_encode_f32")
    return back(out)


def pair_decode(p: torch.Tensor) -> torch.Tensor:
    _hip.require_gpu(p, "pair_decode input")
    d, back = _pair_storage(p)
    out = torch.empty_like(d)
    _hip.check(_hip.lib().isi_pair_decode_f32(d.data_ptr(), out.data_ptr(), d.numel(), _s(d)), "isi_pair_decode_f32")
    return back(out)


PAIR_IN0, PAIR_IN1, PAIR_OUT = 32, 64, 128   # ISI_CONV_IN0_PAIR / IN1_PAIR / OUT_PAIR, OR-ed into `extra_flags`


def _prec_flag(bf16x3) -> int:
    """False / 0 -> exact fp32, True / 1 -> ISI_CONV_BF16X3, 2 -> ISI_CONV_BF16X6, 3 -> ISI_CONV_F16X3,
    4 -> ISI_CONV_F16X3 | ISI_CONV_W16 (the packed weights carry their split-f16 pair copy, `with_f16=True`)."""
    return {0: 0, 1: 2, 2: 4, 3: 8, 4: 24}[int(bf16x3)]


def _check_gate(gate_nhwc: torch.Tensor, out_bchw: torch.Tensor) -> None:
    """The gate of a gated convolution must be laid out exactly like the (channels-last) output."""
    want = out_bchw.permute(0, 2, 3, 1)
    if (gate_nhwc.shape != want.shape or gate_nhwc.stride() != want.stride() or gate_nhwc.dtype != torch.float32
            or gate_nhwc.device != out_bchw.device):
        raise ValueError(f"gate {tuple(gate_nhwc.shape)} / strides {gate_nhwc.stride()} does not match the "
                         f"output's channels-last layout {tuple(want.shape)} / {want.stride()}")


def conv2d(x_bchw: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
           k: int, stride: int, pad: int, relu: bool, x2_bchw: Optional[torch.Tensor] = None,
           residual_bchw: Optional[torch.Tensor] = None, bf16x3: bool = False, extra_flags: int = 0,
           gate_nhwc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Convolution of a tensor indexed [B,C,H,W] (any strides); returns a
    [B,Cout,OH,OW]-shaped view of freshly allocated channels-last storage.  `gate_nhwc` (dense [B,OH,OW,Cout]):
    the output is zeroed where the gate is not positive (isi_conv2d_gated_f32: a ReLU's backward mask)."""
    _hip.require_gpu(x_bchw, "conv input")
    B, _, H, W = x_bchw.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    out = torch.empty(B, OH, OW, cout, dtype=torch.float32, device=x_bchw.device).permute(0, 3, 1, 2)
    s0 = _hip.src_nchw_view(x_bchw)
    s1 = _hip.src_nchw_view(x2_bchw) if x2_bchw is not None else None
    res = _hip.src_nchw_view(residual_bchw) if residual_bchw is not None else None
    dst = _hip.dst_nchw_view(out)
    if gate_nhwc is not None:
        _check_gate(gate_nhwc, out)
        rc = _hip.lib().isi_conv2d_gated_f32(
            C.byref(s0), C.byref(s1) if s1 is not None else None, packed_w.data_ptr(),
            bias.data_ptr() if bias is not None else None, C.byref(res) if res is not None else None,
            gate_nhwc.data_ptr(), C.byref(dst), B, H, W, cout, k, k, stride, pad,
            int(relu) | _prec_flag(bf16x3) | extra_flags, _s(x_bchw))
        _hip.check(rc, "isi_conv2d_gated_f32")
        return out
    rc = _hip.lib().isi_conv2d_f32(
        C.byref(s0), C.byref(s1) if s1 is not None else None, packed_w.data_ptr(),
        bias.data_ptr() if bias is not None else None, C.byref(res) if res is not None else None,
        C.byref(dst), B, H, W, cout, k, k, stride, pad, int(relu) | _prec_flag(bf16x3) | extra_flags, _s(x_bchw))
    _hip.check(rc, "isi_conv2d_f32")
    return out


def conv_transpose2d_k4s2(x_bchw: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor],
                          cout: int, relu: bool, out_nchw: bool = False, bf16x3: bool = False, extra_flags: int = 0,
                          gate_nhwc: Optional[torch.Tensor] = None) -> torch.Tensor:
    _hip.require_gpu(x_bchw, "convT input")
    B, _, H, W = x_bchw.shape
    if out_nchw:
        out = torch.empty(B, cout, 2 * H, 2 * W, dtype=torch.float32, device=x_bchw.device)
    else:
        out = torch.empty(B, 2 * H, 2 * W, cout, dtype=torch.float32, device=x_bchw.device).permute(0, 3, 1, 2)
    s0 = _hip.src_nchw_view(x_bchw)
    dst = _hip.dst_nchw_view(out)
    if gate_nhwc is not None:
        if out_nchw:
            raise ValueError("gated transposed convolution writes channels-last")
        _check_gate(gate_nhwc, out)
        rc = _hip.lib().isi_conv_transpose2d_k4s2_gated_f32(
            C.byref(s0), packed_w.data_ptr(), bias.data_ptr() if bias is not None else None, gate_nhwc.data_ptr(),
            C.byref(dst), B, H, W, cout, int(relu) | _prec_flag(bf16x3) | extra_flags, _s(x_bchw))
        _hip.check(rc, "isi_conv_transpose2d_k4s2_gated_f32")
        return out
    rc = _hip.lib().isi_conv_transpose2d_k4s2_f32(
        C.byref(s0), packed_w.data_ptr(), bias.data_ptr() if bias is not None else None,
        C.byref(dst), B, H, W, cout, int(relu) | _prec_flag(bf16x3) | extra_flags, _s(x_bchw))
    _hip.check(rc, "isi_conv_transpose2d_k4s2_f32")
    return out


def decoder_tail(x_pair_bchw: torch.Tensor, packed_w1: torch.Tensor, bias1: Optional[torch.Tensor], packed_w2: torch.Tensor,
                 bias2: Optional[torch.Tensor], cmid: int, cout: int) -> torch.Tensor:
    """ConvTranspose2d(Cin -> 64) + ReLU + ConvTranspose2d(64 -> cout <= 2) of the pair pipeline without the activation
    between them (isi_decoder_tail_f32): pair-format channels-last input viewed [B,Cin,H,W], packed weights with their
    split-f16 copies (`pack_convT_weight(..., with_f16=True)`), NCHW fp32 result [B,cout,4H,4W]."""
    _hip.require_gpu(x_pair_bchw, "decoder tail input")
    B, cin, H, W = x_pair_bchw.shape
    x = x_pair_bchw.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        raise ValueError("decoder_tail needs dense channels-last storage")
    out = torch.empty(B, cout, 4 * H, 4 * W, dtype=torch.float32, device=x.device)
    ws = torch.empty(B * 2 * H * 2 * W * 32, dtype=torch.float32, device=x.device)
    dst = _hip.dst_nchw_view(out)
    rc = _hip.lib().isi_decoder_tail_f32(x.data_ptr(), packed_w1.data_ptr(), bias1.data_ptr() if bias1 is not None else None,
                                         packed_w2.data_ptr(), bias2.data_ptr() if bias2 is not None else None,
                                         ws.data_ptr(), C.byref(dst), B, H, W, cin, cmid, cout, _s(x))
    _hip.check(rc, "isi_decoder_tail_f32")
    return out


def resblock_fusable(C_: int, R: int) -> bool:
    return bool(_hip.lib().isi_resblock_fusable(C_, R))


def resblock(r_bchw: torch.Tensor, packed_w3, b3, packed_w1, b1, R: int, relu: bool,
             bf16x3: bool = False, extra_flags: int = 0) -> torch.Tensor:
    """Fused residual block on a rectified, dense channels-last input viewed as [B,C,H,W]."""
    _hip.require_gpu(r_bchw, "resblock input")
    B, C_, H, W = r_bchw.shape
    nhwc = r_bchw.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    out = torch.empty_like(nhwc)
    rc = _hip.lib().isi_resblock_f32(nhwc.data_ptr(), packed_w3.data_ptr(), b3.data_ptr(), packed_w1.data_ptr(),
                                     b1.data_ptr(), out.data_ptr(), B, H, W, C_, R,
                                     int(relu) | _prec_flag(bf16x3) | extra_flags,
                                     _s(r_bchw))
    _hip.check(rc, "isi_resblock_f32")
    return out.permute(0, 3, 1, 2)


def vq_nearest(z: torch.Tensor, codes: torch.Tensor, e2: torch.Tensor, split_f16: bool = False):
    """z [..., D] dense channels-last -> (q_st [..., D], diff [], idx int64 [...], perplexity []).
    `split_f16`: z.e products as three split-f16 terms (ISI_CONV_F16X3) instead of the exact-fp32 matrix pipe."""
    _hip.require_gpu(z, "quantizer input")
    if not z.is_contiguous():
        z = z.contiguous()
    D = z.shape[-1]
    K = codes.shape[0]
    N = z.numel() // D
    L = _hip.lib()
    idx = torch.empty(z.shape[:-1], dtype=torch.int64, device=z.device)
    q = torch.empty_like(z)
    counts = torch.zeros(K, dtype=torch.int32, device=z.device)
    n_part = L.isi_vq_num_partials(N)
    part = torch.empty(n_part, dtype=torch.float32, device=z.device)
    out2 = torch.empty(2, dtype=torch.float32, device=z.device)
    _hip.check(L.isi_vq_nearest_flags_f32(z.data_ptr(), codes.data_ptr(), e2.data_ptr(), idx.data_ptr(),
                                          q.data_ptr(), counts.data_ptr(), part.data_ptr(), N, D, K,
                                          8 if split_f16 else 0, _s(z)),
               "isi_vq_nearest_flags_f32")
    _hip.check(L.isi_vq_finalize_f32(part.data_ptr(), n_part, counts.data_ptr(), K, N, D,
                                     out2.data_ptr(), _s(z)), "isi_vq_finalize_f32")
    return q, out2[0], idx, out2[1]


def vq_conv1x1_nearest(x_bhwc: torch.Tensor, packed_w16: torch.Tensor, bias: torch.Tensor, codes: torch.Tensor,
                       e2: torch.Tensor, x2_bhwc: Optional[torch.Tensor] = None, return_all: bool = False):
    """quantize_conv (1x1 on cat(x, x2)) fused with the codebook search (isi_vq_conv1x1_nearest_f32): dense
    channels-last fp32 inputs [B,H,W,C] (converted to the pair format here), `packed_w16` =
    pack_conv_weight(w, with_f16=True).  Returns idx int64 [B,H,W] (or (q, diff, idx, perplexity))."""
    _hip.require_gpu(x_bhwc, "fused quantizer input")
    L = _hip.lib()
    B, H, W, C0 = x_bhwc.shape
    K, D = codes.shape
    srcs = [pair_encode(x_bhwc.contiguous())]
    if x2_bhwc is not None:
        srcs.append(pair_encode(x2_bhwc.contiguous()))
    C1 = srcs[1].shape[-1] if len(srcs) > 1 else 0

    def src(t):
        return _hip.isi_src(t.data_ptr(), t.shape[-1], H * W * t.shape[-1], 1, W * t.shape[-1], t.shape[-1])

    s0 = src(srcs[0])
    s1 = src(srcs[1]) if C1 else None
    n = packed_w16.numel() // 2
    N = B * H * W
    idx = torch.empty(B, H, W, dtype=torch.int64, device=x_bhwc.device)
    q = torch.empty(B, H, W, D, dtype=torch.float32, device=x_bhwc.device)
    counts = torch.zeros(K, dtype=torch.int32, device=x_bhwc.device)
    n_part = L.isi_vq_num_partials(N)
    part = torch.empty(n_part, dtype=torch.float32, device=x_bhwc.device)
    ws = torch.empty(max(4, L.isi_vq_conv1x1_workspace_floats(C0, C1, D)), dtype=torch.float32, device=x_bhwc.device)
    _hip.check(L.isi_vq_conv1x1_nearest_f32(C.byref(s0), C.byref(s1) if s1 is not None else None,
                                            packed_w16.data_ptr() + 4 * n, bias.data_ptr(), codes.data_ptr(),
                                            e2.data_ptr(), idx.data_ptr(), q.data_ptr(), None, counts.data_ptr(),
                                            part.data_ptr(), ws.data_ptr(), B, H, W, D, K, _s(x_bhwc)),
               "isi_vq_conv1x1_nearest_f32")
    if not return_all:
        return idx
    out2 = torch.empty(2, dtype=torch.float32, device=x_bhwc.device)
    _hip.check(L.isi_vq_finalize_f32(part.data_ptr(), n_part, counts.data_ptr(), K, N, D, out2.data_ptr(), _s(x_bhwc)),
               "isi_vq_finalize_f32")
    return q, out2[0], idx, out2[1]


def embed_code(idx: torch.Tensor, codes: torch.Tensor) -> torch.Tensor:
    """idx int64 [...] -> [..., D] rows of the codebook."""
    _hip.require_gpu(idx, "code indices")
    if idx.dtype != torch.int64:
        raise TypeError(f"code indices must be int64, got {idx.dtype}")
    idx = idx.contiguous()
    K, D = codes.shape
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=idx.device)
    if idx.numel() == 0:
        return out
    _hip.check(_hip.lib().isi_embed_code_f32(idx.data_ptr(), codes.data_ptr(), out.data_ptr(),
                                             idx.numel(), D, K, _s(idx)), "isi_embed_code_f32")
    return out


def relu_(x: torch.Tensor) -> torch.Tensor:
    """In-place ReLU on a dense tensor (any memory format that is non-overlapping and dense)."""
    _hip.require_gpu(x, "relu input")
    if not (x.is_contiguous() or x.permute(0, 2, 3, 1).is_contiguous()):
        raise NotImplementedError("in-place ReLU needs dense NCHW or channels-last storage")
    _hip.check(_hip.lib().isi_relu_inplace_f32(x.data_ptr(), x.numel(), _s(x)), "isi_relu_inplace_f32")
    return x
