#!/usr/bin/env python3
"""A/B/C of the split-f16 convolution kernels (register-staged against LDS-DMA; the transposed convolutions are forced onto the DMA kernel for the comparison) on the forward's big layers (B = 64): the register-staged
conv_igemm_f32.hip path (ISI_NO_CONV_PAIR_KERNEL=1) against the LDS-DMA kernel conv_pair_f16.hip, pair-format
sources, interleaved rounds in one process."""
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting.vqvae import _ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(0)
    cases = [("k4s2 64->128 @256x64... (enc_b.down1)", 64, 128, 4, 2, 64, 256, False),
             ("3x3 128->128 @32x128 (enc_b.conv3)", 128, 128, 3, 1, 32, 128, False),
             ("k4s2 128->64 @32x128 (enc_t.down0)", 128, 64, 4, 2, 32, 128, False),
             ("3x3 64->128 @16x64 (enc_t.conv3)", 64, 128, 3, 1, 16, 64, False),
             ("3x3 (64+64)->128 @32x128 (dec.conv3) as 128", 128, 128, 3, 1, 32, 128, False),
             ("convT 128->64 @32x128 (dec.up0)", 128, 64, 4, 2, 32, 128, True),
             ("convT 64->64 @16x64 (upsample)", 64, 64, 4, 2, 16, 64, True)]
    for name, cin, cout, k, s, H, W, tr in cases:
        x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
        xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
        if tr:
            w = torch.randn(cin, cout, 4, 4, generator=g) * 0.05
            pw = _ops.pack_convT_weight(w.to(dev), with_f16=True)
            run = lambda: _ops.conv_transpose2d_k4s2(xp, pw, None, cout, relu=True, bf16x3=4,
                                                     extra_flags=_ops.PAIR_IN0 | (0 if old else _ops.PAIR_OUT))
            flops = 2.0 * B * H * W * 4 * cout * 4 * cin
        else:
            w = torch.randn(cout, cin, k, k, generator=g) * 0.05
            pw = _ops.pack_conv_weight(w.to(dev), with_f16=True)
            pad = 1 if k > 1 else 0
            run = lambda: _ops.conv2d(xp, pw, None, cout, k, s, pad, relu=True, bf16x3=4,
                                      extra_flags=_ops.PAIR_IN0 | (0 if old else _ops.PAIR_OUT))
            OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            flops = 2.0 * B * OH * OW * cout * k * k * cin
        res = {}
        from interactive_spectrogram_inpainting import _hip
        old = False
        variants = {"bm256": ("ISI_CONV_PAIR_BM", 256), "bm128": ("ISI_CONV_PAIR_BM", 128)}
        if tr:
            variants = {"convT": ("ISI_CONV_PAIR_BM", 0)}
        for rnd in range(3):
            for vname, (knob, val) in variants.items():
                with _hip.knob(knob, val):
                    res.setdefault(vname, []).append(timed(run))
        t = {k_: min(v) for k_, v in res.items()}
        print(f"{name:42s} " + "   ".join(f"{k_} {v:7.1f} us ({flops / v / 1e6 / 833.3:.3f})" for k_, v in t.items()))


if __name__ == "__main__":
    main()
