#!/usr/bin/env python3
"""Weight-gradient kernels on the VQ-VAE's layer shapes at B=64 ([2,128,512] spectrograms): time per launch and
the rate in split-product TFLOP/s (3 MFMA terms per product).  `python tools/bench_wgrad.py [B]`."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting.vqvae import _train  # noqa: E402
from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams  # noqa: E402

# (name, Cin, Cout, k, stride, pad, transposed, input H, input W)
LAYERS = [
    ("enc_b k4s2 2->64", 2, 64, 4, 2, 1, False, 128, 512),
    ("enc_b k4s2 64->128", 64, 128, 4, 2, 1, False, 64, 256),
    ("3x3 128->128 @32x128", 128, 128, 3, 1, 1, False, 32, 128),
    ("res 3x3 128->32 @32x128", 128, 32, 3, 1, 1, False, 32, 128),
    ("res 1x1 32->128 @32x128", 32, 128, 1, 1, 0, False, 32, 128),
    ("enc_t k4s2 128->64", 128, 64, 4, 2, 1, False, 32, 128),
    ("3x3 64->128 @16x64", 64, 128, 3, 1, 1, False, 16, 64),
    ("res 3x3 128->32 @16x64", 128, 32, 3, 1, 1, False, 16, 64),
    ("res 1x1 32->128 @16x64", 32, 128, 1, 1, 0, False, 16, 64),
    ("quantize_t 1x1 128->64", 128, 64, 1, 1, 0, False, 16, 64),
    ("quantize_b 1x1 192->64", 192, 64, 1, 1, 0, False, 32, 128),
    ("dec_t convT 128->64", 128, 64, 4, 2, 1, True, 16, 64),
    ("upsample convT 64->64", 64, 64, 4, 2, 1, True, 16, 64),
    ("dec convT 128->64", 128, 64, 4, 2, 1, True, 32, 128),
    ("dec convT 64->2", 64, 2, 4, 2, 1, True, 64, 256),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    total = 0.0
    for name, cin, cout, k, s, p, tr, H, W in LAYERS:
        layer = _ConvParams(cin, cout, k, s, p, transposed=tr).to(dev)
        x = torch.randn(B, H, W, cin, device=dev).permute(0, 3, 1, 2)
        if tr:
            OH, OW = 2 * H, 2 * W
        else:
            OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = torch.randn(B, OH, OW, cout, device=dev)
        for _ in range(2):
            _train.conv_wgrad(layer, x, dy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            _train.conv_wgrad(layer, x, dy)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        macs = B * (H * W if tr else OH * OW) * cin * cout * k * k
        total += us
        print(f"{name:28s} {us:8.1f} us   {2 * macs / us * 1e-6:7.1f} TFLOP/s (x3 terms: {6 * macs / us * 1e-6:6.1f})")
    print(f"sum {total:.0f} us (the step has 29 launches: residual blocks x2 per stack, 4 stacks)")


if __name__ == "__main__":
    main()
