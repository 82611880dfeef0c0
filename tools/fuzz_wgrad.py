#!/usr/bin/env python3
"""Random layer shapes through the weight-gradient kernels: halo-staged (`conv_wgrad_halo_kernel`) and row-major linear
(`linear_wgrad_kernel`) against the per-tap kernel they replace (ISI_NO_WGRAD_HALO / ISI_NO_GEMM_KERNEL) and fp64.
`python tools/fuzz_wgrad.py [cases] [seed]`"""
import pathlib
import random
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402
from interactive_spectrogram_inpainting.priors import _train as PT  # noqa: E402
from interactive_spectrogram_inpainting.vqvae import _train  # noqa: E402
from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    worst = 0.0
    for case in range(n_cases):
        if rnd.random() < 0.25:      # a linear layer
            M, N, K = rnd.randint(40, 3000), 128 * rnd.randint(1, 6), 128 * rnd.randint(1, 6)
            x, dy = torch.randn(M, K, generator=g), torch.randn(M, N, generator=g)
            dw, db = PT.linear_wgrad(x.to(dev), dy.to(dev))
            with _hip.knob("ISI_NO_GEMM_KERNEL", 1):
                dw0, db0 = PT.linear_wgrad(x.to(dev), dy.to(dev))
            e = max(rel(dw, dy.double().t() @ x.double()), rel(db, dy.double().sum(0)), rel(dw, dw0), rel(db, db0))
            assert e < 2e-5, ("linear", M, N, K, e)
            worst = max(worst, e)
            continue
        k, s = rnd.choice([(3, 1), (4, 2)])
        tr = s == 2 and rnd.random() < 0.4
        cin, cout = rnd.choice([32, 64, 128, 192]), rnd.choice([32, 64, 128])
        B = rnd.randint(1, 3)
        if tr:      # layer input [B, cin, H, W] -> output 2H x 2W; the adjoint convolution's output width is W
            H, W = rnd.randint(1, 6), 32 * rnd.randint(1, 3)
            OH, OW = 2 * H, 2 * W
        else:
            OH, OW = 2 * rnd.randint(1, 5), 32 * rnd.randint(1, 3)
            H, W = (OH, OW) if s == 1 else (2 * OH, 2 * OW)
        c0 = 0
        if not tr and cin >= 64 and rnd.random() < 0.3:
            c0 = 32 * rnd.randint(1, cin // 32 - 1)
        layer = _ConvParams(cin, cout, k, s, 1, transposed=tr)
        x = torch.randn(B, H, W, cin, generator=g).permute(0, 3, 1, 2)
        dy = torch.randn(B, OH, OW, cout, generator=g)
        w64 = layer.weight.detach().double().requires_grad_(True)
        b64 = layer.bias.detach().double().requires_grad_(True)
        f = torch.nn.functional.conv_transpose2d if tr else torch.nn.functional.conv2d
        (f(x.double(), w64, b64, stride=s, padding=1) * dy.permute(0, 3, 1, 2).double()).sum().backward()
        layer = layer.to(dev)
        xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        if c0:
            a = xd[:, :c0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            b = xd[:, c0:].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            args = (a, dy.to(dev), b)
        else:
            args = (xd, dy.to(dev))
        dw, db = _train.conv_wgrad(layer, *args)
        with _hip.knob("ISI_NO_WGRAD_HALO", 1):
            dw0, db0 = _train.conv_wgrad(layer, *args)
        e = max(rel(dw, w64.grad), rel(db, b64.grad), rel(dw, dw0), rel(db, db0))
        assert e < 3e-5, (case, k, s, tr, cin, cout, B, H, W, c0, e)
        worst = max(worst, e)
    print(f"{n_cases} cases ok, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
