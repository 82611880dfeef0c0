#!/usr/bin/env python3
"""Random-shape check of the LDS-DMA kernels against the register-staged kernels (bit for bit with the accumulator flush
off) and against float64 (with it on): conv_pair_kernel (one / two sources, 3x3 s1, 4x4 s2, 64 / 128 output channels,
fp32 / pair outputs) and resblock_pair_kernel (both row counts).  tools/fuzz_dma_kernels.py [cases] [seed]"""
import os
import pathlib
import random
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402
from interactive_spectrogram_inpainting.vqvae import _ops  # noqa: E402

F = torch.nn.functional


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    cl = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    enc = lambda t: _ops.pair_encode(t.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    worst = 0.0
    for case in range(n_cases):
        B, H, W = rnd.randint(1, 3), rnd.randint(2, 45), rnd.randint(2, 90)
        C0 = rnd.choice([32, 64, 128]); C1 = rnd.choice([0, 0, 32, 64]); cout = rnd.choice([64, 128])
        k, stride = rnd.choice([(3, 1), (4, 2)])
        if k * k * (C0 + C1) < 128 or (stride == 2 and (H < 2 or W < 2)):
            continue
        x0 = torch.relu(torch.randn(B, C0, H, W, generator=g)); x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
        w = torch.randn(cout, C0 + C1, k, k, generator=g) * 0.03; bias = torch.randn(cout, generator=g) * 0.1
        xin = x0 if x1 is None else torch.cat([x0, x1], 1)
        ref64 = torch.relu(F.conv2d(xin.double(), w.double(), bias.double(), stride=stride, padding=1))
        x0d, x1d = cl(x0), (cl(x1) if x1 is not None else None)
        x0p, x1p = enc(x0d), (enc(x1d) if x1d is not None else None)
        pw = _ops.pack_conv_weight(w.to(dev), with_f16=True)
        flags = _ops.PAIR_IN0 | (_ops.PAIR_IN1 if x1 is not None else 0)
        kw = dict(relu=True, bf16x3=4)
        old = _ops.conv2d(x0d, pw, bias.to(dev), cout, k, stride, 1, x2_bchw=x1d, **kw)
        got = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, 1, x2_bchw=x1p, extra_flags=flags, **kw)
        got_p = _ops.pair_decode(_ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, 1, x2_bchw=x1p,
                                             extra_flags=flags | _ops.PAIR_OUT, **kw))
        scale = ref64.abs().max().item() + 1e-30
        e = max((got.cpu().double() - ref64).abs().max().item(), (got_p.cpu().double() - ref64).abs().max().item()) / scale
        worst = max(worst, e)
        assert e < 1.5e-6, (case, B, H, W, C0, C1, cout, k, e)
        with _hip.knob("ISI_CONV_FLUSH", 0):     # (switches are read once; isi_knob_set is the A/B entry point)
            unflushed = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, 1, x2_bchw=x1p, extra_flags=flags, **kw)
        assert torch.equal(unflushed, old), (case, B, H, W, C0, C1, cout, k)
        # residual block on the first source
        if C0 in (64, 128):
            R = 32
            w3 = torch.randn(R, C0, 3, 3, generator=g) * 0.03; b3 = torch.randn(R, generator=g) * 0.1
            w1 = torch.randn(C0, R, 1, 1, generator=g) * 0.1; b1 = torch.randn(C0, generator=g) * 0.1
            r64 = torch.relu(x0.double() + F.conv2d(torch.relu(F.conv2d(x0.double(), w3.double(), b3.double(), padding=1)),
                                                    w1.double(), b1.double()))
            p3, p1 = _ops.pack_conv_weight(w3.to(dev), with_f16=True), _ops.pack_conv_weight(w1.to(dev), with_f16=True)
            for th in ("4", "8"):
                with _hip.knob("ISI_RESPAIR_TH", int(th)):
                    o = _ops.pair_decode(_ops.resblock(x0p, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=4,
                                                       extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT))
                e = (o.cpu().double() - r64).abs().max().item() / (r64.abs().max().item() + 1e-30)
                worst = max(worst, e)
                assert e < 2e-6, ("resblock", case, th, B, H, W, C0, e)
    print(f"{n_cases} cases ok, worst error / max {worst:.2e}")


if __name__ == "__main__":
    main()
