#!/usr/bin/env python3
"""A/B of the transposed convolutions of the forward (B = 64): phase-per-launch register-staged kernel
(ISI_NO_CONVT_PAIR_KERNEL=1) against the fused-phase LDS-DMA kernel csrc/convT_pair_f16.hip (tile heights 4 and 8),
pair-format sources, interleaved rounds in one process."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402
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
    cases = [("convT 128->64 @32x128 fp32 out (dec.up0)", 128, 64, 32, 128, False),
             ("convT 128->64 @16x64 pair out (dec_t.up0)", 128, 64, 16, 64, True),
             ("convT 64->64 @16x64 pair out (upsample)", 64, 64, 16, 64, True)]
    for name, cin, cout, H, W, outp in cases:
        x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
        xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
        w = torch.randn(cin, cout, 4, 4, generator=g) * 0.05
        pw = _ops.pack_convT_weight(w.to(dev), with_f16=True)
        run = lambda: _ops.conv_transpose2d_k4s2(xp, pw, None, cout, relu=True, bf16x3=4,
                                                 extra_flags=_ops.PAIR_IN0 | (_ops.PAIR_OUT if outp else 0))
        flops = 2.0 * B * H * W * 4 * cout * 4 * cin
        res = {}
        variants = {"old": ("ISI_NO_CONVT_PAIR_KERNEL", 1), "th4": ("ISI_CONVT_PAIR_TH", 4), "th8": ("ISI_CONVT_PAIR_TH", 8)}
        for rnd in range(3):
            for vname, (knob, val) in variants.items():
                with _hip.knob(knob, val):
                    res.setdefault(vname, []).append(timed(run))
        t = {k_: min(v) for k_, v in res.items()}
        print(f"{name:44s} " + "   ".join(f"{k_} {v:7.1f} us ({flops / v / 1e6 / 833.3:.3f})" for k_, v in t.items()))




def last_layer():
    """the few-channel last layer 64 -> 2 at 64 x 256 (B = 64): exact-fp32 kernel on an fp32 input against the
    pair-pipeline form (LDS-DMA, split-f16 products)"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, cin, cout, H, W = 64, 64, 2, 64, 256
    x = torch.relu(torch.randn(B, H, W, cin, generator=g)).to(dev)
    xd = x.permute(0, 3, 1, 2)
    xp = _ops.pair_encode(x).permute(0, 3, 1, 2)
    pw = _ops.pack_convT_weight((torch.randn(cin, cout, 4, 4, generator=g) * 0.05).to(dev), with_f16=True)
    f32 = lambda: _ops.conv_transpose2d_k4s2(xd, pw, None, cout, relu=False, out_nchw=True)
    pair = lambda: _ops.conv_transpose2d_k4s2(xp, pw, None, cout, relu=False, out_nchw=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    r = {"fp32": [], "pair": []}
    for _ in range(3):
        r["fp32"].append(timed(f32)); r["pair"].append(timed(pair))
    gb = (B * H * W * cin * 4 + B * 4 * H * W * cout * 4) / 1e9
    print("last layer convT 64->2 @64x256:  " + "   ".join(f"{k} {min(v):7.1f} us ({gb / min(v) * 1e6 / 1e3:.2f} TB/s)" for k, v in r.items()))


if __name__ == "__main__":
    main()
    last_layer()
