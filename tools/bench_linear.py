#!/usr/bin/env python3
"""Linear layers of the prior as the library runs them (1x1 implicit-GEMM convolution), per product mode:
rows M = B * S = 8200 (ISI_BENCH_ROWS overrides), (N, K) of the top prior's projections.  Prints us and TFLOP/s-equivalent."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402
from interactive_spectrogram_inpainting.priors import _ops  # noqa: E402


def timed(fn, n=20):
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
    import os
    M = int(os.environ.get("ISI_BENCH_ROWS", 8200))      # B * S = 8 * 1025
    for N, K in ((512, 512), (1536, 512), (1024, 512), (2048, 512), (512, 2048)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        pw = _ops.pack_linear_weight(w, range_check="now")
        line = f"M={M} N={N:4d} K={K:4d}:"
        for prec in ("f32", "bf16x6", "bf16x3", "f16x3"):
            t = timed(lambda: _ops.linear(x, pw, b, N, precision=prec))
            line += f"  {prec} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF"
        print(line)
        if len(sys.argv) > 1:      # A/B of a library switch: `bench_linear.py ISI_SOME_KNOB value`
            with _hip.knob(sys.argv[1], int(sys.argv[2])):
                line = f"   with {sys.argv[1]}={sys.argv[2]}:"
                for prec in ("bf16x3", "f16x3"):
                    t = timed(lambda: _ops.linear(x, pw, b, N, precision=prec))
                    y1 = _ops.linear(x, pw, b, N, precision=prec)
                    line += f"  {prec} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF"
            y0 = _ops.linear(x, pw, b, N, precision="f16x3")
            print(line, " max diff", float((y1 - y0).abs().max()))


if __name__ == "__main__":
    main()
