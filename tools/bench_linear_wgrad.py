#!/usr/bin/env python3
"""Weight gradients of the prior's linear layers (dW = dY^T X, db): rows M = 8 x 1025, (N, K) of the top prior."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting.priors import _train as PT  # noqa: E402


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
    M = 8200
    for N, K in ((512, 512), (1536, 512), (1024, 512), (2048, 512), (512, 2048)):
        x = torch.randn(M, K, device=dev)
        dy = torch.randn(M, N, device=dev)
        t = timed(lambda: PT.linear_wgrad(x, dy))
        print(f"M={M} N={N:4d} K={K:4d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF")


if __name__ == "__main__":
    main()
