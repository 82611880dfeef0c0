#!/usr/bin/env python3
"""Front-end micro-benchmark: B clips of 4 s NSynth-shape audio (64 000 samples @ 16 kHz) ->
mel log-magnitude / IF spectrograms [B, 2, 1024, 125] and back (n_fft 2048, hop 512)."""
import argparse
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper  # noqa: E402


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
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h = MelSpectrogramsHelper(16000, 2048, 512, 2048).to(dev)
    x = torch.randn(a.batch, 64000, device=dev) * 0.1
    spec = h.to_spectrogram(x)
    t_f = timed(lambda: h.to_spectrogram(x))
    t_i = timed(lambda: h.to_audio(spec))
    B, T, F = a.batch, spec.shape[-1], 1024
    gf_f = 2.0 * B * T * 2048 * 2 * F + 2 * 2.0 * B * T * F * F
    print(f"to_spectrogram B={B}: {t_f:.3f} ms  ({B / t_f * 1e3:.0f} clips/s, {gf_f / t_f / 1e9:.1f} TFLOP/s in the GEMMs)")
    print(f"to_audio       B={B}: {t_i:.3f} ms  ({B / t_i * 1e3:.0f} clips/s)")


if __name__ == "__main__":
    main()
