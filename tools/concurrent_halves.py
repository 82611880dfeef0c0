#!/usr/bin/env python3
"""Experiment (round 5): does the forward gain from running two half batches CONCURRENTLY on disjoint halves of the chip?
The forward is a chain of launches that are either matrix-bound (LDS-DMA convolutions) or memory-bound (residual blocks,
first layer, fused search): on two streams with complementary CU masks (hipExtStreamCreateWithCUMask) a memory-bound launch
of one half batch can run beside a matrix-bound launch of the other.  ISI_CU_COUNT sizes the persistent kernels' grids for
the masked streams.  Prints ms per 64 spectrograms for: one stream / full chip; two unmasked streams; two masked streams
(several mask patterns)."""
import ctypes as C
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)


def run(models, xs, streams, n=30):
    def once():
        for m, x, s in zip(models, xs, streams):
            with torch.cuda.stream(s):
                m(x)
    with torch.no_grad():
        for _ in range(10):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            once()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    m1 = bench._build_model(dev)[0]
    m2 = bench._build_model(dev)[0]
    x = torch.randn(64, 2, 128, 512, device=dev)
    h1, h2 = x[:32].contiguous(), x[32:].contiguous()
    cur = torch.cuda.current_stream()
    print(f"one stream, B = 64, full chip:              {run([m1], [x], [cur]):.3f} ms per 64")
    print(f"one stream, 2 x B = 32 back to back:        {run([m1, m2], [h1, h2], [cur, cur]):.3f} ms per 64")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    print(f"two plain streams, 2 x B = 32:              {run([m1, m2], [h1, h2], [a, b]):.3f} ms per 64")
    ALL = (1 << 256) - 1
    even = sum(1 << i for i in range(0, 256, 2))
    lo = (1 << 128) - 1
    xcd_lo = sum(((1 << 16) - 1) << (32 * i) for i in range(8))      # 16 CUs of every group of 32
    for name, ma in (("even / odd CUs", even), ("low / high 128", lo), ("16 of every 32", xcd_lo)):
        try:
            sa, sb = masked_stream(ma), masked_stream(ALL ^ ma)
        except AssertionError as e:
            print(name, e)
            continue
        with _hip.knob("ISI_CU_COUNT", 128):
            t = run([m1, m2], [h1, h2], [sa, sb])
        print(f"two masked streams ({name:16s}), grids of 128:  {t:.3f} ms per 64")
        t = run([m1, m2], [h1, h2], [sa, sb])
        print(f"two masked streams ({name:16s}), grids of 256:  {t:.3f} ms per 64")
    # the two streams OUT OF PHASE: stream b starts a fraction of a forward late, so that its matrix-bound launches meet the
    # other half batch's memory-bound ones (steady state of a pipelined loop)
    sa, sb = masked_stream(lo), masked_stream(ALL ^ lo)
    for delay_ms in (0.2, 0.45, 0.7, 0.9, 1.1, 1.4):
        with _hip.knob("ISI_CU_COUNT", 128):
            with torch.cuda.stream(sb):
                torch.cuda._sleep(int(delay_ms * 1e-3 * 2.0e9))
            t = run([m1, m2], [h1, h2], [sa, sb])
        torch.cuda.synchronize()
        print(f"two masked streams (low / high 128), second one {delay_ms:.2f} ms late: {t:.3f} ms per 64")
    pa, pb = torch.cuda.Stream(), torch.cuda.Stream()
    for delay_ms in (0.45, 0.9):
        with torch.cuda.stream(pb):
            torch.cuda._sleep(int(delay_ms * 1e-3 * 2.0e9))
        t = run([m1, m2], [h1, h2], [pa, pb])
        torch.cuda.synchronize()
        print(f"two plain streams, second one {delay_ms:.2f} ms late: {t:.3f} ms per 64")
    # one masked stream alone: what half the chip does with a half batch
    sa = masked_stream(even)
    with _hip.knob("ISI_CU_COUNT", 128):
        print(f"one masked stream (even CUs), B = 32 alone:  {run([m1], [h1], [sa]):.3f} ms per 32")


if __name__ == "__main__":
    main()
