#!/usr/bin/env python3
"""Interactive-size latency of the VQ-VAE (B = 1, [2,128,512]): eager launches vs a replayed HIP graph, for the
full forward and for decode_code (the path behind /get-audio, /erase and every inpainting request)."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from bench import _build_model  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def main():
    dev = torch.device("cuda:0")
    model, _ = _build_model(dev)
    with torch.no_grad():
        for B in (1, 4):
            x = torch.randn(B, 2, 128, 512, device=dev)
            out = model(x)
            id_t, id_b = out[4], out[5]
            t_f = timed(lambda: model(x))
            t_d = timed(lambda: model.decode_code(id_t, id_b))
            gf, _ = graphed(lambda: model(x))
            gd, _ = graphed(lambda: model.decode_code(id_t, id_b))
            print(f"B={B}: forward eager {t_f:.3f} ms, graph {timed(gf.replay):.3f} ms | "
                  f"decode_code eager {t_d:.3f} ms, graph {timed(gd.replay):.3f} ms")


if __name__ == "__main__":
    main()
