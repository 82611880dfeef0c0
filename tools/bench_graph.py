#!/usr/bin/env python3
"""VQ-VAE forward at B=64: direct launches against a replayed HIP graph of the same launches (torch.cuda.CUDAGraph
capture of `model(x)`: the library allocates nothing and never synchronises, so the whole forward is capturable)."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = bench._build_model(dev)[0]
    x = torch.randn(64, 2, 128, 512, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            ref = m(x)
        torch.cuda.synchronize()

        def timed(fn, n=50):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        t_direct = min(timed(lambda: m(x)) for _ in range(3))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
            with torch.cuda.graph(g, stream=s):
                out = m(x)
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[4], ref[4]) and torch.equal(out[5], ref[5])
        t_graph = min(timed(g.replay) for _ in range(3))
    print(f"forward B=64: direct {t_direct:.3f} ms, graph replay {t_graph:.3f} ms")


if __name__ == "__main__":
    main()
