#!/usr/bin/env python3
"""Decode loop, B = 1: direct launches against graphs of W positions (ISI_PRIOR_GRAPH = W), median of 5 codemaps each."""
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import sample as S  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    top = bench._top_prior(dev).eval()
    cls = {"pitch": torch.tensor([[24]]), "instrument_family_str": torch.tensor([[0]])}
    for w in [int(a) for a in sys.argv[1:]] or [0, 1, 8, 32, 64]:
        with _hip.knob("ISI_PRIOR_GRAPH", w):
            ts = []
            for i in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                S.sample_model(top, dev, 1, [32, 32], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.8,
                               generator=torch.Generator().manual_seed(i))
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[1:])
            print(f"ISI_PRIOR_GRAPH={w:3d}: {1024 / ts[2]:8.1f} codes/s  ({ts[2] * 1e3:.1f} ms per codemap, min {ts[0] * 1e3:.1f}, max {ts[-1] * 1e3:.1f})", flush=True)
