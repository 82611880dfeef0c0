#!/usr/bin/env python3
"""codes/s of KV-cached sampling of the top prior at batch B (argv[1], default 32): median of 3 codemaps after a warm-up."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bench
import sample as S
dev = torch.device("cuda", 0)
m = bench._top_prior(dev).eval()
cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ts = []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S.sample_model(m, dev, B, [32, 32], 1.0, generator=torch.Generator().manual_seed(i), class_conditioning=cls, top_p_sampling_p=0.8)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = sorted(ts[1:])[1]
print(f"B={B}: {B * 1024 / t:.0f} codes/s, {t * 1e3:.1f} ms per codemap batch, {t / 1024 * 1e6:.1f} us per position")
