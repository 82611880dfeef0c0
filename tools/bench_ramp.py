#!/usr/bin/env python3
"""Per-forward time of the B=64 VQ-VAE forward from a cold process (clock ramp of the part), then the steady
back-to-back rate: the reason bench.py spins the device up before its warm-up steps."""
import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/interactive-spectrogram-inpainting_amd')
import bench as Bn
dev=torch.device('cuda:0')
m,sd=Bn._build_model(dev)
x=torch.randn(64,2,128,512,generator=torch.Generator().manual_seed(100)).to(dev)
ts=[]
with torch.no_grad():
    for i in range(80):
        torch.cuda.synchronize(); t0=time.perf_counter()
        out=m(x)
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print(" ".join(f"{t:.2f}" for t in ts))
# back-to-back groups of 10 without per-iteration sync
with torch.no_grad():
    for g in range(6):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): out=m(x)
        torch.cuda.synchronize(); print(f"group {g}: {(time.perf_counter()-t0)*100:.3f} ms/iter")
