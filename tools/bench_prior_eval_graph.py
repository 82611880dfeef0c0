import pathlib, sys, time
ROOT = pathlib.Path("/root/repo") if pathlib.Path("/root/repo/tools").exists() else pathlib.Path.cwd()
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from bench_prior import build
dev = torch.device("cuda:0")
m = build(dev).eval()
B = 8
code = torch.randint(0, 512, (B, 32, 32), device=dev)
mask = torch.rand(B, 32, 32, device=dev) < 0.5
cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
with torch.no_grad():
    src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
    for _ in range(3):
        out = m(tgt, condition=src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = m(tgt, condition=src)
    th = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 20
    print(f"eager eval forward: {te*1e3:.2f} ms (host enqueue {th*1e3:.2f} ms)")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m(tgt, condition=src)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_g = m(tgt, condition=src)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 20
    print(f"graph eval forward: {tg*1e3:.2f} ms; max diff {float((out_g[0]-out[0]).abs().max()):.2e}")
