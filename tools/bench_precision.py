import sys, pathlib, time, torch
ROOT = pathlib.Path('/root/repo')
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import bench as Bn
dev = torch.device('cuda:0')
m, sd = Bn._build_model(dev)
x = torch.randn(64, 2, 128, 512, generator=torch.Generator().manual_seed(100)).to(dev)
res = {}
with torch.no_grad():
    for mode in ('f32', 'bf16x3_decoder', 'bf16x3'):
        m.conv_precision = mode
        for _ in range(3): out = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        res[mode] = (dt, [o.clone() for o in out])
        print(mode, f"{dt*1e3:.3f} ms/step  {64/dt:.0f} spectrograms/s")
ref = res['f32'][1]
for mode in ('bf16x3_decoder', 'bf16x3'):
    o = res[mode][1]
    e = ((o[0]-ref[0]).abs().max()/ref[0].abs().max()).item()
    print(mode, "dec max err/max|ref|", e, "id_t equal", (o[4]==ref[4]).float().mean().item(), "id_b equal", (o[5]==ref[5]).float().mean().item())
