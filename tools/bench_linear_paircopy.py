import pathlib, sys
ROOT = pathlib.Path.cwd()
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from bench_linear import timed
from interactive_spectrogram_inpainting.priors import _ops
dev = torch.device("cuda:0")
M = 8200
for N, K in ((512, 512), (1536, 512), (1024, 512), (2048, 512), (512, 2048)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    p0 = _ops.pack_linear_weight(w, range_check="now")
    p1 = _ops.pack_linear_weight(w, range_check="now", with_f16=True)
    t0 = timed(lambda: _ops.linear(x, p0, b, N, precision="f16x3"))
    t1 = timed(lambda: _ops.linear(x, p1, b, N, precision="f16x3"))
    print(f"N={N} K={K}: f16x3 weights converted in the kernel {t0:.1f} us, pair copy {t1:.1f} us")
