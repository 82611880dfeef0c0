import sys, time, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/interactive-spectrogram-inpainting_amd")
import torch, bench
import sample as S
from interactive_spectrogram_inpainting.priors import _decode
dev = torch.device("cuda", 0)
m = bench._top_prior(dev).eval()
cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
orig = _decode.NativeSampler.run
def timed(self, *a):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); orig(self, *a); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
_decode.NativeSampler.run = timed
for i in range(2):
    S.sample_model(m, dev, 1, [32, 32], 1.0, generator=torch.Generator().manual_seed(i), class_conditioning=cls, top_p_sampling_p=0.8)
