#!/usr/bin/env python3
"""cProfile of the HOST side of the top prior's training steps (the step is within ~10 % of being bound by it)."""
import cProfile
import pathlib
import pstats
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from bench_prior import build  # noqa: E402
from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam  # noqa: E402

dev = torch.device("cuda:0")
m = build(dev).train()
for mod in m.modules():
    if hasattr(mod, "dropout") and isinstance(mod.dropout, float):
        mod.dropout = 0.1
B = 8
code = torch.randint(0, 512, (B, 32, 32), device=dev)
mask = torch.rand(B, 32, 32, device=dev) < 0.5
cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
opt = make_adam(m.parameters(), lr=3e-4)
crit = LabelSmoothingLoss(512, 0.1, dim=1)


def step():
    opt.zero_grad(set_to_none=True)
    src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
    logits, _ = m(tgt, condition=src)
    loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
