#!/usr/bin/env python3
"""How many times per training step of the top prior does the batched W^T re-pack launch?  (expected: once)"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from interactive_spectrogram_inpainting.priors import _ops  # noqa: E402
from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam  # noqa: E402

dev = torch.device("cuda:0")
m = bench._top_prior(dev).train()
B = 2
g = torch.Generator().manual_seed(1)
code = torch.randint(0, 512, (B, 32, 32), generator=g).to(dev)
mask = (torch.rand(B, 32, 32, generator=g) < 0.5).to(dev)
cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
opt = make_adam(m.parameters(), lr=3e-4)
crit = LabelSmoothingLoss(512, 0.1, dim=1)
calls = []
orig = _ops._WT_GROUP._repack


def counted(dev_, device):
    calls.append(len(_ops._WT_GROUP.entries))
    return orig(dev_, device)


_ops._WT_GROUP._repack = counted
for step in range(4):
    calls.clear()
    opt.zero_grad(set_to_none=True)
    src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
    logits, _ = m(tgt, condition=src)
    loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {step}: {len(calls)} re-pack launches (entries at each: {calls[:6]}{' ...' if len(calls) > 6 else ''}), "
          f"{len(_ops._WT_GROUP.entries)} weights in the group")
