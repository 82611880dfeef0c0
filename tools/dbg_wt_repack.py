import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
from bench_prior import build
from interactive_spectrogram_inpainting.priors import _ops
from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
dev = torch.device("cuda:0")
m = build(dev).train()
B = 2
code = torch.randint(0, 512, (B, 32, 32), device=dev); mask = torch.rand(B, 32, 32, device=dev) < 0.5
cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
opt = make_adam(m.parameters(), lr=3e-4); crit = LabelSmoothingLoss(512, 0.1, dim=1)
g = _ops._WT_GROUP
orig = g._repack; log = []
def spy(dev_, device):
    stale = [(k[2], k[3], e[2], _hip.version_of(e[0]()) if e[0]() is not None else None) for k, e in g.entries.items()]
    log.append((len(g.entries), sum(1 for s in stale if s[2] != s[3])))
    return orig(dev_, device)
g._repack = spy
for i in range(3):
    log.clear()
    opt.zero_grad(set_to_none=True)
    src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
    logits, _ = m(tgt, condition=src)
    loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
    loss.backward(); opt.step()
    print("step", i, "repacks", len(log), log[:30])
