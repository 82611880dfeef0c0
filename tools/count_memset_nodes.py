#!/usr/bin/env python3
"""Replays a recorded training step N times (argv: prior|vqvae N) -- run under `rocprofv3 --kernel-trace` twice with
different N: the difference in `__amd_rocclr_fillBufferAligned` launches / the extra replays = memset nodes per recorded
step (tools/count_memset_nodes.sh).  A hipMemsetAsync node inside a replayed HIP graph is what made ~3 % of the toy prior's
replayed steps read unwritten partial sums on ROCm 7.2."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam  # noqa: E402

which, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
if which == "prior":
    from bench_prior import build
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    m = build(dev).train()
    for mod in m.modules():
        if hasattr(mod, "dropout") and isinstance(mod.dropout, float):
            mod.dropout = 0.1
    B = 8
    code = torch.randint(0, 512, (B, 32, 32), device=dev)
    mask = torch.rand(B, 32, 32, device=dev) < 0.5
    cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
    opt = make_adam(m.parameters(), lr=3e-4, capturable=True)
    crit = LabelSmoothingLoss(512, 0.1, dim=1)

    def step(code_, mask_):
        opt.zero_grad(set_to_none=True)
        src, tgt = m.to_sequences(code_, condition=code_, class_conditioning=cls, mask=mask_)
        logits, _ = m(tgt, condition=src)
        loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code_)
        loss.backward()
        opt.step()
        return loss.detach()
    inputs = (code, mask)
else:
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from interactive_spectrogram_inpainting.utils.losses.mse import mse_loss
    torch.manual_seed(1)
    m = VQVAE(in_channel=2).to(dev).train()
    opt = make_adam(m.parameters(), lr=3e-4, capturable=True)
    x = torch.randn(64, 2, 128, 512, device=dev)

    def step(x_):
        m.zero_grad()
        out, latent, *_ = m(x_)
        loss = mse_loss(out, x_) + 0.25 * latent.mean()
        loss.backward()
        opt.step()
        return loss.detach()
    inputs = (x,)
g = GraphedTrainingStep(step, inputs, warmup=2)
for _ in range(n):
    loss = g(*inputs)
print(which, n, "replays, loss", float(loss))
g.finish()
