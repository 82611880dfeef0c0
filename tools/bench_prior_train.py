#!/usr/bin/env python3
"""Training-step micro-benchmark of the top prior (config 4 of BASELINE.json:
shape [32,32] = 1024 tokens + start symbol, d_model 512, 6 encoder + 8 decoder
layers, 8 heads, fp32): forward + loss + backward + Adam step, B codemaps per step."""
import argparse
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from bench_prior import build  # noqa: E402
from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--plain-adam", action="store_true", help="torch's default (multi-tensor) Adam instead of the fused one")
    ap.add_argument("--graph", action="store_true", help="record the step into a HIP graph and replay it (GraphedTrainingStep)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = build(dev).train()
    for mod in m.modules():
        if hasattr(mod, "dropout") and isinstance(mod.dropout, float):
            mod.dropout = a.dropout
    B = a.batch
    code = torch.randint(0, 512, (B, 32, 32), device=dev)
    mask = torch.rand(B, 32, 32, device=dev) < 0.5
    cls = {"pitch": torch.full((B, 1), 24, device=dev), "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=dev)}
    opt = (torch.optim.Adam(m.parameters(), lr=3e-4) if a.plain_adam
           else make_adam(m.parameters(), lr=3e-4, **({"capturable": True} if a.graph else {})))
    crit = LabelSmoothingLoss(512, 0.1, dim=1)

    def step():
        opt.zero_grad(set_to_none=True)
        src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
        logits, _ = m(tgt, condition=src)
        loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
        loss.backward()
        opt.step()
        return loss

    if a.graph:
        from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
        eager_step = step
        graphed = GraphedTrainingStep(lambda c, k: eager_step(), (code, mask), warmup=3, index_limits={0: 512})

        def step():      # noqa: F811
            return graphed(code, mask)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    t_host = (time.perf_counter() - t0) / a.steps      # host time to ENQUEUE a step (the loop never synchronises)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"host enqueue time {t_host * 1e3:.1f} ms/step (GPU-bound while this stays below the step time)")
    print(f"prior training step B={B} S=1025: {dt * 1e3:.1f} ms/step  {B / dt:.1f} codemaps/s  "
          f"{B * 1024 / dt:.0f} tokens/s  loss {float(loss):.4f}")
    if a.graph:
        graphed.finish()
    # forward only (no_grad, eval) for comparison
    m.eval()
    with torch.no_grad():
        src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
        for _ in range(2):
            m(tgt, condition=src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(tgt, condition=src)
        torch.cuda.synchronize()
    print(f"eval forward B={B}: {(time.perf_counter() - t0) / a.steps * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
