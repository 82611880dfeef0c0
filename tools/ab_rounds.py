#!/usr/bin/env python3
"""the three legs whose driver-timed numbers moved the wrong way between rounds 3 and 4 -- decode at
B = 1, the VQ-VAE training step, the /timerange-change request -- measured for several TREES of this repository on ONE box,
each tree in its own process, interleaved and repeated so that box-to-box and warm-up effects cancel:

    git worktree add -f build/r03 185647f && make -C build/r03/interactive-spectrogram-inpainting_amd/csrc -j8     (round 3)
    git worktree add -f build/r04 7165488 && make -C build/r04/interactive-spectrogram-inpainting_amd/csrc -j8     (round 4)
    gpurun -- python tools/ab_rounds.py build/r03 build/r04 .

(`build/` is git-ignored but travels to the GPU box.)  Prints one table; `--host-load N` adds N busy host threads (a contended
host).  Keep N well below the box's thread count: with as many burners as hardware threads (tried: 256 of 256) nothing finishes
inside a 20-minute call."""
import json
import os
import pathlib
import subprocess
import sys

CHILD = r"""
import json, sys, time, pathlib
tree = pathlib.Path(sys.argv[1]).resolve()
sys.path[:0] = [str(tree), str(tree / "interactive-spectrogram-inpainting_amd")]
import torch, bench
dev = torch.device("cuda:0")
import inspect
out = {}
if len(sys.argv) > 2 and sys.argv[2] == "light":      # (under host load: no CPU-oracle leg inside the measurement)
    import sample as S
    top = bench._top_prior(dev).eval()
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    ts = []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.sample_model(top, dev, 1, [32, 32], 1.0, generator=torch.Generator().manual_seed(i), class_conditioning=cls, top_p_sampling_p=0.8)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out["decode_B1_codes_per_s"] = round(1024 / sorted(ts[1:])[1], 1)
    del top
else:
    s = bench._prior_sampling(dev)
    out["decode_B1_codes_per_s"] = s["codes_per_s_B1"]
kw = {"steps": 20} if "steps" in inspect.signature(bench._vqvae_training).parameters else {}
t = bench._vqvae_training(dev, None, 1, **kw)
out["vqvae_train_ms"] = t.get("ms_per_step_eager", t["ms_per_step"])
out["vqvae_train_ms_graph"] = t.get("ms_per_step_hip_graph")
model = bench._build_model(dev)[0]
out["timerange_p50_ms"] = bench._timerange_change(dev, model)["p50_ms"]
print("AB " + json.dumps(out))
"""


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    load = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--host-load=")), 0))
    reps = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--reps=")), 2))
    burners = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(load)]
    rows = {t: [] for t in args}
    try:
        for rep in range(reps):
            for t in args:
                env = dict(os.environ)
                env.pop("ISI_HIP_LIBRARY", None)
                r = subprocess.run([sys.executable, "-c", CHILD, t] + (["light"] if load else []), capture_output=True, text=True,
                                   env=env, timeout=1200)
                line = next((ln for ln in r.stdout.splitlines() if ln.startswith("AB ")), None)
                if line is None:
                    print(f"{t}: failed\n{r.stderr[-1500:]}")
                    continue
                rows[t].append(json.loads(line[3:]))
                print(f"rep {rep} {t}: {rows[t][-1]}", flush=True)
    finally:
        for b in burners:
            b.kill()
    print(f"\nhost load: {load} busy threads; {reps} repetitions per tree, best of each")
    print(f"{'tree':12s} {'decode B=1 codes/s':>20s} {'VQ-VAE step ms (eager)':>24s} {'(graph)':>9s} {'timerange p50 ms':>18s}")
    for t in args:
        if not rows[t]:
            continue
        dec = max(r["decode_B1_codes_per_s"] for r in rows[t])
        tr = min(r["vqvae_train_ms"] for r in rows[t])
        trg = [r["vqvae_train_ms_graph"] for r in rows[t] if r.get("vqvae_train_ms_graph")]
        tm = min(r["timerange_p50_ms"] for r in rows[t])
        print(f"{t:12s} {dec:20.1f} {tr:24.2f} {min(trg) if trg else float('nan'):9.2f} {tm:18.1f}")


if __name__ == "__main__":
    main()
