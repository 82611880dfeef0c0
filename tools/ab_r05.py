#!/usr/bin/env python3
"""Round 6 against round 5 on ONE box: the forward, both training steps, the attention op and decoding, each tree in its own
process, interleaved and repeated (box-to-box differences of +-4 % otherwise swamp a round's work).

    git worktree add -f build/r05 cbdf5e0 && make -C build/r05/interactive-spectrogram-inpainting_amd/csrc -j8
    gpurun -- python tools/ab_r05.py build/r05 .
"""
import json, os, pathlib, subprocess, sys

CHILD = r"""
import json, sys, time, pathlib
tree = pathlib.Path(sys.argv[1]).resolve()
sys.path[:0] = [str(tree), str(tree / "interactive-spectrogram-inpainting_amd")]
import torch, bench
dev = torch.device("cuda:0")
out = {}
model, x = bench._build_model(dev)[:2] if isinstance(bench._build_model(dev), tuple) else (bench._build_model(dev), None)
if x is None or not torch.is_tensor(x):
    x = torch.randn(64, 2, 128, 512, generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    for _ in range(5): model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): model(x)
    torch.cuda.synchronize(); out["forward_ms"] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
s = bench._prior_sampling(dev)
for k in ("codes_per_s_B1", "codes_per_s_B8", "codes_per_s_B32", "codes_per_s_B128"):
    out[k] = s.get(k)
t = bench._vqvae_training(dev, None, 1)
out["vqvae_train_ms"] = t["ms_per_step"]
p = bench._prior_training(dev)
out["prior_train_ms"] = p["ms_per_step"]
a = bench._attention(dev)
out["attn_fwd_us"] = a["bf16x3"]["fwd_us"]; out["attn_bwd_us"] = a["bf16x3"]["bwd_us"]
print("AB " + json.dumps(out))
"""

def main():
    trees = sys.argv[1:]
    rows = {t: [] for t in trees}
    for rep in range(2):
        for t in trees:
            env = dict(os.environ)
            env["ISI_HIP_LIBRARY"] = str(pathlib.Path(t).resolve() / "interactive-spectrogram-inpainting_amd" / "lib" / "libisi_hip.so")
            r = subprocess.run([sys.executable, "-c", CHILD, t], capture_output=True, text=True, env=env, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
            if not line:
                print(t, "FAILED", r.stderr[-800:]); continue
            rows[t].append(json.loads(line[0][3:]))
    keys = list(next(iter(rows.values()))[0]) if all(rows.values()) else []
    print(f"{'':22s}" + "".join(f"{k:>18s}" for k in keys))
    for t, rs in rows.items():
        for r in rs:
            print(f"{t:22s}" + "".join(f"{r[k]:18}" for k in keys))

if __name__ == "__main__":
    main()
