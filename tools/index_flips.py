#!/usr/bin/env python3
"""Code-index agreement on a probe of 24 bench-model spectrograms ([2,128,512], default constructor):
GPU path (per product mode) vs the torch-CPU fp32 oracle (= the reference's arithmetic) and vs the same oracle in
float64 ("exact"), plus oracle-fp32 vs float64 -- how many of the flips are the reference's own rounding."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p_ in (str(ROOT), str(ROOT / "interactive-spectrogram-inpainting_amd")):
    sys.path.insert(0, p_)
import torch  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402
from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda:0")
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 128, 512, generator=g))
    x = torch.randn(n, 2, 128, 512, generator=g)
    with torch.no_grad():
        ref = O.encode(x, sd, cfg)
        sd64 = {k: v.double() for k, v in sd.items()}
        ref64 = O.encode(x.double(), sd64, cfg)
    print(f"{n} spectrograms: {ref[3].numel()} top / {ref[4].numel()} bottom codes")
    print(f"oracle fp32 vs float64          : top {(ref[3] != ref64[3]).sum().item():4d}  bottom {(ref[4] != ref64[4]).sum().item():4d}")
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    for mode in ("split_f16", "f32", "split_bf16"):
        m.conv_precision = mode
        with torch.no_grad():
            out = m.encode(x.to(dev))
        it, ib = out[3].cpu(), out[4].cpu()
        print(f"GPU {mode:10s} vs oracle fp32 : top {(it != ref[3]).sum().item():4d}  bottom {(ib != ref[4]).sum().item():4d}"
              f"   | vs float64: top {(it != ref64[3]).sum().item():4d}  bottom {(ib != ref64[4]).sum().item():4d}")


if __name__ == "__main__":
    main()
