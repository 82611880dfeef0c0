#!/usr/bin/env python3
"""Forward relative attention: the 64-key-tile kernels (rel_attention_fwd2.hip) against the exact-fp32 kernel over a
sweep of shapes / masks / channel layouts, and their time at the top prior's shape next to the round-3 kernels
(ISI_ATTN_OLD_FWD=1)."""
import argparse
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
from interactive_spectrogram_inpainting import _hip  # noqa: E402
from interactive_spectrogram_inpainting.priors import _ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(q, k, v, rel, H, Cq, Ck, Ek, mode, prec, lse=None):
    _ops.ATTENTION_PRECISION = prec
    return _ops.rel_attention(q, k, v, rel, H, Cq, Ck, Ek, mask_mode=mode, lse=lse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--S", type=int, default=1025)
    ap.add_argument("--hd", type=int, default=64)
    ap.add_argument("--modes", type=int, nargs="*", default=[1, 0, 2])
    ap.add_argument("--precs", nargs="*", default=["bf16x3", "bf16", "f16"])
    ap.add_argument("--new-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bad = 0
    if not a.no_sweep:
        shapes = [(16, 4, 33, 33, 1, 1, 1), (16, 4, 33, 33, 1, 1, 2), (16, 4, 132, 33, 4, 1, 0), (64, 2, 200, 200, 1, 1, 1),
                  (32, 3, 260, 260, 4, 4, 1), (64, 2, 77, 150, 2, 1, 0), (16, 2, 1, 1, 1, 1, 0), (64, 1, 1, 97, 1, 1, 0),
                  (32, 2, 64, 64, 1, 1, 2), (64, 2, 129, 129, 1, 1, 1), (32, 3, 257, 257, 1, 1, 0), (16, 5, 258, 130, 2, 1, 0),
                  (64, 3, 385, 300, 1, 1, 0), (64, 3, 257, 257, 1, 1, 1), (32, 2, 300, 300, 1, 1, 1), (64, 8, 1025, 1025, 1, 1, 1),
                  (64, 8, 1025, 1025, 1, 1, 2), (64, 8, 1025, 1025, 1, 1, 0), (64, 8, 4100, 1025, 4, 1, 0), (64, 4, 4100, 4100, 4, 4, 1),
                  (64, 2, 700, 700, 3, 3, 1), (32, 2, 513, 640, 1, 2, 0), (16, 3, 1030, 1030, 1, 1, 2), (64, 2, 640, 640, 1, 1, 1)]
        for hd, H, Sq, Sk, Cq, Ck, mode in shapes:
            d, B = hd * H, 2
            Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
            q, k, v = (torch.randn(s, B, d, device=dev) for s in (Sq, Sk, Sk))
            rel = torch.randn(H, Eq + Ek - 1, hd, device=dev) * 0.5
            for r in (rel, None):
                lse0 = torch.empty(B, H, Sq, device=dev)
                ref = run(q, k, v, r, H, Cq, Ck, Ek, mode, "f32", lse0)
                for prec, tol in (("bf16x3", 3e-5), ("bf16", 2e-2), ("f16", 2.5e-3)):
                    lse = torch.empty(B, H, Sq, device=dev)
                    got = run(q, k, v, r, H, Cq, Ck, Ek, mode, prec, lse)
                    err = ((got - ref).abs().max() / ref.abs().max()).item()
                    lerr = (lse - lse0).abs().max().item()
                    ok = torch.isfinite(got).all().item() and err < tol and lerr < tol * 30
                    bad += not ok
                    if not ok or prec == "bf16x3":
                        print(f"hd{hd} H{H} {Sq}x{Sk} Cq{Cq} Ck{Ck} mode{mode} rel={'y' if r is not None else 'n'} {prec:7s}"
                              f" err {err:.2e} lse {lerr:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        print("sweep failures:", bad)
    B, H, S, hd = a.B, a.H, a.S, a.hd
    d = H * hd
    qkv = torch.randn(S, B, 3 * d, device=dev)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    rel = torch.randn(H, 2 * S - 1, hd, device=dev) * 0.1
    dense = 2.0 * S * S * hd * B * H
    for mode in a.modes:
        ref = run(q, k, v, rel, H, 1, 1, S, mode, "f32")
        for prec in a.precs:
            for old in (0, 1):
                if old and (prec == "f16" or a.new_only):
                    continue
                with _hip.knob("ISI_ATTN_OLD_FWD", old):
                    t = timed(lambda: run(q, k, v, rel, H, 1, 1, S, mode, prec))
                    got = run(q, k, v, rel, H, 1, 1, S, mode, prec)
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                print(f"mode {mode} {prec:7s} {'old' if old else 'new'}: {t:7.1f} us  {3 * dense / t / 1e6:7.1f} TF(dense)  err/max {err:.2e}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
