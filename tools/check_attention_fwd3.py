#!/usr/bin/env python3
"""Forward relative attention: the plane-staged kernel (rel_attention_fwd3.hip: K / V / e as 16-bit planes, LDS-DMA,
two wave groups half a step apart) against the exact-fp32 kernel over a sweep of one-channel-per-event shapes / masks,
and its time at the top prior's shape next to the round-4 kernel (ISI_ATTN_NO_FWD3=1)."""
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
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return sorted(ts)[1]


def run(q, k, v, rel, H, Ek, mode, prec, lse=None, logits=None, dense=None):
    _ops.ATTENTION_PRECISION = prec
    return _ops.rel_attention(q, k, v, rel, H, 1, 1, Ek, mask_mode=mode, lse=lse, logits=logits, dense_mask=dense)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--S", type=int, default=1025)
    ap.add_argument("--hd", type=int, default=64)
    ap.add_argument("--modes", type=int, nargs="*", default=[1, 0, 2])
    ap.add_argument("--precs", nargs="*", default=["bf16x3", "bf16", "f16"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    _hip.check(_hip.lib().isi_knob_set(b"ISI_ATTN_FWD3_ALL", 1), "isi_knob_set")   # (default: three-term products only)
    bad = 0
    if not a.no_sweep:
        #          hd  H  Sq    Sk   mode
        shapes = [(64, 2, 33, 33, 1), (64, 2, 33, 33, 2), (64, 2, 200, 200, 1), (32, 3, 260, 260, 1), (64, 2, 77, 150, 0),
                  (64, 1, 1, 1, 0), (64, 1, 1, 97, 0), (32, 2, 64, 64, 2), (64, 2, 129, 129, 1), (32, 3, 257, 257, 0),
                  (64, 3, 385, 300, 0), (64, 3, 257, 257, 1), (32, 2, 300, 300, 1), (64, 8, 1025, 1025, 1),
                  (64, 8, 1025, 1025, 2), (64, 8, 1025, 1025, 0), (32, 16, 1025, 1025, 1), (64, 2, 640, 640, 1),
                  (64, 2, 130, 130, 0), (64, 2, 128, 128, 1), (32, 2, 1030, 1030, 2), (64, 1, 4100, 4100, 1)]
        for hd, H, Sq, Sk, mode in shapes:
            d, B = hd * H, 2
            q, k, v = (torch.randn(s, B, d, device=dev) for s in (Sq, Sk, Sk))
            rel = torch.randn(H, Sq + Sk - 1, hd, device=dev) * 0.5
            for r in (rel, None):
                lse0 = torch.empty(B, H, Sq, device=dev)
                ref = run(q, k, v, r, H, Sk, mode, "f32", lse0)
                for prec, tol in (("bf16x3", 3e-5), ("bf16", 2e-2), ("f16", 2.5e-3)):
                    lse = torch.empty(B, H, Sq, device=dev)
                    got = run(q, k, v, r, H, Sk, mode, prec, lse)
                    with _hip.knob("ISI_ATTN_NO_FWD3", 1):
                        old = run(q, k, v, r, H, Sk, mode, prec)
                    err = ((got - ref).abs().max() / ref.abs().max()).item()
                    err_old = ((old - ref).abs().max() / ref.abs().max()).item()
                    lerr = (lse - lse0).abs().max().item()
                    ok = torch.isfinite(got).all().item() and err < tol and lerr < tol * 30
                    bad += not ok
                    if not ok or prec == "bf16x3":
                        print(f"hd{hd} H{H} {Sq}x{Sk} mode{mode} rel={'y' if r is not None else 'n'} {prec:7s}"
                              f" err {err:.2e} (round-4 kernel {err_old:.2e}) lse {lerr:.2e} {'ok' if ok else 'FAIL'}", flush=True)
            # kept logits and an additive mask tensor, once per shape
            if Sq == Sk and Sq <= 1100:
                ld = (Sk + 31) // 32 * 32
                lg_new = torch.full((B, H, Sq, ld), float("nan"), device=dev)
                lg_old = torch.full((B, H, Sq, ld), float("nan"), device=dev)
                run(q, k, v, rel, H, Sk, mode, "bf16x3", logits=lg_new)
                with _hip.knob("ISI_ATTN_NO_FWD3", 1):
                    run(q, k, v, rel, H, Sk, mode, "bf16x3", logits=lg_old)
                i, j = torch.arange(Sq, device=dev)[:, None], torch.arange(Sk, device=dev)[None, :]
                allowed = (j <= i) if mode == 1 else (j >= i) if mode == 2 else torch.ones(Sq, Sk, dtype=torch.bool, device=dev)
                dl = (lg_new[..., :Sk] - lg_old[..., :Sk])[:, :, allowed].abs().max().item()
                ok = dl < 2e-4
                bad += not ok
                print(f"hd{hd} H{H} {Sq}x{Sk} mode{mode} kept logits vs round-4 kernel: max diff {dl:.2e} {'ok' if ok else 'FAIL'}", flush=True)
                dm = torch.randn(Sq, Sk, device=dev)
                ref = run(q, k, v, rel, H, Sk, 0, "f32", dense=dm)
                got = run(q, k, v, rel, H, Sk, 0, "bf16x3", dense=dm)
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                ok = err < 3e-5
                bad += not ok
                print(f"hd{hd} H{H} {Sq}x{Sk} additive mask tensor: err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        print("sweep failures:", bad)
    if a.no_time:
        return 1 if bad else 0
    B, H, S, hd = a.B, a.H, a.S, a.hd
    d = H * hd
    qkv = torch.randn(S, B, 3 * d, device=dev)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    rel = torch.randn(H, 2 * S - 1, hd, device=dev) * 0.1
    dense = 2.0 * S * S * hd * B * H
    for mode in a.modes:
        ref = run(q, k, v, rel, H, S, mode, "f32")
        for prec in a.precs:
            for old in (0, 1):
                with _hip.knob("ISI_ATTN_NO_FWD3", old):
                    t = timed(lambda: run(q, k, v, rel, H, S, mode, prec))
                    got = run(q, k, v, rel, H, S, mode, prec)
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                print(f"mode {mode} {prec:7s} {'fwd2' if old else 'fwd3'}: {t:7.1f} us  {3 * dense / t / 1e6:7.1f} TF(dense)  err/max {err:.2e}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
