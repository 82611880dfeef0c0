#!/usr/bin/env python3
"""VERDICT r04 item 1(b), last sentence: Winograd F(2x2, 3x3) on the 3x3 128 -> 128 / 128 -> 32 layers, measured ONCE
against the near-tie certification.  Numerics only (no kernel exists): the CPU oracle's encoder / decoder with every
3x3 stride-1 convolution of >= 64 input channels replaced by the Winograd form

    Y = A^T [ (G g G^T) . (B^T d B) ] A        4 x 4 input tiles, 2 x 2 outputs, 16 element-wise products per tile

in fp32, and once more with both transformed operands rounded to the 22 significand bits the split-f16 pieces carry
(what the three-term products would see).  Reported: codes that move against the direct fp32 oracle, whether each moved
code is a certified near-tie of the reference's own distance formula, and the output error of one layer against fp64.
Runs on the CPU (a few seconds at B = 4)."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def round22(t):
    """Round to 22 significand bits (two 11-bit f16 pieces): drop the last two mantissa bits, to nearest."""
    i = t.contiguous().view(torch.int32)
    return (((i + 2) >> 2) << 2).view(torch.float32)


def winograd3x3(x, w, b, split=False):
    dt = x.dtype
    Bn, Cc, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    Gd, BTd, ATd = G.to(dt), BT.to(dt), AT.to(dt)
    U = Gd @ w.to(dt) @ Gd.t()                                           # [O, C, 4, 4]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                               # [B, C, H/2, W/2, 4, 4]
    V = BTd @ d @ BTd.t()
    if split:
        U, V = round22(U), round22(V)
    M = torch.einsum("ocij,bcnmij->bonmij", U, V)
    Y = ATd @ M @ ATd.t()                                                # [B, O, H/2, W/2, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], H, W)
    return y + b.to(dt).view(1, -1, 1, 1)


class FProxy:
    """torch.nn.functional with conv2d routed through Winograd where it applies."""

    def __init__(self, split):
        self.split, self.count = split, 0

    def __getattr__(self, name):
        return getattr(F, name)

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if (w.shape[2:] == (3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)) and groups == 1 and w.shape[1] >= 64
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
            self.count += 1
            return winograd3x3(x, w, b if b is not None else torch.zeros(w.shape[0]), self.split)
        return F.conv2d(x, w, b, stride, padding, dilation, groups)


def main():
    torch.manual_seed(0)
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 128, 512, generator=g))
    x = torch.randn(4, 2, 128, 512, generator=torch.Generator().manual_seed(100))
    ref = O.forward(x, sd, cfg)
    # one layer against fp64: the bottom encoder's 3x3 128 -> 128 on its real input
    w, b = sd["enc_b.blocks.4.weight"], sd["enc_b.blocks.4.bias"]
    xin = F.relu(F.conv2d(F.relu(F.conv2d(x, sd["enc_b.blocks.0.weight"], sd["enc_b.blocks.0.bias"], stride=2, padding=1)),
                          sd["enc_b.blocks.2.weight"], sd["enc_b.blocks.2.bias"], stride=2, padding=1))
    y64 = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    sc = y64.abs().max()
    print(f"layer 3x3 {w.shape[1]} -> {w.shape[0]}, max error / max |y| against fp64:")
    print(f"  direct fp32 (torch CPU)          {((F.conv2d(xin, w, b, padding=1).double() - y64).abs().max() / sc).item():.2e}")
    print(f"  Winograd fp32                    {((winograd3x3(xin, w, b).double() - y64).abs().max() / sc).item():.2e}")
    print(f"  Winograd, operands to 22 bits    {((winograd3x3(xin, w, b, True).double() - y64).abs().max() / sc).item():.2e}")
    print(f"  transformed input range: max |B^T d B| / max |d| = {((BT @ F.pad(xin, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2) @ BT.t()).abs().max() / xin.abs().max()).item():.2f}")
    for split in (False, True):
        proxy = FProxy(split)
        O.F = proxy
        try:
            out = O.forward(x, sd, cfg)
        finally:
            O.F = F
        moved_t, moved_b = int((out[4] != ref[4]).sum()), int((out[5] != ref[5]).sum())
        chk = O.teacher_forced_code_check(x, sd, cfg, out[4], out[5], eps=1e-6)
        print(f"Winograd {'22-bit operands' if split else 'fp32'}: {proxy.count} layers per forward rerouted; top codes moved "
              f"{moved_t} / {ref[4].numel()}, bottom {moved_b} / {ref[5].numel()} (end to end); teacher-forced: top "
              f"{chk['top_moved']}, bottom {chk['bottom_moved_teacher_forced']}, largest normalised gap "
              f"{chk['largest_normalised_gap']:.2e}, certified near-ties (< 1e-6): {chk['certified_near_ties']}")


if __name__ == "__main__":
    main()
