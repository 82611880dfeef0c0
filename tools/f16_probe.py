"""Accuracy of the split-product modes of one 3x3 128->128 convolution and one residual block against fp64,
over operand magnitudes (f16 range behaviour of ISI_CONV_F16X3)."""
import sys, torch
sys.path.insert(0, '/root/repo/interactive-spectrogram-inpainting_amd')
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, C, H, W = 4, 128, 32, 48
w = torch.randn(C, C, 3, 3, generator=g) * 0.03
b = torch.randn(C, generator=g) * 0.1
pw = _ops.pack_conv_weight(w.to(dev))
for xs in (1e-4, 1e-2, 1.0, 100.0, 4000.0, 2e4):
    x = (torch.randn(B, C, H, W, generator=g).abs() * xs)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    line = [f"|x|~{xs:g}"]
    for mode in (0, 1, 2, 3):
        y = _ops.conv2d(xd, pw, b.to(dev), C, 3, 1, 1, relu=False, bf16x3=mode).cpu().double()
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        line.append(f"m{mode} {err:.2e}")
    print("conv ", "  ".join(line), flush=True)
for ws in (1e-3, 1.0, 30.0, 100.0):
    x = torch.randn(B, C, H, W, generator=g).abs()
    w2 = w * (ws / 0.03)
    ref = torch.nn.functional.conv2d(x.double(), w2.double(), None, padding=1)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    pw2 = _ops.pack_conv_weight(w2.to(dev))
    line = [f"|w|~{ws:g}"]
    for mode in (0, 2, 3):
        y = _ops.conv2d(xd, pw2, None, C, 3, 1, 1, relu=False, bf16x3=mode).cpu().double()
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        line.append(f"m{mode} {err:.2e}")
    print("convw", "  ".join(line), flush=True)
# residual block
R = 32
w3 = torch.randn(R, C, 3, 3, generator=g) * 0.03; b3 = torch.randn(R, generator=g) * 0.1
w1 = torch.randn(C, R, 1, 1, generator=g) * 0.1; b1 = torch.randn(C, generator=g) * 0.1
p3, p1 = _ops.pack_conv_weight(w3.to(dev)), _ops.pack_conv_weight(w1.to(dev))
for xs in (1e-3, 1.0, 100.0):
    x = torch.randn(B, C, H, W, generator=g).abs() * xs
    h = torch.relu(torch.nn.functional.conv2d(x.double(), w3.double(), b3.double(), padding=1))
    ref = torch.relu(x.double() + torch.nn.functional.conv2d(h, w1.double(), b1.double()))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    line = [f"|x|~{xs:g}"]
    for mode in (0, 1, 2, 3):
        y = _ops.resblock(xd, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=mode).cpu().double()
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        line.append(f"m{mode} {err:.2e}")
    print("resbl", "  ".join(line), flush=True)
