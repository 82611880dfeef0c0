import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "interactive-spectrogram-inpainting_amd"))
from interactive_spectrogram_inpainting.vqvae import _ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
D, K = 64, 512
embed = torch.randn(D, K, generator=g) * 0.7
codes, e2 = _ops.pack_codebook(embed.to(dev))
for n in (1, 33, 256, 20011):
    z = torch.randn(n, D, generator=g) * 0.8
    q0, d0, i0, p0 = _ops.vq_nearest(z.to(dev), codes, e2)
    q1, d1, i1, p1 = _ops.vq_nearest(z.to(dev), codes, e2, split_f16=True)
    bad = (i0 != i1).nonzero().reshape(-1)
    print(n, "mismatches", bad.numel(), "first", [(int(b), int(i0[b]), int(i1[b])) for b in bad[:12]])
    if bad.numel():
        dd = (z.double() ** 2).sum(1, keepdim=True) - 2 * z.double() @ embed.double() + (embed.double() ** 2).sum(0, keepdim=True)
        b = int(bad[0]); order = dd[b].argsort()[:4]
        print("  true order", order.tolist(), dd[b][order].tolist(), "got rank", int((dd[b].argsort() == int(i1[b])).nonzero()[0]))
