"""Helpers shared by CPU and GPU tests (test infrastructure)."""
import numpy as np
import torch


def trajectory_case(golden_dir, tag):
    """(fixture, initial state dict, constructor kwargs, the two input batches) of tests/golden/train_trajectory.npz
    (generated from the imported reference by oracle/make_golden.py::train_trajectory_fixtures)."""
    z = np.load(golden_dir / "train_trajectory.npz")
    if tag == "small":
        sd = {k[len("small::w::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("small::w::")}
        kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8, embed_dim=16,
                  num_embeddings=64)
        xs = [torch.from_numpy(z[f"small::x{i}"]) for i in range(2)]
        return z, sd, kw, xs
    zw = np.load(golden_dir / "vqvae_default_tiny.npz")
    sd = {k[3:]: torch.from_numpy(zw[k]) for k in zw.files if k.startswith("w::")}
    g = torch.Generator().manual_seed(int(z["full::x_seed"]))
    xs = [torch.randn(8, 2, 128, 512, generator=g) for _ in range(2)]
    for i, xb in enumerate(xs):
        xb[:, 1].tanh_()
        assert np.array_equal(xb.reshape(-1)[:64].numpy(), z[f"full::x{i}_head"]), "the seeded inputs must be the fixture's"
        assert abs(xb.double().sum().item() - float(z[f"full::x{i}_sum"])) < 1e-6 * xb.numel()
    return z, sd, dict(in_channel=2), xs
