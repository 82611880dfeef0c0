"""GPU parity of the VQ-VAE training step: parameter gradients of the hand-written
backward and the EMA codebook update against autograd on the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


@pytest.mark.parametrize("cfgk", [
    dict(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8, embed_dim=16, num_embeddings=64),
    dict(in_channel=2, num_hidden_channels=16, n_res_block=1, num_residual_channels=8, embed_dim=8, num_embeddings=32,
         resolution_factors={"bottom": 8, "top": 4}),
    # grouped convolutions (vqvae.py:76) and index corruption in both quantisers (bottleneck.py:63-73)
    dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16, num_embeddings=64,
         groups=2, corruption_weights={"top": [0.1, 0.8, 0.1], "bottom": [0.2, 0.6, 0.2]}),
])
def test_training_step_gradients_and_ema(cfgk):
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(**cfgk)
    sd = O.init_state_dict(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    H, W = (32, 48) if cfg.resolution_factors["bottom"] == 4 else (64, 64)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, H, W, generator=g))
    x = torch.randn(3, 2, H, W, generator=g)
    # ---- oracle: autograd on CPU
    params = {k: v.clone().requires_grad_(not k.startswith("quantize_t.") and not k.startswith("quantize_b."))
              for k, v in sd.items()}
    torch.manual_seed(77)     # the corruption offsets come from the CPU default generator on both sides
    dec, diff, id_t, id_b, (new_t, new_b) = O.forward_train(x, params, cfg)
    loss = torch.nn.functional.mse_loss(dec, x) + 0.25 * diff.mean()
    loss.backward()
    if cfg.corruption_weights["top"] is not None:
        clean = O.forward(x, sd, cfg)
        assert (clean[4] != id_t).any() and (clean[5] != id_b).any()
    # ---- HIP path
    m = VQVAE(**cfgk)
    m.load_state_dict(sd)
    m = m.to(_dev()).train()
    xd = x.to(_dev())
    torch.manual_seed(77)
    out, latent, perp_t, perp_b, it, ib = m(xd)
    assert out.requires_grad and latent.requires_grad
    assert torch.equal(it.cpu(), id_t) and torch.equal(ib.cpu(), id_b)
    assert _rel(out, dec) < 1e-4 and _rel(latent, diff) < 1e-4
    loss_d = torch.nn.functional.mse_loss(out, xd) + 0.25 * latent.mean()
    assert _rel(loss_d, loss) < 1e-4
    loss_d.backward()
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        ref = params[name].grad
        e = _rel(p.grad, ref)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] < 2e-4, f"gradient mismatch: {worst}"
    # EMA-updated buffers (bottleneck.py:79-92)
    for lvl, new in (("t", new_t), ("b", new_b)):
        q = getattr(m, f"quantize_{lvl}")
        assert _rel(q.embed, new[0]) < 1e-5 and _rel(q.cluster_size, new[1]) < 1e-6 and _rel(q.embed_avg, new[2]) < 1e-6
    # the eval path sees the updated codebook
    m.eval()
    sd2 = dict(sd)
    for lvl, new in (("t", new_t), ("b", new_b)):
        sd2[f"quantize_{lvl}.embed"], sd2[f"quantize_{lvl}.cluster_size"], sd2[f"quantize_{lvl}.embed_avg"] = new
    ref_eval = O.forward(x, sd2, cfg)
    got_eval = m(xd)
    assert torch.equal(got_eval[4].cpu(), ref_eval[4]) and _rel(got_eval[0], ref_eval[0]) < 1e-4


def test_two_adam_steps_track_the_oracle():
    """Fixed-seed 2-step loss trajectory (loss = MSE + 0.25 * latent, Adam lr 3e-4)."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfgk = dict(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8, embed_dim=16,
                num_embeddings=64)
    cfg = O.Config(**cfgk)
    sd = O.init_state_dict(cfg, seed=21)
    g = torch.Generator().manual_seed(22)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 32, 32, generator=g))
    xs = [torch.randn(4, 2, 32, 32, generator=g) for _ in range(2)]
    # oracle trajectory
    state = {k: v.clone() for k, v in sd.items()}
    learn = [k for k in state if not k.startswith("quantize_t.") and not k.startswith("quantize_b.")]
    for k in learn:
        state[k].requires_grad_(True)
    opt = torch.optim.Adam([state[k] for k in learn], lr=3e-4)
    ref_losses = []
    for x in xs:
        opt.zero_grad()
        dec, diff, _, _, (nt, nb) = O.forward_train(x, state, cfg)
        loss = torch.nn.functional.mse_loss(dec, x) + 0.25 * diff.mean()
        loss.backward()
        opt.step()
        for lvl, new in (("t", nt), ("b", nb)):
            for name, val in zip(("embed", "cluster_size", "embed_avg"), new):
                state[f"quantize_{lvl}.{name}"] = val
        ref_losses.append(loss.item())
    m = VQVAE(**cfgk)
    m.load_state_dict(sd)
    m = m.to(_dev()).train()
    opt2 = torch.optim.Adam(m.parameters(), lr=3e-4)
    for x, ref in zip(xs, ref_losses):
        m.zero_grad()
        xd = x.to(_dev())
        out, latent, *_ = m(xd)
        loss = torch.nn.functional.mse_loss(out, xd) + 0.25 * latent.mean()
        loss.backward()
        opt2.step()
        assert abs(loss.item() - ref) / abs(ref) < 1e-3, (loss.item(), ref)
