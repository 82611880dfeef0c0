"""Pins the CPU oracle (oracle/vqvae_oracle.py) against fixtures produced by
the reference itself (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import vqvae_oracle as O


def _load(golden_dir, name):
    z = np.load(golden_dir / name)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return z, sd


def _cfg(z):
    return O.Config(
        in_channel=int(z["cfg_in_channel"]), num_hidden_channels=int(z["cfg_num_hidden_channels"]),
        n_res_block=int(z["cfg_n_res_block"]), num_residual_channels=int(z["cfg_num_residual_channels"]),
        embed_dim=int(z["cfg_embed_dim"]), num_embeddings=int(z["cfg_num_embeddings"]),
        resolution_factors={"bottom": int(z["cfg_factor_bottom"]), "top": int(z["cfg_factor_top"])},
        groups=int(z["cfg_groups"]) if "cfg_groups" in z.files else 1)


def _close(a, b, tol=1e-5):
    a = torch.as_tensor(a, dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)
    denom = b.abs().max().clamp(min=1e-12)
    err = (a - b).abs().max() / denom
    assert err <= tol, f"max rel-to-max error {err:.3e} > {tol}"


@pytest.mark.parametrize("name", ["vqvae_small.npz", "vqvae_default_tiny.npz", "vqvae_f8_f4.npz", "vqvae_f16_f2.npz", "vqvae_groups2.npz"])
def test_vqvae_forward_matches_reference(golden_dir, name):
    z, sd = _load(golden_dir, name)
    cfg = _cfg(z)
    x = torch.from_numpy(z["x"])
    enc_b = O.encoder(x, sd, "enc_b.", cfg.resolution_factors["bottom"], cfg.n_res_block)
    _close(enc_b, z["enc_b"])
    enc_t = O.encoder(enc_b, sd, "enc_t.", cfg.resolution_factors["top"], cfg.n_res_block)
    _close(enc_t, z["enc_t"])
    q_t, q_b, diff, id_t, id_b, p_t, p_b = O.encode(x, sd, cfg)
    # bit-exact integer outputs
    assert torch.equal(id_t, torch.from_numpy(z["id_t"]))
    assert torch.equal(id_b, torch.from_numpy(z["id_b"]))
    _close(q_t, z["quant_t"]); _close(q_b, z["quant_b"])
    _close(diff, z["diff"]); _close(p_t, z["perplexity_t"]); _close(p_b, z["perplexity_b"])
    dec, diff2, p_t2, p_b2, id_t2, id_b2 = O.forward(x, sd, cfg)
    assert torch.equal(id_t2, id_t) and torch.equal(id_b2, id_b)
    _close(dec, z["dec"])
    _close(O.decode_code(id_t, id_b, sd, cfg), z["dec_code"])


def test_resblock_inplace_semantics(golden_dir):
    z, sd = _load(golden_dir, "resblock.npz")
    x = torch.from_numpy(z["x"])
    y = O.res_block(x, sd, "")
    _close(y, z["y"], 1e-6)
    # the reference mutated its input into relu(x)
    assert np.array_equal(z["x_after"], np.maximum(z["x"], 0))
    # and the result is NOT x + f(relu(x))
    wrong = y - torch.relu(x) + x
    assert (wrong - torch.from_numpy(z["y"])).abs().max() > 1e-3


def test_layers(golden_dir):
    import torch.nn.functional as F
    z = np.load(golden_dir / "layers.npz")
    t = lambda k: torch.from_numpy(z[k])
    for name in ["conv_k4s2", "conv_k4s2_odd"]:
        _close(F.conv2d(t(name + "::x"), t(name + "::weight"), t(name + "::bias"), stride=2, padding=1),
               z[name + "::y"], 1e-6)
    _close(F.conv2d(t("conv_k3::x"), t("conv_k3::weight"), t("conv_k3::bias"), padding=1), z["conv_k3::y"], 1e-6)
    for name in ["convT_k4s2", "convT_k4s2_c2"]:
        _close(F.conv_transpose2d(t(name + "::x"), t(name + "::weight"), t(name + "::bias"), stride=2, padding=1),
               z[name + "::y"], 1e-6)


def test_quantizer(golden_dir):
    z = np.load(golden_dir / "quantizer.npz")
    t = lambda k: torch.from_numpy(z[k])
    for p in ("g", "t"):
        q, diff, ind, perp = O.quantize(t(p + "_z"), t(p + "_embed"))
        assert torch.equal(ind, t(p + "_ind"))
        _close(q, z[p + "_quant"], 1e-6); _close(diff, z[p + "_diff"], 1e-6); _close(perp, z[p + "_perp"], 1e-6)
    assert torch.equal(O.embed_code(t("g_ind"), t("g_embed")), t("g_embed_code"))
    # engineered ties resolve to the lowest index
    assert int(z["t_ind"][0, 0, 0]) == 5 and int(z["t_ind"][0, 1, 2]) == 2
    # train-mode EMA trajectory
    embed = t("e_embed0"); cs = torch.zeros(embed.shape[1]); ea = embed.clone()
    for step in (1, 2):
        zt = t(f"e_z{step}")
        _, diff, ind, perp = O.quantize(zt, embed)
        assert torch.equal(ind, t(f"e_ind{step}"))
        embed, cs, ea = O.ema_update(zt.reshape(-1, zt.shape[-1]), ind, embed, cs, ea)
        _close(embed, z[f"e_embed{step}"], 1e-5); _close(cs, z[f"e_cluster_size{step}"], 1e-6)
        _close(ea, z[f"e_embed_avg{step}"], 1e-6)
    # train mode with index corruption (bottleneck.py:63-73): same generator state -> same offsets
    embed = t("c_embed0")
    torch.manual_seed(16)
    q_st, diff, ind, perp, (e1, cs1, ea1) = O.quantize_train(t("c_z"), embed, torch.zeros(embed.shape[1]), embed.clone(),
                                                             corruption_weights=[0.1, 0.8, 0.1])
    assert torch.equal(ind, t("c_ind"))
    clean = O.quantize(t("c_z"), embed)[2]
    moved = (ind != clean).float().mean().item()
    assert 0.1 < moved < 0.3 and set(((ind - clean) % embed.shape[1]).unique().tolist()) <= {0, 1, embed.shape[1] - 1}
    _close(q_st, z["c_quant"], 1e-6); _close(diff, z["c_diff"], 1e-6); _close(perp, z["c_perp"], 1e-6)
    _close(e1, z["c_embed1"], 1e-5); _close(cs1, z["c_cluster_size1"], 1e-6); _close(ea1, z["c_embed_avg1"], 1e-6)


def test_init_state_dict_keys_match_reference(golden_dir):
    z, sd = _load(golden_dir, "vqvae_default_tiny.npz")
    mine = O.init_state_dict(_cfg(z))
    assert set(mine) == set(sd)
    for k in sd:
        assert mine[k].shape == sd[k].shape, k
    for name in ("vqvae_f8_f4.npz", "vqvae_f16_f2.npz", "vqvae_groups2.npz"):
        z, sd = _load(golden_dir, name)
        mine = O.init_state_dict(_cfg(z))
        assert set(mine) == set(sd)
        for k in sd:
            assert mine[k].shape == sd[k].shape, (name, k)


def test_spectral_loss_oracle_matches_reference(golden_dir):
    """oracle/spectral_loss_oracle.py against values and gradients of the reference's MultiscaleSpectralLoss
    (utils/losses/spectral.py:10-171: DDSP and Jukebox parameter sets, L2Loss)."""
    from oracle import spectral_loss_oracle as S
    z = np.load(golden_dir / "spectral_loss.npz")
    target = torch.from_numpy(z["target"])
    cases = {"ddsp": dict(n_ffts=[64, 128, 256, 512, 1024, 2048], kind="l1"),
             "jukebox": dict(n_ffts=[2048, 1024, 512], window_lengths=[1200, 600, 240], overlap_ratio=0.80, kind="mse",
                             log_loss_alpha=0.0),
             "l2": dict(n_ffts=[256, 512], window_lengths=[200, 512], kind="l2norm", lin_loss_alpha=0.5, log_loss_alpha=2.0)}
    for name, kw in cases.items():
        p = torch.from_numpy(z["pred"]).clone().requires_grad_(True)
        loss = S.multiscale_spectral_loss(p, target, **kw)
        loss.backward()
        _close(loss.detach(), z[f"{name}::loss"], 1e-5)
        _close(p.grad, z[f"{name}::grad"], 1e-4)


def test_unquantized_bottleneck_matches_reference(golden_dir):
    """VQVAE(disable_quantization=True): UnquantizedBottleneck (bottleneck.py:107-119) at both levels."""
    z, sd = _load(golden_dir, "vqvae_unquantized.npz")
    cfg = O.Config(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
                   num_embeddings=64, disable_quantization=True)
    x = torch.from_numpy(z["x"])
    q_t, q_b, diff, id_t, id_b, p_t, p_b = O.encode(x, sd, cfg)
    assert id_t is None and id_b is None
    assert diff.shape == z["diff"].shape == (1, 1) and float(diff) == 0.0
    assert torch.isinf(p_t).all() and torch.isinf(p_b).all() and p_t.shape == z["perplexity_t"].shape
    _close(q_t, z["quant_t"]); _close(q_b, z["quant_b"])
    _close(O.forward(x, sd, cfg)[0], z["dec"])
    _close(O.decode(q_t, q_b, sd, cfg), z["dec_from_quant"])


def reference_training_steps(sd, cfg, batches):
    """The reference's training-step semantics (train_vqvae.py:168-192) on the oracle: yields per-step
    (recon, latent, loss, id_t, id_b) and finally the state."""
    state = {k: v.clone() for k, v in sd.items()}
    learn = [k for k in state if not k.startswith("quantize_t.") and not k.startswith("quantize_b.")]
    for k in learn:
        state[k].requires_grad_(True)
    opt = torch.optim.Adam([state[k] for k in learn], lr=3e-4)
    steps = []
    for x in batches:
        opt.zero_grad()
        dec, diff, id_t, id_b, (nt, nb) = O.forward_train(x, state, cfg)
        recon = torch.nn.functional.mse_loss(dec, x)
        latent = diff.mean()
        loss = recon + 0.25 * latent
        loss.backward()
        opt.step()
        for lvl, new in (("t", nt), ("b", nb)):
            for name, val in zip(("embed", "cluster_size", "embed_avg"), new):
                state[f"quantize_{lvl}.{name}"] = val
        steps.append((recon.item(), latent.item(), loss.item(), id_t, id_b))
    return steps, state


@pytest.mark.parametrize("tag", ["small", "full"])
def test_training_trajectory_matches_reference(golden_dir, tag):
    """BASELINE config 1: two training steps of the imported reference (batch 8; the default constructor at the
    NSynth shape [8,2,128,512] and a reduced configuration): losses, perplexity-defining code indices, EMA codebooks
    and Adam-updated parameters of the oracle's restatement."""
    from tests_support import trajectory_case
    z, sd, kw, xs = trajectory_case(golden_dir, tag)
    cfg = O.Config(**kw)
    steps, state = reference_training_steps(sd, cfg, xs)
    for i, (recon, latent, loss, id_t, id_b) in enumerate(steps):
        assert abs(recon - float(z[f"{tag}::recon{i}"])) <= 1e-4 * abs(float(z[f"{tag}::recon{i}"]))
        assert abs(loss - float(z[f"{tag}::loss{i}"])) <= 1e-4 * abs(float(z[f"{tag}::loss{i}"]))
        assert abs(latent - float(z[f"{tag}::latent{i}"])) <= 2e-3 * abs(float(z[f"{tag}::latent{i}"])) + 1e-7
        for got, key in ((id_t, "id_t"), (id_b, "id_b")):
            ref = torch.from_numpy(z[f"{tag}::{key}{i}"].astype(np.int64))
            agree = (got == ref).float().mean().item()
            # step 0: identical on the machine that made the fixture; another host's torch-CPU convolutions (thread count,
            # instruction set) sum in another order and move a few near-tie codes -- 3 of 4096 on the GPU boxes' EPYC 9575F
            assert agree > 0.998 if i == 0 else agree > 0.995, (tag, i, key, agree)
    for k in z.files:
        if k.startswith(f"{tag}::after::"):
            name = k[len(f"{tag}::after::"):]
            _close(state[name].detach(), z[k], 2e-3 if "quantize_" in name else 2e-4)


def test_quantizer_large_fixture(golden_dir):
    """65 536 near-tie-free Gaussian vectors (regenerated from the seed) against the indices the REFERENCE's
    QuantizedBottleneck.forward produced for them (make_golden.py::quantizer_large_fixtures): the oracle restates it
    bit-exactly, and the fixture is what it claims to be (no vector within 1e-5 of a tie in float64)."""
    from oracle import vqvae_oracle as O
    z = np.load(golden_dir / "quantizer_large.npz")
    embed = torch.from_numpy(z["embed"])
    n = int(z["n"])
    vec = O.near_tie_free_vectors(embed, n, int(z["seed"]), scale=float(z["scale"]), min_gap=float(z["min_gap"]))
    want = torch.from_numpy(z["ind"].astype(np.int64))
    assert n >= 65536 and vec.shape == (n, 64)
    got = torch.cat([O.quantize(vec[lo:lo + 8192], embed)[2] for lo in range(0, n, 8192)])
    assert torch.equal(got, want)
    e = embed.double()
    d = vec.double().pow(2).sum(1, keepdim=True) - 2 * vec.double() @ e + e.pow(2).sum(0)
    best, idx = d.topk(2, dim=1, largest=False)
    assert torch.equal(idx[:, 0], want), "the float64 arg-min agrees: no vector is a rounding coin toss"
    gap = (best[:, 1] - best[:, 0]) / (vec.double().pow(2).sum(1) + e.pow(2).sum(0)[idx[:, 0]])
    assert gap.min().item() > float(z["min_gap"])
