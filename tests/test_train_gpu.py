"""GPU parity of the VQ-VAE training step: parameter gradients of the hand-written
backward and the EMA codebook update against autograd on the CPU oracle."""
import numpy as np
import pytest
import torch


def _free_port() -> int:
    """A TCP port nobody listens on right now (asked from the OS) for a rendezvous."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


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


def test_training_step_gradients_default_constructor():
    """BASELINE config 3's model: the DEFAULT 128-channel constructor (conv_wgrad_split_kernel, the 8-wave six-term
    residual block, 128x128 tiles -- kernels the reduced configurations above never launch) on [2,2,64,128]:
    every parameter gradient and the EMA buffers against autograd on the CPU oracle."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=13)
    g = torch.Generator().manual_seed(14)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 64, 128, generator=g))
    x = torch.randn(2, 2, 64, 128, generator=g)
    params = {k: v.clone().requires_grad_(not k.startswith("quantize_t.") and not k.startswith("quantize_b."))
              for k, v in sd.items()}
    dec, diff, id_t, id_b, (new_t, new_b) = O.forward_train(x, params, cfg)
    loss = torch.nn.functional.mse_loss(dec, x) + 0.25 * diff.mean()
    loss.backward()
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(_dev()).train()
    xd = x.to(_dev())
    out, latent, perp_t, perp_b, it, ib = m(xd)
    # 2 x 512 top and 2 x 2048 bottom codes of a 128-channel model: a near-tie may move (certified at the full size
    # in test_hip_parity); everything below is compared on the oracle's own codes when they agree
    agree_t, agree_b = (it.cpu() == id_t).float().mean().item(), (ib.cpu() == id_b).float().mean().item()
    assert agree_t > 0.995 and agree_b > 0.995, (agree_t, agree_b)
    exact = agree_t == 1.0 and agree_b == 1.0
    assert _rel(out, dec) < (1e-4 if exact else 2e-2) and _rel(latent, diff) < (1e-4 if exact else 1e-2)
    loss_d = torch.nn.functional.mse_loss(out, xd) + 0.25 * latent.mean()
    loss_d.backward()
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        e = _rel(p.grad, params[name].grad)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] < (2e-4 if exact else 5e-2), f"gradient mismatch: {worst} (codes identical: {exact})"
    if exact:
        for lvl, new in (("t", new_t), ("b", new_b)):
            q = getattr(m, f"quantize_{lvl}")
            assert _rel(q.embed, new[0]) < 1e-5 and _rel(q.cluster_size, new[1]) < 1e-6 and _rel(q.embed_avg, new[2]) < 1e-6


class _Batches(torch.utils.data.Dataset):
    def __init__(self, x):
        self.x = x

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return (self.x[i],)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_train_loop_tracks_reference_trajectory(golden_dir, tag):
    """BASELINE config 1 through the product's own loop: `train_vqvae.train` (the counterpart of the reference's
    train_vqvae.py:133-240) on the fixture's two batches of 8 reproduces the losses / perplexities the imported
    REFERENCE produced (tests/golden/train_trajectory.npz: reduced configuration, and the default constructor at the
    NSynth shape [8,2,128,512]), and leaves the codebooks and Adam-updated parameters where the reference's are."""
    import train_vqvae as T
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from tests_support import trajectory_case
    z, sd, kw, xs = trajectory_case(golden_dir, tag)
    m = VQVAE(**kw)
    m.load_state_dict(sd)
    m = m.to(_dev())
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    crit = torch.nn.MSELoss()
    for i, xb in enumerate(xs):
        loader = torch.utils.data.DataLoader(_Batches(xb), batch_size=8, shuffle=False)
        means = T.train(i, loader, m, crit, opt, device=_dev())
        ref = {k: float(z[f"{tag}::{k}{i}"]) for k in ("recon", "latent", "perp_t", "perp_b")}
        assert abs(means["reconstruction_loss"] - ref["recon"]) <= 1e-3 * ref["recon"], (i, means, ref)
        assert abs(means["latent_loss"] - ref["latent"]) <= 2e-2 * ref["latent"] + 1e-6, (i, means, ref)
        assert abs(means["perplexity_t"] - ref["perp_t"]) <= 2e-2 * ref["perp_t"], (i, means, ref)
        assert abs(means["perplexity_b"] - ref["perp_b"]) <= 2e-2 * ref["perp_b"], (i, means, ref)
    got = m.state_dict()
    for k in z.files:
        if k.startswith(f"{tag}::after::"):
            name = k[len(f"{tag}::after::"):]
            assert _rel(got[name], torch.from_numpy(z[k])) < (2e-2 if "quantize_" in name and "conv" not in name else 2e-3), name
    # evaluate(): sample-weighted means of the same quantities in eval mode
    m_eval_loss, m_eval = T.evaluate(torch.utils.data.DataLoader(_Batches(xs[0]), batch_size=4), m, crit, device=_dev())
    assert np.isfinite(m_eval_loss) and m_eval["perplexity_t"] >= 1.0


def test_train_vqvae_epoch_on_64_synthetic_spectrograms():
    """BASELINE config 1 as worded: `train_vqvae.py` plumbing on 64 synthetic NSynth-shape [2,128,512] mel+IF
    spectrograms, batch 8, one epoch of `train` followed by `evaluate` (default model, MSE + 0.25 latent, Adam 3e-4)."""
    import train_vqvae as T
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    torch.manual_seed(1)
    m = VQVAE(in_channel=2).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    data = T.SyntheticSpectrograms(64)
    loader = torch.utils.data.DataLoader(data, batch_size=8, shuffle=False, drop_last=True)
    crit = torch.nn.MSELoss()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    means = T.train(0, loader, m, crit, opt, device=dev, clip_grad_norm=10.0)
    assert set(means) == set(T.RunningMeans.NAMES) and all(np.isfinite(v) for v in means.values())
    assert 0.2 < means["reconstruction_loss"] < 2.0 and means["perplexity_t"] >= 1.0
    moved = sum(float((m.state_dict()[k] - v).abs().max()) > 0 for k, v in before.items())
    assert moved == len(before), "every parameter and codebook buffer is updated by an epoch"
    val_loss, val = T.evaluate(loader, m, crit, device=dev)
    assert np.isfinite(val_loss) and abs(val_loss - (val["reconstruction_loss"] + 0.25 * val["latent_loss"])) < 1e-6
    # eight Adam steps from a random initialisation do not blow the reconstruction error up
    torch.manual_seed(1)
    fresh = VQVAE(in_channel=2).to(dev)
    v0, val0 = T.evaluate(loader, fresh, crit, device=dev)
    assert val["reconstruction_loss"] < 1.25 * val0["reconstruction_loss"], (val, val0)
    # train-mode encode / decode / decode_code are usable (reference: same modules under model.train())
    m.train()
    x = next(iter(loader))[0][:2].to(dev)
    q_t, q_b, diff, id_t, id_b, p_t, p_b = m.encode(x)
    assert q_t.shape == (2, 64, 16, 64) and id_b.shape == (2, 32, 128) and torch.isfinite(diff).all()
    assert m.decode_code(id_t, id_b).shape == (2, 2, 128, 512)


def test_train_vqvae_epoch_replayed_from_hip_graph_follows_the_eager_epoch():
    """`train_vqvae.train(..., hip_graph=True)`: the loop body recorded once (GraphedVQVAEStep) and replayed per batch --
    model, codebooks and optimizer moments are put back in place after the recording's warm-up steps, so two epochs (four
    batches, then four and a ragged fifth) end with the eager epochs' parameters, codebooks and running means."""
    import train_vqvae as T
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    data = T.SyntheticSpectrograms(16, shape=(2, 64, 128))
    crit = torch.nn.MSELoss()
    res = {}
    for graph in (False, True):
        torch.manual_seed(1)
        m = VQVAE(in_channel=2).to(dev)
        opt = make_adam(m.parameters(), lr=1e-3, capturable=True)
        loader = torch.utils.data.DataLoader(data, batch_size=4, shuffle=False, drop_last=True)
        means = T.train(0, loader, m, crit, opt, device=dev, clip_grad_norm=10.0, hip_graph=graph)
        if graph:
            step0 = m._graphed_train_step[1]
        # a SECOND epoch replays the first one's recording (no re-capture per epoch), with a ragged last batch
        # (18 samples, no drop_last: 4 + 4 + 4 + 4 + 2) run eagerly
        data2 = T.SyntheticSpectrograms(18, shape=(2, 64, 128))
        loader2 = torch.utils.data.DataLoader(data2, batch_size=4, shuffle=False, drop_last=False)
        means = T.train(1, loader2, m, crit, opt, device=dev, clip_grad_norm=10.0, hip_graph=graph)
        if graph:
            assert m._graphed_train_step[1] is step0, "the second epoch recorded again"
        res[graph] = (means, {k: v.clone() for k, v in m.state_dict().items()})
    for k in T.RunningMeans.NAMES:
        a, b = res[True][0][k], res[False][0][k]
        assert abs(a - b) <= 1e-4 * max(1e-6, abs(b)), (k, a, b)
    for k, v in res[False][1].items():
        assert _rel(res[True][1][k], v) < 1e-4, k
    # eager code after the replays sees the replayed weights
    m.eval()
    with torch.no_grad():
        out = m(next(iter(loader))[0].to(dev))
    assert torch.isfinite(out[0]).all()


def test_unquantized_vqvae_against_reference(golden_dir):
    """VQVAE(disable_quantization=True) (vqvae.py:152-160 -> UnquantizedBottleneck, bottleneck.py:107-119): the two
    codebook searches are skipped; fixture from the imported reference; training gradients against oracle autograd."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    z = np.load(golden_dir / "vqvae_unquantized.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16, num_embeddings=64)
    m = VQVAE(disable_quantization=True, **kw)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    x = torch.from_numpy(z["x"]).to(_dev())
    q_t, q_b, diff, id_t, id_b, p_t, p_b = m.encode(x)
    assert id_t is None and id_b is None and diff.shape == (1, 1) and float(diff) == 0.0
    assert p_t.shape == (1,) and torch.isinf(p_t).all() and torch.isinf(p_b).all()
    assert _rel(q_t, torch.from_numpy(z["quant_t"])) < 1e-4 and _rel(q_b, torch.from_numpy(z["quant_b"])) < 1e-4
    dec, diff2, _, _, i_t, i_b = m(x)
    assert i_t is None and i_b is None and _rel(dec, torch.from_numpy(z["dec"])) < 1e-4
    assert _rel(m.decode(q_t, q_b), torch.from_numpy(z["dec_from_quant"])) < 1e-4
    with pytest.raises(NotImplementedError):
        m.quantize_t.embed_code(torch.zeros(1, 2, 2, dtype=torch.int64, device=_dev()))
    # training: gradients flow straight through the identity bottlenecks
    cfg = O.Config(disable_quantization=True, **kw)
    xs = torch.from_numpy(z["x"])[..., :32]          # width divisible by the total down-sampling factor
    params = {k: v.clone().requires_grad_(not k.startswith("quantize_t.") and not k.startswith("quantize_b."))
              for k, v in sd.items()}
    ref_dec, *_ = O.forward_train(xs, params, cfg)
    torch.nn.functional.mse_loss(ref_dec, xs).backward()
    m.train()
    out, latent, perp_t, perp_b, it, ib = m(xs.to(_dev()))
    assert it is None and latent.shape == (1, 1) and _rel(out, ref_dec) < 1e-4
    torch.nn.functional.mse_loss(out, xs.to(_dev())).backward()
    for name, p in m.named_parameters():
        assert _rel(p.grad, params[name].grad) < 2e-4, name


def test_bench_two_ranks_dry_run():
    """bench.py's multi-rank path end to end, two ranks sharing this box's GPU with gloo collectives
    (ISI_BENCH_BACKEND=gloo: a functional dry run -- the measured configuration is one rank per GPU over RCCL): the
    forward line aggregates both ranks, the data-parallel training legs run on every rank -- VQ-VAE (bucketed gradient
    all-reduce + EMA-statistics all-reduce) and the top prior (GradBucketReducer, BASELINE configs[3]) -- and leave the
    ranks with identical weights and codebooks."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parents[1]
    env = dict(os.environ, ISI_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--spinup-ms", "10", "--prior-batch", "1", "--prior-steps", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["batch_per_gpu"] == 4
    # the launcher path: every rank brings its process group up before its own first GPU call
    assert line["process_group_before_first_gpu_call"] is True
    dp = line["vqvae_training_dp"]
    assert "error" not in dp, dp
    assert dp["n_gpus"] == 2 and dp["global_batch"] == 8 and dp["ranks_in_sync"] is True
    pdp = line["prior_training_dp"]
    assert "error" not in pdp, pdp
    assert pdp["n_gpus"] == 2 and pdp["global_batch"] == 2 and pdp["ranks_in_sync"] is True and pdp["value"] > 0
    assert pdp["collectives_per_step"].startswith(tuple("123456789"))
    # both data-parallel legs are replayed from HIP graph segments cut at their collectives
    for leg_ in (dp, pdp):
        assert leg_["mode"] == "hip-graph replay", leg_
        assert leg_["graph_segments"] >= 3 and leg_["ms_per_step_hip_graph"] > 0
        assert leg_["host_enqueue_ms_per_step_hip_graph"] is not None and leg_["ms_per_step_eager"] > 0
    assert dp["graph_segments"] == 1 + 2 + 1 + 4 + 1      # 2 EMA messages + their wait, 4 gradient buckets + their wait


_FORCED_COLLECTIVES_SCRIPT = r"""
import os, sys, pathlib
root = pathlib.Path(sys.argv[1])
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
os.environ["ISI_FORCE_COLLECTIVES"] = "1"
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1)
dev = torch.device("cuda:0")
from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
def build():
    torch.manual_seed(1)
    m = VQVAE(in_channel=2).to(dev).train()
    return m, make_adam(m.parameters(), lr=1e-3, capturable=True)
xs = [torch.randn(4, 2, 64, 128, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(4)]
def make_step(m, opt):
    def step(x):
        m.zero_grad()
        out, latent, *_ = m(x)
        loss = torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()
        loss.backward()
        opt.step()
        return loss
    return step
me, oe = build()
se = make_step(me, oe)
le = [float(se(x)) for x in xs]
mg, og = build()
g = GraphedTrainingStep(make_step(mg, og), (xs[0].clone(),), warmup=1)
lg = [float(g(x)) for x in xs[1:]]
g.finish()
assert g.n_segments == 9, g.n_segments
for a, b in zip(lg, le[1:]):
    assert abs(a - b) <= 1e-5 * abs(b), (lg, le)
for (n, pe), pg in zip(me.named_parameters(), mg.parameters()):
    d = float((pe - pg).abs().max())
    assert d <= 1e-5 * max(1e-3, float(pe.abs().max())), (n, d)
assert torch.equal(me.quantize_t.embed, mg.quantize_t.embed) or float((me.quantize_t.embed - mg.quantize_t.embed).abs().max()) < 1e-5
print("SEGMENTS", g.n_segments, "OK")
dist.destroy_process_group()
"""


def test_graphed_vqvae_step_with_real_rccl_collectives_between_segments():
    """The data-parallel replay path on the GPU with RCCL itself: a 1-rank "nccl" group and ISI_FORCE_COLLECTIVES=1 make the
    VQ-VAE step issue its 2 EMA-statistics messages and 4 gradient-bucket all-reduces although nobody else is there.  The
    recording is cut at each of them -- the bucket boundaries are reached from autograd's device thread, inside the
    hand-written backward -- into 9 segments; replays on three batches reproduce the eager losses and parameters."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, "-c", _FORCED_COLLECTIVES_SCRIPT, str(root), str(_free_port())], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "SEGMENTS 9 OK" in out.stdout


def test_graphed_vqvae_training_step_equals_eager():
    """A VQ-VAE training step (train-mode forward with the in-forward EMA codebook update, hand-written backward, fused
    Adam) recorded into a HIP graph and replayed: same losses, parameters and codebooks as eager steps on the same
    batches; eager code after `finish()` sees the replayed weights (version-keyed caches marked stale)."""
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()

    def build():
        torch.manual_seed(1)
        m = VQVAE(in_channel=2).to(dev).train()
        return m, make_adam(m.parameters(), lr=1e-3, capturable=True)

    def make_step(m, opt):
        def step(x):
            m.zero_grad()
            out, latent, *_ = m(x)
            loss = torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()
            loss.backward()
            opt.step()
            return loss
        return step
    xs = [torch.randn(4, 2, 64, 128, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(4)]
    me, oe = build()
    se = make_step(me, oe)
    le = [float(se(x)) for x in xs]
    mg, og = build()
    g = GraphedTrainingStep(make_step(mg, og), (xs[0].clone(),), warmup=1)
    assert g.n_segments == 1
    lg = [float(g(x)) for x in xs[1:]]
    g.finish()
    for a, b in zip(lg, le[1:]):
        assert abs(a - b) <= 1e-5 * abs(b), (lg, le)
    for (n, pe), pg in zip(me.named_parameters(), mg.parameters()):
        assert _rel(pg, pe) < 1e-5, n
    for q in ("quantize_t", "quantize_b"):
        assert _rel(getattr(mg, q).embed, getattr(me, q).embed) < 1e-5
    me.eval(), mg.eval()
    with torch.no_grad():
        oe_, og_ = me(xs[0]), mg(xs[0])
    assert _rel(og_[0], oe_[0]) < 1e-4


@pytest.mark.parametrize("shape", [
    # (Cin, Cout, k, stride, transposed, B, H, W of the layer INPUT, channels of the first source when two)
    (128, 128, 3, 1, False, 3, 4, 64, 0),     # four channel groups x one input-channel slice, 4 units
    (128, 32, 3, 1, False, 2, 6, 32, 0),      # residual 3x3: one workgroup unit covers the whole gradient
    (64, 128, 3, 1, False, 2, 4, 32, 32),     # concatenated input (two sources)
    (64, 64, 3, 1, False, 2, 2, 96, 0),       # two channel groups x two slices; 3 tiles per row
    (64, 128, 4, 2, False, 2, 8, 64, 0),      # k4 s2: two tap groups, de-interleaved halo columns
    (128, 64, 4, 2, False, 2, 4, 128, 0),
    (128, 64, 4, 2, True, 2, 4, 32, 0),       # transposed: the adjoint stride-2 convolution, roles swapped
    (64, 64, 4, 2, True, 3, 2, 32, 0),
])
def test_weight_gradient_halo_kernel(shape):
    """The halo-staged weight-gradient kernel (`conv_wgrad_halo_kernel`: transposing LDS reads, every tap from one
    staged window) against the per-tap kernel it replaces and against fp64 autograd; bias gradients included."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.vqvae import _train
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams
    cin, cout, k, s, tr, B, H, W, c0 = shape
    dev = _dev()
    g = torch.Generator().manual_seed(sum(shape))
    layer = _ConvParams(cin, cout, k, s, 1, transposed=tr)
    x = torch.randn(B, H, W, cin, generator=g).permute(0, 3, 1, 2)
    if tr:
        OH, OW = 2 * H, 2 * W
    else:
        OH, OW = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    dy = torch.randn(B, OH, OW, cout, generator=g)
    w64 = layer.weight.detach().double().requires_grad_(True)
    b64 = layer.bias.detach().double().requires_grad_(True)
    f = torch.nn.functional.conv_transpose2d if tr else torch.nn.functional.conv2d
    y = f(x.double(), w64, b64, stride=s, padding=1)
    (y * dy.permute(0, 3, 1, 2).double()).sum().backward()
    layer = layer.to(dev)
    xd, dyd = x.to(dev), dy.to(dev)
    if c0:
        a = xd[:, :c0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        b = xd[:, c0:].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        args = (a, dyd, b)
    else:
        args = (xd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), dyd)
    dw, db = _train.conv_wgrad(layer, *args)
    with _hip.knob("ISI_NO_WGRAD_HALO", 1):
        dw_old, db_old = _train.conv_wgrad(layer, *args)
    assert _rel(dw, w64.grad) < 2e-5 and _rel(dw_old, w64.grad) < 2e-5, (_rel(dw, w64.grad), _rel(dw_old, w64.grad))
    assert _rel(dw, dw_old) < 1e-5
    assert _rel(db, b64.grad) < 1e-5 and _rel(db_old, b64.grad) < 1e-5
    # the same gradients written by the reduction straight into parameter-shaped storage (the flat gradient buffer)
    ow, ob = torch.full_like(layer.weight, float("nan")), torch.full_like(layer.bias, float("nan"))
    dw2, db2 = _train.conv_wgrad(layer, *args, out=(ow, ob))
    assert dw2 is ow and db2 is ob and torch.equal(ow, dw.contiguous()) and torch.equal(ob, db)


@pytest.mark.parametrize("kind", ["conv3", "conv1_residual", "k4s2", "convT"])
def test_gated_convolution_epilogue(kind):
    """isi_conv2d_gated_f32 / isi_conv_transpose2d_k4s2_gated_f32 (the ReLU backward mask applied by the
    input-gradient convolution itself): the gated launch equals the plain launch zeroed where the gate is <= 0."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 6, 40
    cin, cout = (64, 32) if kind == "conv1_residual" else (32, 64)
    x = torch.randn(B, H, W, cin, generator=g).to(dev).permute(0, 3, 1, 2)
    if kind == "convT":
        w = torch.randn(cin, cout, 4, 4, generator=g).to(dev) * 0.1
        packed = _ops.pack_convT_weight(w)
        run = lambda gate: _ops.conv_transpose2d_k4s2(x, packed, None, cout, relu=False, bf16x3=1, gate_nhwc=gate)
        oshape = (B, 2 * H, 2 * W, cout)
    else:
        k, s, p = {"conv3": (3, 1, 1), "conv1_residual": (1, 1, 0), "k4s2": (4, 2, 1)}[kind]
        w = torch.randn(cout, cin, k, k, generator=g).to(dev) * 0.1
        packed = _ops.pack_conv_weight(w)
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        oshape = (B, OH, OW, cout)
        res = torch.randn(*oshape, generator=g).to(dev).permute(0, 3, 1, 2) if kind == "conv1_residual" else None
        run = lambda gate: _ops.conv2d(x, packed, None, cout, k, s, p, relu=False, residual_bchw=res, bf16x3=1,
                                       gate_nhwc=gate)
    gate = torch.relu(torch.randn(*oshape, generator=g)).to(dev)          # a rectified activation: ~half zeros
    plain, gated = run(None), run(gate)
    want = torch.where(gate.permute(0, 3, 1, 2) > 0, plain, torch.zeros_like(plain))
    assert (gate == 0).float().mean().item() > 0.3
    assert torch.equal(gated, want)
    with pytest.raises(ValueError):
        run(gate[:, :, :-1])                                               # a gate of another layout is refused
    # round 5: the gate as a PAIR-format tensor (ISI_CONV_GATE_PAIR: the training forward's pair tensors serve as masks)
    gate_pair = _ops.pair_encode(gate)
    if kind == "convT":
        gated_p = _ops.conv_transpose2d_k4s2(x, packed, None, cout, relu=False, bf16x3=1, gate_nhwc=gate_pair, extra_flags=512)
    else:
        gated_p = _ops.conv2d(x, packed, None, cout, k, s, p, relu=False, residual_bchw=res, bf16x3=1, gate_nhwc=gate_pair,
                              extra_flags=512)
    assert torch.equal(gated_p, want)


def test_gated_two_channel_convolution_runs_the_first_layer_kernel():
    """The input gradient of the decoder's 2-channel last layer is a Conv2d(2 -> 64, k4 s2 p1): with a gate it now runs
    conv_first_f32.hip (gated epilogue) instead of the scalar-gather form of the generic kernel; same values."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(32)
    B, H, W = 2, 12, 40
    x = torch.randn(B, H, W, 2, generator=g).to(dev).permute(0, 3, 1, 2)          # channels-last 2-channel gradient
    w = torch.randn(64, 2, 4, 4, generator=g).to(dev) * 0.1
    packed = _ops.pack_conv_weight(w)
    gate = torch.relu(torch.randn(B, H // 2, W // 2, 64, generator=g)).to(dev)
    got = _ops.conv2d(x, packed, None, 64, 4, 2, 1, relu=False, bf16x3=1, gate_nhwc=gate)
    with _hip.knob("ISI_NO_CONV_FIRST", 1):
        ref = _ops.conv2d(x, packed, None, 64, 4, 2, 1, relu=False, bf16x3=1, gate_nhwc=gate)
    assert torch.equal(got == 0, ref == 0)
    assert _rel(got, ref) < 2e-6
    assert ((gate == 0).permute(0, 3, 1, 2) <= (got == 0)).all()


def test_fused_optimizer_step_invalidates_packed_weights():
    """torch's fused optimizers update parameters WITHOUT bumping `tensor._version`; the packed-weight caches are
    keyed on the version plus a global optimizer-step count (`_hip.version_of`), so a forward after such a step must
    see the new weights -- in the VQ-VAE (conv weights, split-f16 copies, dgrad transposes) and in a prior's linear."""
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from VQCPCB.transformer.transformer_custom import _LinearParams
    dev = _dev()
    torch.manual_seed(5)
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16, num_embeddings=64)
    m = VQVAE(**kw).to(dev).train()
    x = torch.randn(2, 2, 32, 64, device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, fused=True)   # (a step that keeps the model sane: at 5e-2 the
    for _ in range(2):                                            #  second forward selects the EMA update's 4e5 dead codes)
        m.zero_grad()
        out, latent, *_ = m(x)
        (torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()).backward()
        opt.step()
    m.eval()
    fresh = VQVAE(**kw).to(dev).eval()
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        a, b = m(x)[0], fresh(x)[0]
    assert torch.isfinite(a).all() and torch.equal(a, b), (a - b).abs().max().item()

    lin = _LinearParams(64, 32).to(dev)
    xs = torch.randn(5, 3, 64, device=dev)
    opt = torch.optim.Adam(lin.parameters(), lr=5e-2, fused=True)
    for _ in range(2):
        lin.zero_grad()
        lin.run(xs).square().mean().backward()
        opt.step()
    with torch.no_grad():
        got = lin.run(xs)
        want = torch.nn.functional.linear(xs.double(), lin.weight.double(), lin.bias.double())
    assert _rel(got, want) < 1e-5


@pytest.mark.parametrize("shape", [(128, 32, 3), (32, 128, 1), (64, 128, 3), (6, 10, 3)])
def test_input_gradient_weight_pack(shape):
    """isi_pack_conv_dgrad_weight_f32 == packing the flipped, transposed weight (one launch instead of three)."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    cout, cin, k = shape
    w = torch.randn(cout, cin, k, k, generator=torch.Generator().manual_seed(k + cout)).to(_dev())
    assert torch.equal(_ops.pack_conv_dgrad_weight(w), _ops.pack_conv_weight(w.flip(2, 3).transpose(0, 1).contiguous()))
    # the packed weight with its split-f16 pair copy in one launch == the two launches it replaces
    n = _ops.pack_conv_weight(w).numel()
    fused = _ops.pack_conv_weight(w, with_f16=True)
    two = _ops._with_f16_copy(torch.cat([_ops.pack_conv_weight(w), torch.empty(n, device=w.device)]), n)
    assert torch.equal(fused[:n], two[:n]) and torch.equal(fused[n:].view(torch.int32), two[n:].view(torch.int32))


@pytest.mark.parametrize("layer_kind", ["conv3 128->128 @32x128", "k4s2 64->128 @64x256", "convT 128->64 @32x128", "res3 128->32 @32x128"])
def test_backward_adjoint_identities_at_full_size(layer_kind):
    """BASELINE config 3's per-GPU shard (B = 64 of [2,128,512]) is too large for the CPU oracle's backward, so the
    backward kernels are checked there through a size-independent property: a convolution is linear in its input and
    in its weight, so for random X, W, dY
        <dY, conv(X; W)>  ==  <dX(dY; W), X>  ==  <dW(X, dY), W>                  (bias-free)
    with the forward on the exact fp32 pipe and the inner products in fp64.  The input-gradient (three-term split,
    gated epilogue off) and weight-gradient kernels (halo-staged, two wave groups) at the sizes the bench times."""
    from interactive_spectrogram_inpainting.vqvae import _ops, _train
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams
    dev = _dev()
    cin, cout, k, s, tr, H, W = {"conv3 128->128 @32x128": (128, 128, 3, 1, False, 32, 128),
                                 "k4s2 64->128 @64x256": (64, 128, 4, 2, False, 64, 256),
                                 "convT 128->64 @32x128": (128, 64, 4, 2, True, 32, 128),
                                 "res3 128->32 @32x128": (128, 32, 3, 1, False, 32, 128)}[layer_kind]
    B = 64
    torch.manual_seed(len(layer_kind))
    layer = _ConvParams(cin, cout, k, s, 1, transposed=tr).to(dev)
    x = torch.randn(B, H, W, cin, device=dev).permute(0, 3, 1, 2)
    y = layer.run(x, relu=False)                                    # exact fp32 products, channels-last storage
    y = y - layer.bias.view(1, -1, 1, 1)                             # the bilinear part
    dy = torch.randn(y.shape[0], y.shape[2], y.shape[3], y.shape[1], device=dev)   # NHWC
    lhs = (dy.double() * y.permute(0, 2, 3, 1).double()).sum().item()
    dw, db = _train.conv_wgrad(layer, x, dy)
    via_w = (dw.double() * layer.weight.detach().double()).sum().item()
    dx = _train.conv_dgrad(_train._DgradWeights(), layer, dy.permute(0, 3, 1, 2))
    via_x = (dx.double() * x.double()).sum().item()
    scale = (dy.double().pow(2).sum().sqrt() * y.double().pow(2).sum().sqrt()).item()
    assert abs(via_w - lhs) < 2e-5 * scale and abs(via_x - lhs) < 2e-5 * scale, (lhs, via_w, via_x, scale)
    assert _rel(db, dy.double().sum((0, 1, 2))) < 1e-5


def test_pair_route_training_forward_side_outputs():
    """The training forms of the pair pipeline's producers (isi_conv2d_twin_f32 / isi_conv_transpose2d_k4s2_twin_f32 /
    isi_resblock_tape_f32): the pair-format output decodes to the fp32 twin (within the pair format's 2^-22), the twin is
    BIT-EQUAL to what the same launch writes as a plain fp32 output, and the residual block's hidden activation and output
    equal the two-convolution composition the round-4 training forward ran."""
    from interactive_spectrogram_inpainting.vqvae import _ops, _train
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import RosinalityResBlock, _ConvParams
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    assert _train._pair_mode()

    def cl(t):      # [B,C,H,W] view of dense channels-last storage
        return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    # plain convolutions (3x3, k4 s2, two sources), ragged sizes
    for cin, cout, k, s, c1, B, H, W in ((128, 128, 3, 1, 0, 2, 9, 70), (64, 128, 4, 2, 0, 3, 10, 36), (64, 128, 3, 1, 64, 2, 5, 33),
                                         (128, 64, 4, 2, 0, 2, 8, 64)):
        layer = _ConvParams(cin + c1, cout, k, s, 1).to(dev)
        x = cl(torch.randn(B, cin, H, W, generator=g).to(dev))
        x2 = cl(torch.randn(B, c1, H, W, generator=g).to(dev)) if c1 else None
        a = _train._conv_fwd(layer, _train._Act(x), True, x2=_train._Act(x2) if c1 else None)
        assert a.pair is not None, (cin, cout, k)
        ref = _ops.conv2d(_ops.pair_encode(x), layer.packed(), layer.bias, cout, k, s, 1, True,
                          x2_bchw=_ops.pair_encode(x2) if c1 else None, bf16x3=4,
                          extra_flags=_ops.PAIR_IN0 | (_ops.PAIR_IN1 if c1 else 0))
        assert torch.equal(a.f32, ref), (cin, cout, k)
        assert _rel(_ops.pair_decode(a.pair), a.f32) < 1e-6
        assert _rel(a.f32, layer.run(x, relu=True, x2=x2, bf16x3=4)) < 2e-6
    # first layer (NCHW spectrogram -> pairs + twin)
    layer = _ConvParams(2, 64, 4, 2, 1).to(dev)
    x = torch.randn(3, 2, 24, 40, generator=g).to(dev)
    a = _train._conv_fwd(layer, _train._Act(x), True)
    assert a.pair is not None and torch.equal(a.f32, layer.run(x, relu=True, bf16x3=4))
    assert _rel(_ops.pair_decode(a.pair), a.f32) < 1e-6
    # transposed convolution: pair + twin, and pair in -> fp32 out
    layer = _ConvParams(128, 64, 4, 2, 1, transposed=True).to(dev)
    x = cl(torch.randn(2, 128, 7, 33, generator=g).to(dev))
    a = _train._conv_fwd(layer, _train._Act(x), True)
    b = _train._conv_fwd(layer, _train._Act(x), True, keep_pair=False)
    assert a.pair is not None and b.pair is None and torch.equal(a.f32, b.f32)
    assert _rel(_ops.pair_decode(a.pair), a.f32) < 1e-6
    assert _rel(a.f32, layer.run(x, relu=True, bf16x3=4)) < 2e-6
    # fused residual block with the tape outputs
    for Cc, B, H, W in ((128, 2, 32, 64), (64, 3, 33, 40)):
        blk = RosinalityResBlock(Cc, 32).to(dev)
        r = cl(torch.randn(B, Cc, H, W, generator=g).to(dev).relu())
        tape = _train.Tape()
        y = _train._res_block_fwd(blk, _train._Act(r), tape, "t")
        assert y.pair is not None
        h_ref = blk.conv[1].run(r, relu=True, bf16x3=4)
        y_ref = blk.conv[3].run(h_ref, relu=True, residual=r, bf16x3=4)
        assert _rel(tape["t.h"], h_ref) < 2e-6 and _rel(y.f32, y_ref) < 2e-6
        assert ((tape["t.h"] > 0) == (h_ref > 0)).float().mean() > 0.9999
        assert _rel(_ops.pair_decode(y.pair), y.f32) < 1e-6


def test_batched_weight_pack_equals_per_layer_packs():
    """isi_pack_multi (one launch for every weight layout of a training step: forward operands with their split-f16 pair
    copies, input-gradient operands, both codebooks) writes the bits of the per-layer pack calls, stamps the per-layer
    caches, and re-packs after an optimizer step."""
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    from interactive_spectrogram_inpainting.vqvae import _ops, _train
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    torch.manual_seed(4)
    m = VQVAE(in_channel=2).to(dev).train()
    m._dgrad_weights = _train._DgradWeights()
    _train.refresh_packs(m)
    grp = m._pack_group
    layers = [l for l in m.modules() if isinstance(l, _ConvParams)]
    assert len(grp.rows) >= 2 * len(layers)      # forward + input-gradient operand per layer (first layer: forward only), 2 codebooks

    def check_all():
        fresh = _train._DgradWeights()
        for l in layers:
            got = l._packed
            want = (_ops.pack_convT_weight if l.transposed else _ops.pack_conv_weight)(l.dense_weight(), with_f16=True)
            assert torch.equal(got, want), l
            assert l.packed() is got                       # the layer's own version check hits the group's buffer
            if id(l) in m._dgrad_weights.cache:
                assert torch.equal(m._dgrad_weights.cache[id(l)][1], fresh.get(l)), l
                assert m._dgrad_weights.get(l) is m._dgrad_weights.cache[id(l)][1]
        for q in (m.quantize_t, m.quantize_b):
            codes, e2 = _ops.pack_codebook(q.embed)
            assert torch.equal(q._packed[0], codes) and torch.equal(q._packed[1], e2)
    check_all()
    n_dgrad = len(m._dgrad_weights.cache)
    assert n_dgrad == len(layers) - 1
    opt = make_adam(m.parameters(), lr=1e-2)
    x = torch.randn(2, 2, 32, 64, device=dev)
    out, latent, *_ = m(x)
    (torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()).backward()
    opt.step()
    before = m.enc_b.blocks[0]._packed.clone()
    _train.refresh_packs(m)
    assert not torch.equal(before, m.enc_b.blocks[0]._packed)
    check_all()


def test_deferred_weight_gradient_reductions_are_bit_identical(monkeypatch):
    """The split reductions behind the weight-gradient GEMMs, collected and run in one launch per 48 jobs at the end of the
    backward (isi_conv_wgrad_deferred_f32 + isi_reduce_jobs_f32), give the bits of the one-launch-per-layer form: same
    partial sums, same order of additions."""
    from interactive_spectrogram_inpainting.vqvae import _train
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    x = torch.randn(3, 2, 64, 128, generator=torch.Generator().manual_seed(3)).to(dev)
    grads = {}
    for defer in (True, False):
        monkeypatch.setattr(_train, "DEFER_REDUCTIONS", defer)
        torch.manual_seed(5)
        m = VQVAE(in_channel=2).to(dev).train()
        out, latent, *_ = m(x)
        (torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()).backward()
        grads[defer] = [p.grad.clone() for p in m.parameters()]
        assert all(torch.isfinite(g_).all() for g_ in grads[defer])
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)


def test_pair_only_tape_entries_give_the_same_gradients(monkeypatch):
    """Round 5: where every consumer of an activation reads pairs -- the next layer's LDS-DMA forward, the halo-staged
    weight-gradient kernel (ISI_CONV_IN*_PAIR), the ReLU mask of the input-gradient convolution (ISI_CONV_GATE_PAIR) -- the
    training forward writes no fp32 twin.  Same gradients as with every twin kept (a pair holds 22 significand bits: the
    operands differ by at most 2^-22 of their value), and the tape really holds pair-only entries."""
    from interactive_spectrogram_inpainting.vqvae import _train
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    x = torch.randn(2, 2, 128, 256, generator=torch.Generator().manual_seed(3)).to(dev)    # (bottom maps of 32 x 64: the fused residual blocks take the pair route)
    grads, n_pair_only = {}, {}
    for drop in (True, False):
        monkeypatch.setattr(_train, "DROP_TWINS", drop)
        seen = []
        orig = _train.Tape.__setitem__

        def spy(self, k, v, _seen=seen, _orig=orig):
            _seen.append(isinstance(v, _train.PairOnly))
            _orig(self, k, v)
        monkeypatch.setattr(_train.Tape, "__setitem__", spy)
        torch.manual_seed(5)
        m = VQVAE(in_channel=2).to(dev).train()
        out, latent, *_ = m(x)
        (torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()).backward()
        monkeypatch.setattr(_train.Tape, "__setitem__", orig)
        grads[drop] = {n: p.grad.clone() for n, p in m.named_parameters()}
        n_pair_only[drop] = sum(seen)
    assert n_pair_only[False] == 0 and n_pair_only[True] >= 6, n_pair_only
    for n in grads[True]:
        assert _rel(grads[True][n], grads[False][n]) < 2e-5, n


def test_mse_loss_kernels_against_torch():
    """utils/losses/mse.py (reference train_vqvae.py:168-176, `nn.MSELoss()`): value and both gradients against torch's for
    a batch of spectrograms and for sizes that are not multiples of 4 / smaller than a workgroup's share."""
    from interactive_spectrogram_inpainting.utils.losses.mse import MSELoss, mse_loss
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    for shape in ((8, 2, 128, 512), (3, 5, 7), (1,), (2, 1027)):
        a = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
        b = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
        a0, b0 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.mse_loss(a0, b0)
        (ref * 3.0).backward()
        got = MSELoss()(a, b)
        (got * 3.0).backward()
        assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref)), shape
        assert _rel(a.grad, a0.grad) < 1e-6 and _rel(b.grad, b0.grad) < 1e-6, shape
    x = torch.randn(4, 4, device=dev)
    assert torch.equal(mse_loss(x, x), torch.zeros((), device=dev))


def test_fused_training_quantizer_is_bit_identical(monkeypatch):
    """Round 5: the training forward runs quantize_conv_* + the codebook search as the eval path's ONE launch
    (isi_vq_conv1x1_nearest_tape_f32: z written for the backward / EMA sums, q with its pair twin for the decoders) instead of
    a 1x1 convolution, the stand-alone search and a pair-encode pass (train_vqvae.py:168-192 -> vqvae.py:260-273,
    bottleneck.py:53-101).  The fused kernel forms z with the products and the order of the convolution kernel: codes,
    loss, codebooks after the EMA update and every parameter gradient are bit-identical, and the fused launch is really taken."""
    from interactive_spectrogram_inpainting.vqvae import _train
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    # (top maps of 32 x 64: the encoders' residual stacks take the pair route at both levels, so both quantisers fuse)
    x = torch.randn(2, 2, 256, 512, generator=torch.Generator().manual_seed(21)).to(dev)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(_train, "FUSED_QUANTIZER", fused)
        calls = []
        orig = _train.quantize_conv_train

        def spy(*a, _orig=orig, _calls=calls, **k):
            _calls.append(1)
            return _orig(*a, **k)
        monkeypatch.setattr(_train, "quantize_conv_train", spy)
        torch.manual_seed(9)
        m = VQVAE(in_channel=2).to(dev).train()
        out, latent, perp_t, perp_b, id_t, id_b = m(x)
        (torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()).backward()
        monkeypatch.setattr(_train, "quantize_conv_train", orig)
        res[fused] = dict(out=out.detach().clone(), latent=latent.detach().clone(), id_t=id_t.clone(), id_b=id_b.clone(),
                          perp=(float(perp_t), float(perp_b)), n=len(calls),
                          embed=(m.quantize_t.embed.clone(), m.quantize_b.embed.clone()),
                          grads={n: p.grad.clone() for n, p in m.named_parameters()})
    assert res[True]["n"] == 2 and res[False]["n"] == 0, (res[True]["n"], res[False]["n"])
    a, b = res[True], res[False]
    assert torch.equal(a["id_t"], b["id_t"]) and torch.equal(a["id_b"], b["id_b"])
    assert torch.equal(a["out"], b["out"]) and torch.equal(a["latent"], b["latent"]) and a["perp"] == b["perp"]
    assert torch.equal(a["embed"][0], b["embed"][0]) and torch.equal(a["embed"][1], b["embed"][1])
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), n


def test_training_step_pair_route_equals_fp32_route_at_full_size(monkeypatch):
    """BASELINE configs[2]'s per-GPU shard at full size (B = 64 of [2, 128, 512]): the training step on the LDS-DMA pair
    kernels (pair tensors, twins only where needed, pair-format masks and weight-gradient sources) against the round-4 step
    on fp32 activations -- same loss, same reconstruction, parameter gradients within 2e-4 of each tensor's maximum (the
    two routes sum in different orders; a pair holds 22 significand bits)."""
    from interactive_spectrogram_inpainting.vqvae import _train
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    x = torch.randn(64, 2, 128, 512, generator=torch.Generator().manual_seed(11)).to(dev)
    res = {}
    for pair in (True, False):
        monkeypatch.setattr(_train, "PAIR_FORWARD", pair)
        torch.manual_seed(7)
        m = VQVAE(in_channel=2).to(dev).train()
        out, latent, perp_t, perp_b, id_t, id_b = m(x)
        loss = torch.nn.functional.mse_loss(out, x) + 0.25 * latent.mean()
        loss.backward()
        res[pair] = (float(loss), out.detach(), id_t, id_b, {n: p.grad.clone() for n, p in m.named_parameters()})
        del m, out, loss
        torch.cuda.empty_cache()
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]), (a[0], b[0])
    # (a code may differ only at a near-tie of the two routes' 1x1 outputs; the reconstruction follows the codes)
    moved = int((a[2] != b[2]).sum()) + int((a[3] != b[3]).sum())
    assert moved <= 64, moved
    if moved == 0:
        assert _rel(a[1], b[1]) < 1e-5
    for n in b[4]:
        assert _rel(a[4][n], b[4][n]) < (2e-4 if moved == 0 else 2e-3), n
