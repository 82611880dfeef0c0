"""GPU parity tests: the HIP path (through the C-ABI) against the reference's
golden vectors and against the CPU oracle on seeded inputs.

Tolerances: code indices bit-exact (integer work); fp32 activations within
1e-4 of the tensor's max magnitude (north_star allows 1e-3 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


class _Hip:
    """lazy `_hip` module (the package is importable only with the tests' sys.path set up by conftest)"""
    def __getattr__(self, name):
        from interactive_spectrogram_inpainting import _hip as real
        return getattr(real, name)


_hip = _Hip()


def _dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _close(a, b, tol=TOL, what=""):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max() / b.abs().max().clamp(min=1e-12)
    assert err <= tol, f"{what}: max error / max|ref| = {err:.3e} > {tol}"


def _cfg_kwargs(z):
    return dict(in_channel=int(z["cfg_in_channel"]), num_hidden_channels=int(z["cfg_num_hidden_channels"]),
                n_res_block=int(z["cfg_n_res_block"]), num_residual_channels=int(z["cfg_num_residual_channels"]),
                embed_dim=int(z["cfg_embed_dim"]), num_embeddings=int(z["cfg_num_embeddings"]),
                resolution_factors={"bottom": int(z["cfg_factor_bottom"]), "top": int(z["cfg_factor_top"])},
                groups=int(z["cfg_groups"]) if "cfg_groups" in z.files else 1)


def _model_from_golden(z):
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    m = VQVAE(**_cfg_kwargs(z))
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")})
    return m.to(_dev()).eval()


def test_native_library_is_loaded():
    from interactive_spectrogram_inpainting import _hip
    assert _hip.lib().isi_version().startswith(b"isi_hip gfx950")


def test_single_layers_against_reference(golden_dir):
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams
    z = np.load(golden_dir / "layers.npz")
    dev = _dev()
    specs = {"conv_k4s2": (4, 2, 1, False), "conv_k4s2_odd": (4, 2, 1, False), "conv_k3": (3, 1, 1, False),
             "conv_k1": (1, 1, 0, False), "convT_k4s2": (4, 2, 1, True), "convT_k4s2_c2": (4, 2, 1, True)}
    for name, (k, s, p, tr) in specs.items():
        w = torch.from_numpy(z[name + "::weight"])
        cin, cout = (w.shape[0], w.shape[1]) if tr else (w.shape[1], w.shape[0])
        layer = _ConvParams(cin, cout, k, stride=s, padding=p, transposed=tr)
        layer.load_state_dict({"weight": w, "bias": torch.from_numpy(z[name + "::bias"])})
        layer = layer.to(dev)
        x = torch.from_numpy(z[name + "::x"]).to(dev)
        for xin in (x, x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)):  # NCHW and channels-last
            _close(layer.run(xin, relu=False), z[name + "::y"], 1e-5, name)
            _close(layer.run(xin, relu=True), np.maximum(z[name + "::y"], 0), 1e-5, name + "+relu")


@pytest.mark.parametrize("cin,cout,B,H,W", [(64, 2, 2, 9, 37), (32, 1, 1, 4, 32), (32, 3, 1, 5, 70), (64, 4, 1, 3, 33),
                                              (32, 4, 2, 8, 31)])
def test_last_layer_transposed_conv_kernel(cin, cout, B, H, W):
    """The few-output-channel ConvTranspose2d(4, 2, 1) kernel (GEMM + col2im gather,
    csrc/convT_small_f32.hip) against the op the oracle uses (encoder_decoder.py:204-207)."""
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import _ConvParams
    g = torch.Generator().manual_seed(cin * 100 + cout)
    w = torch.randn(cin, cout, 4, 4, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, H, W, generator=g)
    want = torch.nn.functional.conv_transpose2d(x, w, bias, stride=2, padding=1)
    layer = _ConvParams(cin, cout, 4, stride=2, padding=1, transposed=True)
    layer.load_state_dict({"weight": w, "bias": bias})
    layer = layer.to(_dev())
    for xin in (x.to(_dev()), x.to(_dev()).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)):
        _close(layer.run(xin, relu=False), want.numpy(), 1e-5, f"convT {cin}->{cout}")
        _close(layer.run(xin, relu=True), want.clamp_min(0).numpy(), 1e-5, f"convT {cin}->{cout}+relu")


@pytest.mark.parametrize("cout,B,H,W", [(2, 2, 9, 37), (1, 1, 4, 32), (2, 1, 5, 70), (2, 3, 16, 64), (2, 1, 1, 1)])
def test_last_layer_transposed_conv_pair_kernel(cout, B, H, W):
    """The last decoder layer inside the pair pipeline (convT_k4s2_small_pair_kernel: pair-format input staged by
    LDS-DMA in one stage, three-term split-f16 products, col2im gather; csrc/convT_small_f32.hip): against torch's
    ConvTranspose2d in float64 on the pair-rounded input, NCHW and channels-last outputs, and against the exact-fp32
    kernel (reference vqvae/encoder_decoder.py:204-207)."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    cin = 64
    g = torch.Generator().manual_seed(3 * H + W + cout)
    w = torch.randn(cin, cout, 4, 4, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    x = torch.relu(torch.randn(B, cin, H, W, generator=g))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xp = _ops.pair_encode(xd)
    pt = _ops.pack_convT_weight(w.to(dev), with_f16=True)
    ref = torch.nn.functional.conv_transpose2d(_ops.pair_decode(xp).cpu().double(), w.double(), bias.double(), stride=2, padding=1)
    for relu in (False, True):
        want = torch.relu(ref) if relu else ref
        for out_nchw in (True, False):
            got = _ops.conv_transpose2d_k4s2(xp, pt, bias.to(dev), cout, relu=relu, out_nchw=out_nchw, bf16x3=4,
                                             extra_flags=_ops.PAIR_IN0)
            assert got.shape == (B, cout, 2 * H, 2 * W)
            err = ((got.cpu().double() - want).abs().max() / want.abs().max()).item()
            assert err < 2e-6, f"relu={relu} nchw={out_nchw}: {err:.2e}"
    exact = _ops.conv_transpose2d_k4s2(xd, pt, bias.to(dev), cout, relu=True, out_nchw=True)
    assert (exact - got).abs().max() <= 2.0 ** -20 * got.abs().max()
    with pytest.raises(_hip.HipLibraryError):       # pair-format OUTPUT is not built for the few-channel layer
        _ops.conv_transpose2d_k4s2(xp, pt, bias.to(dev), cout, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 64, 128, 2), (1, 9, 70, 128, 2), (3, 5, 33, 64, 1), (1, 1, 1, 32, 2)])
def test_decoder_tail_kernels(B, H, W, cin, cout):
    """isi_decoder_tail_f32: ConvTranspose2d(cin -> 64) + ReLU + ConvTranspose2d(64 -> cout) with the 64-channel
    activation never written -- the first kernel projects each pixel onto the second layer's taps (Y'), the second
    gathers (csrc/convT_pair_f16.hip YP mode, csrc/convT_small_f32.hip convT_gather_kernel): against torch's two
    ConvTranspose2d layers in float64 on the pair-rounded input with the intermediate rounded like the pair hand-over
    (reference vqvae/encoder_decoder.py:196-209), and against the two-call path of this library."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(11 * H + W + cin + cout)
    x = torch.relu(torch.randn(B, cin, H, W, generator=g))
    w1 = torch.randn(cin, 64, 4, 4, generator=g) * 0.04
    b1 = torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(64, cout, 4, 4, generator=g) * 0.1
    b2 = torch.randn(cout, generator=g)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xp = _ops.pair_encode(xd)
    p1 = _ops.pack_convT_weight(w1.to(dev), with_f16=True)
    p2 = _ops.pack_convT_weight(w2.to(dev), with_f16=True)
    F = torch.nn.functional
    u = torch.relu(F.conv_transpose2d(_ops.pair_decode(xp).cpu().double(), w1.double(), b1.double(), stride=2, padding=1))
    ref = F.conv_transpose2d(u, w2.double(), b2.double(), stride=2, padding=1)
    got = _ops.decoder_tail(xp, p1, b1.to(dev), p2, b2.to(dev), 64, cout)
    assert got.shape == (B, cout, 4 * H, 4 * W)
    err = ((got.cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 3e-6, err
    with _hip.knob("ISI_NO_TAIL_FUSION", 1):     # (the knob only steers the fused forward; these are the plain layers)
        mid = _ops.conv_transpose2d_k4s2(xp, p1, b1.to(dev), 64, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
        two = _ops.conv_transpose2d_k4s2(mid, p2, b2.to(dev), cout, relu=False, out_nchw=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    assert (two - got).abs().max() <= 2.0 ** -19 * got.abs().max()


def test_resblock_against_reference(golden_dir):
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import RosinalityResBlock
    z = np.load(golden_dir / "resblock.npz")
    blk = RosinalityResBlock(16, 8)
    blk.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")})
    blk = blk.to(_dev())
    x = torch.from_numpy(z["x"]).to(_dev())
    y = blk(x)
    _close(y, z["y"], 1e-5, "resblock")
    assert torch.equal(x.cpu(), torch.from_numpy(z["x_after"])), "input must be rectified in place"


def test_quantizer_against_reference(golden_dir):
    from interactive_spectrogram_inpainting.vqvae.bottleneck import QuantizedBottleneck
    z = np.load(golden_dir / "quantizer.npz")
    # "t": the reference's engineered exact ties (duplicate codes -> first index wins) at K = 16, a codebook
    # smaller than the 32-code MFMA tile
    for p, (d, k) in {"g": (64, 512), "t": (8, 16)}.items():
        q = QuantizedBottleneck(d, k)
        q.embed.copy_(torch.from_numpy(z[p + "_embed"]))
        q = q.to(_dev()).eval()
        quant, diff, ind, perp = q(torch.from_numpy(z[p + "_z"]).to(_dev()))
        assert torch.equal(ind.cpu(), torch.from_numpy(z[p + "_ind"])), "indices must be bit-exact"
        _close(quant, z[p + "_quant"], 1e-6, "quantize")
        _close(diff, z[p + "_diff"], 1e-5, "diff")
        _close(perp, z[p + "_perp"], 1e-5, "perplexity")
    q = QuantizedBottleneck(64, 512)
    q.embed.copy_(torch.from_numpy(z["g_embed"]))
    q = q.to(_dev()).eval()
    got = q.embed_code(torch.from_numpy(z["g_ind"]).to(_dev()))
    assert torch.equal(got.cpu(), torch.from_numpy(z["g_embed_code"]))


def test_quantizer_train_mode_against_reference(golden_dir):
    """Stand-alone train-mode forward: EMA trajectory (bottleneck.py:79-92) and index corruption
    (bottleneck.py:63-73; offsets from the CPU default generator, so the same seed gives the
    reference's corrupted indices bit for bit), straight-through / commitment gradients."""
    from interactive_spectrogram_inpainting.vqvae.bottleneck import QuantizedBottleneck
    z = np.load(golden_dir / "quantizer.npz")
    t = lambda k: torch.from_numpy(z[k])
    q = QuantizedBottleneck(16, 32)
    q.embed.copy_(t("e_embed0")); q.embed_avg.copy_(t("e_embed0"))
    q = q.to(_dev()).train()
    for step in (1, 2):
        quant, diff, ind, perp = q(t(f"e_z{step}").to(_dev()))
        assert torch.equal(ind.cpu(), t(f"e_ind{step}"))
        _close(diff, z[f"e_diff{step}"], 1e-5, "diff"); _close(perp, z[f"e_perp{step}"], 1e-5, "perplexity")
        _close(q.embed, z[f"e_embed{step}"], 1e-5, "embed"); _close(q.cluster_size, z[f"e_cluster_size{step}"], 2e-6, "cs")
        _close(q.embed_avg, z[f"e_embed_avg{step}"], 2e-6, "embed_avg")
    q = QuantizedBottleneck(16, 32, corruption_weights=[0.1, 0.8, 0.1])
    q.embed.copy_(t("c_embed0")); q.embed_avg.copy_(t("c_embed0"))
    q = q.to(_dev()).train()
    zc = t("c_z").to(_dev()).requires_grad_(True)
    torch.manual_seed(16)
    quant, diff, ind, perp = q(zc)
    assert torch.equal(ind.cpu(), t("c_ind")), "corrupted indices must be bit-exact"
    _close(quant, z["c_quant"], 1e-6, "quantize"); _close(diff, z["c_diff"], 1e-5, "diff")
    _close(perp, z["c_perp"], 1e-5, "perplexity")
    _close(q.embed, z["c_embed1"], 1e-5, "embed"); _close(q.cluster_size, z["c_cluster_size1"], 2e-6, "cs")
    _close(q.embed_avg, z["c_embed_avg1"], 2e-6, "embed_avg")
    # gradients: d(sum(w * quant) + 3 diff)/dz = w + 3 * 2 (z - q) / numel
    w = torch.randn_like(zc)
    (torch.sum(w * quant) + 3.0 * diff).backward()
    qv = torch.from_numpy(z["c_embed0"]).t()[t("c_ind")].to(_dev())
    _close(zc.grad, w + 6.0 * (zc.detach() - qv) / zc.numel(), 1e-5, "dz")


def test_quantizer_exact_ties_pick_lowest_index():
    from interactive_spectrogram_inpainting.vqvae.bottleneck import QuantizedBottleneck
    from oracle import vqvae_oracle as O
    torch.manual_seed(5)
    q = QuantizedBottleneck(16, 64)
    q.embed[:, 40] = q.embed[:, 7]      # duplicate codes in different 32-row tiles / half-waves
    q.embed[:, 33] = q.embed[:, 1]
    q.embed[:, 6] = q.embed[:, 2]
    zt = torch.randn(3, 5, 9, 16)
    zt[0, 0, 0] = q.embed[:, 7]; zt[0, 0, 1] = q.embed[:, 33]; zt[0, 0, 2] = q.embed[:, 6]
    ref = O.quantize(zt, q.embed.clone())
    assert ref[2][0, 0, 0] == 7 and ref[2][0, 0, 1] == 1 and ref[2][0, 0, 2] == 2
    q = q.to(_dev()).eval()
    quant, diff, ind, perp = q(zt.to(_dev()))
    assert torch.equal(ind.cpu(), ref[2])


def test_quantizer_split_f16_products():
    """isi_vq_nearest_flags_f32(ISI_CONV_F16X3): same outputs as the exact kernel except at near-ties
    (certified in fp64), ties between duplicate codes still go to the lowest index, ragged N, range violations
    give index -1."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    D, K = 64, 512
    embed = torch.randn(D, K, generator=g) * 0.7
    embed[:, 300] = embed[:, 17]
    embed[:, 45] = embed[:, 9]
    codes, e2 = _ops.pack_codebook(embed.to(dev))
    for n in (1, 33, 20011):
        z = torch.randn(n, D, generator=g) * 0.8
        if n > 2:
            z[1] = embed[:, 17]
            z[2] = embed[:, 45]
        q0, d0, i0, p0 = _ops.vq_nearest(z.to(dev), codes, e2)
        q1, d1, i1, p1 = _ops.vq_nearest(z.to(dev), codes, e2, split_f16=True)
        moved = _certify_index_mismatches(z, embed, i1.cpu(), i0.cpu())
        print(f"[split-f16 search, n = {n}] indices differing from the exact kernel (certified near-ties): {moved}")
        assert moved == 0, moved        # (seeded Gaussian data: the f16-candidate search equals the exact kernel)
        same = (i0 == i1)
        assert torch.equal(q0[same], q1[same])
        if n > 2:
            assert i1[1] == 17 and i1[2] == 9
        assert abs(d1.item() - d0.item()) <= 1e-5 * abs(d0.item()) and abs(p1.item() - p0.item()) <= 1e-3 * p0.item()
    z = torch.randn(40, D, generator=g)
    z[7, 3] = 3e4
    z[9, 60] = float("nan")
    q1, d1, i1, p1 = _ops.vq_nearest(z.to(dev), codes, e2, split_f16=True)
    assert i1[7] == -1 and i1[9] == -1 and (i1 >= 0).sum() == 38 and torch.isnan(q1[7]).all() and torch.isnan(d1)
    q0, d0, i0, p0 = _ops.vq_nearest(z.to(dev), codes, e2)
    assert i0[7] >= 0 and i0[9] == -1 and torch.isnan(d0)


def test_quantizer_large_fixture_bit_exact(golden_dir):
    """65 536 Gaussian vectors x the reference's own 512-code codebook, indices produced by the REFERENCE's
    QuantizedBottleneck.forward (oracle/make_golden.py::quantizer_large_fixtures; stored as int16).  The vectors are
    regenerated from the seed and are free of near-ties by construction (float64 gap between best and second-best code
    > 1e-5 of |z|^2 + |e|^2: oracle.near_tie_free_vectors), so every search kernel must return the reference's
    indices BIT-EXACTLY: the exact-fp32 kernel, the f16-candidate / fp32-decision kernel, and the fused
    quantize_conv + search kernel (fed through an identity 1x1 convolution)."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae import _ops
    z = np.load(golden_dir / "quantizer_large.npz")
    embed = torch.from_numpy(z["embed"])
    n, seed = int(z["n"]), int(z["seed"])
    vec = O.near_tie_free_vectors(embed, n, seed, scale=float(z["scale"]), min_gap=float(z["min_gap"]))
    want = torch.from_numpy(z["ind"].astype(np.int64))
    assert vec.shape == (n, 64) and n >= 65536 and want.shape == (n,)
    assert torch.equal(O.quantize(vec, embed)[2], want), "the oracle restates the reference"
    dev = _dev()
    codes, e2 = _ops.pack_codebook(embed.to(dev))
    for split in (False, True):
        q, d, i, p = _ops.vq_nearest(vec.to(dev), codes, e2, split_f16=split)
        assert torch.equal(i.cpu(), want), f"split_f16={split}: {(i.cpu() != want).sum().item()} indices differ"
        assert torch.equal(q.cpu(), embed.t()[want]) or ((q.cpu() - embed.t()[want]).abs().max() < 1e-6)
    # fused kernel: the vectors as a pair8 activation map [B, H, W, 64] through quantize_conv = identity
    B, H, W = 16, 32, 128
    assert B * H * W == n
    pw = _ops.pack_conv_weight(torch.eye(64).reshape(64, 64, 1, 1).to(dev), with_f16=True)
    x = vec.reshape(B, H, W, 64).to(dev)
    got = _ops.vq_conv1x1_nearest(x, pw, torch.zeros(64, device=dev), codes, e2)
    assert torch.equal(got.reshape(-1).cpu(), want), f"fused: {(got.reshape(-1).cpu() != want).sum().item()} indices differ"


def test_codebook_beyond_f16_range():
    """The split-f16 search stores code vectors as f16 pieces of 1024 e (|e| < 63.98).  Every trained codebook has
    FINITE codes far beyond that (unused codes of the EMA update, bottleneck.py:86-92): they are kept out of the f16
    search and handled exactly by the fp32 decision -- a proof per vector that no far code can win, or an fp32 scan of
    the far codes -- so indices equal the reference's, including a vector whose nearest code IS a far code.  A
    NON-FINITE code must stay loud: index -1 / NaN diff in the stand-alone kernel, and the model falls back to
    split_bf16 with a warning."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae import _ops
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    embed = torch.randn(64, 512, generator=g)
    vec = torch.randn(3000, 64, generator=g)
    vec[5] = 0
    vec[5, 7] = 70.0
    embed[:, 200] = vec[5]                      # the true nearest code of vector 5 has a component of 70
    embed[:, 300:340] *= 3.0e4                  # dead codes of a trained codebook
    embed[:, 17] = embed[:, 16] * 1.0e5
    embed[:, 320] = 0
    embed[3, 320] = 200.0                       # a far code of moderate norm ...
    vec[11] = 0
    vec[11, 3] = 198.0                          # ... and a vector next to it: no certificate, decided by the fp32 scan
    want = O.quantize(vec, embed)[2]
    codes, e2 = _ops.pack_codebook(embed.to(dev))
    q0, d0, i0, p0 = _ops.vq_nearest(vec.to(dev), codes, e2)
    assert i0[5] == 200 and torch.equal(i0.cpu(), want)
    q1, d1, i1, p1 = _ops.vq_nearest(vec.to(dev), codes, e2, split_f16=True)
    assert torch.equal(i1.cpu(), want) and i1[5] == 200 and i1[11] == 320
    assert torch.equal(q1, q0) and abs(float(d1) - float(d0)) <= 1e-6 * abs(float(d0))
    bad = embed.clone()
    bad[3, 100] = float("inf")
    codes_b, e2_b = _ops.pack_codebook(bad.to(dev))
    qb, db, ib, pb = _ops.vq_nearest(vec.to(dev), codes_b, e2_b, split_f16=True)
    assert (ib == -1).all() and torch.isnan(db), "a non-finite code must be loud in the split-f16 search"

    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=2)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 32, 64, generator=g))
    sd["quantize_t.embed"][:, 11] *= 1.0e5       # a dead code: the model stays in split_f16, indices as the oracle's
    sd["quantize_b.embed"][:, 40:60] *= 3.0e4
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = torch.randn(2, 2, 32, 64, generator=g)
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")
        out = m(x.to(dev))
    ref = O.forward(x, sd, cfg)
    assert torch.equal(out[4].cpu(), ref[4]) and torch.equal(out[5].cpu(), ref[5])
    sd["quantize_t.embed"][3, 12] = float("nan")
    m2 = VQVAE(in_channel=2)
    m2.load_state_dict(sd)
    m2 = m2.to(dev).eval()
    with pytest.warns(UserWarning, match="split_bf16"):
        m2(x.to(dev))


def _certify_index_mismatches(z_vecs, embed, got, ref, eps=2e-6):
    """Any index that differs from the reference must be a near-tie: the fp64
    distances of the two candidates differ by less than `eps` of the magnitude
    of the terms the reference's fp32 formula |z|^2 - 2 z.e + |e|^2 cancels
    (bottleneck.py:56-60 evaluates the distance with absolute error ~ulp(|z|^2))."""
    bad = (got != ref).reshape(-1).nonzero().reshape(-1)
    if bad.numel() == 0:
        return 0
    flat = z_vecs.reshape(-1, z_vecs.shape[-1]).double()[bad]
    e = embed.double()
    x2 = flat.pow(2).sum(1, keepdim=True)
    d = x2 - 2 * flat @ e + e.pow(2).sum(0, keepdim=True)
    dg = d.gather(1, got.reshape(-1)[bad].unsqueeze(1))
    dr = d.gather(1, ref.reshape(-1)[bad].unsqueeze(1))
    scale = x2 + e.pow(2).sum(0)[ref.reshape(-1)[bad]].unsqueeze(1)
    rel = ((dg - dr).abs() / scale).max().item()
    assert rel < eps, f"index mismatch that is not a near-tie: normalised distance gap {rel:.3e}"
    return bad.numel()


@pytest.mark.parametrize("name", ["vqvae_small.npz", "vqvae_default_tiny.npz", "vqvae_f8_f4.npz", "vqvae_f16_f2.npz", "vqvae_groups2.npz"])
def test_vqvae_against_reference(golden_dir, name):
    z = np.load(golden_dir / name)
    m = _model_from_golden(z)
    x = torch.from_numpy(z["x"]).to(_dev())
    q_t, q_b, diff, id_t, id_b, p_t, p_b = m.encode(x)
    assert q_t.shape == z["quant_t"].shape and id_t.dtype == torch.int64
    # A top index may differ from the reference's only at a certified near-tie (these fixtures draw their codebooks
    # from a handful of encoder outputs: distinct codes 1e-7 apart exist, and the reference's own fp32 rounding
    # decides between them -- vqvae_default_tiny has one such vector).  Everything behind a moved top code (dec_t,
    # the bottom level, the reconstruction) legitimately differs, so those comparisons are then teacher-forced.
    moved = _certify_index_mismatches(torch.from_numpy(z["z_t"]).permute(0, 2, 3, 1), torch.from_numpy(z["w::quantize_t.embed"]),
                                      id_t.cpu(), torch.from_numpy(z["id_t"]))
    # the stacks on their own: whatever the codes do (the default-constructor fixture takes the moved-code branch)
    enc_b = m.enc_b(x)
    enc_t = m.enc_t(enc_b)
    _close(enc_b, z["enc_b"], TOL, "enc_b")
    _close(enc_t, z["enc_t"], TOL, "enc_t")
    if moved:
        assert moved <= 1, moved
        dev = _dev()
        same = (id_t.cpu() == torch.from_numpy(z["id_t"]))
        _close(q_t.cpu().permute(0, 2, 3, 1)[same], torch.from_numpy(z["quant_t"]).permute(0, 2, 3, 1)[same].numpy(), TOL, "quant_t")
        ref_t, ref_b = torch.from_numpy(z["id_t"]).to(dev), torch.from_numpy(z["id_b"]).to(dev)
        _close(m.decode_code(ref_t, ref_b), z["dec_code"], TOL, "decode_code(reference ids)")
        # ---- everything behind the moved code, TEACHER-FORCED on the reference's top ids (vqvae.py:258-278): the top
        # code vectors, dec_t, the bottom 1x1 + search (ids, quantised map, perplexity, commitment term), the decoder
        qt_ref = m.quantize_t.embed_code(ref_t).permute(0, 3, 1, 2).contiguous()
        _close(qt_ref, z["quant_t"], TOL, "embed_code(reference id_t)")
        cat = torch.cat([m.dec_t(qt_ref), enc_b], 1).permute(0, 2, 3, 1)
        w_b = m.quantize_conv_b.weight[:, :, 0, 0]
        z_b = (cat.reshape(-1, cat.shape[-1]).double() @ w_b.double().t() + m.quantize_conv_b.bias.double()).float().reshape(*cat.shape[:3], -1)
        qb, diff_b, idb, pb = m.quantize_b(z_b)
        n_b = _certify_index_mismatches(z_b.cpu(), torch.from_numpy(z["w::quantize_b.embed"]), idb.cpu(), torch.from_numpy(z["id_b"]))
        assert n_b <= 1, n_b
        if n_b == 0:
            _close(qb.permute(0, 3, 1, 2), z["quant_b"], TOL, "quant_b (teacher-forced)")
            _close(pb, z["perplexity_b"], TOL, "perplexity_b (teacher-forced)")
            z_t = torch.from_numpy(z["z_t"]).to(dev)
            diff_t = (qt_ref - z_t).pow(2).mean()
            _close((diff_t + diff_b).reshape(1), z["diff"], TOL, "diff (teacher-forced)")
            _close(m.decode(qt_ref, qb.permute(0, 3, 1, 2)), z["dec"], TOL, "decode (teacher-forced)")
        # perplexity_t is a function of the code histogram alone (bottleneck.py:96-100): the GPU's figure against that of its own ids
        onehot = torch.nn.functional.one_hot(id_t.reshape(-1), m.n_embed_t).float().mean(0)
        _close(p_t, torch.exp(-(onehot * torch.log(onehot.clamp(min=1e-7))).sum()), TOL, "perplexity_t of the GPU's ids")
        return
    assert torch.equal(id_b.cpu(), torch.from_numpy(z["id_b"]))
    _close(q_t, z["quant_t"], TOL, "quant_t"); _close(q_b, z["quant_b"], TOL, "quant_b")
    _close(diff, z["diff"], TOL, "diff")
    _close(p_t, z["perplexity_t"], TOL, "perplexity_t"); _close(p_b, z["perplexity_b"], TOL, "perplexity_b")
    dec, diff2, p_t2, p_b2, id_t2, id_b2 = m(x)
    assert torch.equal(id_t2, id_t) and torch.equal(id_b2, id_b)
    _close(dec, z["dec"], TOL, "dec")
    _close(m.decode_code(id_t, id_b), z["dec_code"], TOL, "decode_code")
    _close(m.decode(q_t, q_b), z["dec"], TOL, "decode")


def test_vqvae_against_oracle_seeded_odd_width():
    """Seeded input whose bottom width is odd: exercises adapt_quantized_durations
    (vqvae.py:266-269) and ragged M tiles."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8,
                   embed_dim=16, num_embeddings=64)
    sd = O.init_state_dict(cfg, seed=7)
    g = torch.Generator().manual_seed(8)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 40, 52, generator=g))
    x = torch.randn(3, 2, 40, 52, generator=g)       # bottom 10x13, top 5x6 -> cropped to 12
    ref = O.forward(x, sd, cfg)
    m = VQVAE(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8,
              embed_dim=16, num_embeddings=64)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    dec, diff, p_t, p_b, id_t, id_b = m(x.to(_dev()))
    assert dec.shape == ref[0].shape == (3, 2, 40, 48)
    assert torch.equal(id_t.cpu(), ref[4]) and torch.equal(id_b.cpu(), ref[5])
    _close(dec, ref[0], TOL, "dec"); _close(diff, ref[1], TOL, "diff")
    _close(p_t, ref[2], TOL, "perplexity_t"); _close(p_b, ref[3], TOL, "perplexity_b")


def test_unequal_codebook_sizes_keep_their_histograms_apart():
    """num_embeddings = (544, 512) on the default constructor (vqvae_run.cpp): the two levels count their codes
    in separate histograms of one workspace buffer, zeroed by one launch -- each level's perplexity must be that of ITS
    ids, and the whole forward equals the oracle's.  (At D = 64 every codebook the two-launch search can hold also fits
    the fused kernel, so a fused bottom level behind an unfused top level cannot be constructed; the unfused top level
    nevertheless writes its own histogram now.)"""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 64, 128, generator=g))
    # 32 more top codes: jittered copies of calibrated ones (in use, well apart from them)
    e = sd["quantize_t.embed"]
    sd["quantize_t.embed"] = torch.cat([e, e[:, :32] * 1.25 + 0.05 * torch.randn(64, 32, generator=g)], 1).contiguous()
    sd["quantize_t.embed_avg"] = sd["quantize_t.embed"].clone()
    sd["quantize_t.cluster_size"] = torch.zeros(544)
    x = torch.randn(3, 2, 64, 128, generator=g)
    ref = O.forward(x, sd, cfg)
    for fused in (True, False):
        m = VQVAE(in_channel=2, num_embeddings=(544, 512))
        m.load_state_dict(sd)
        m = m.to(_dev()).eval()
        from interactive_spectrogram_inpainting import _hip
        with _hip.knob("ISI_NO_VQ_FUSION", 0 if fused else 1):
            dec, diff, p_t, p_b, id_t, id_b = m(x.to(_dev()))
            q_t, q_b, diff2, id_t2, id_b2, p_t2, p_b2 = m.encode(x.to(_dev()))    # (the fp32 maps requested: same launches)
        assert int(id_t.max()) < 544 and id_t.shape == ref[4].shape
        if torch.equal(id_t.cpu(), ref[4]) and torch.equal(id_b.cpu(), ref[5]):
            _close(dec, ref[0], TOL, "dec"); _close(diff, ref[1], TOL, "diff")
            _close(p_t, ref[2], TOL, "perplexity_t"); _close(p_b, ref[3], TOL, "perplexity_b")
        # whatever a near-tie did to single codes: each level's perplexity is that of ITS OWN ids (bottleneck.py:96-100)
        for ids, K, got, what in ((id_t, 544, p_t, "perplexity_t"), (id_b, 512, p_b, "perplexity_b")):
            pr = torch.nn.functional.one_hot(ids.reshape(-1), K).float().mean(0)
            _close(got, torch.exp(-(pr * torch.log(pr.clamp(min=1e-7))).sum()), 1e-5, what + " of the level's own ids")
        assert torch.equal(id_t2, id_t) and torch.equal(id_b2, id_b)
        _close(p_b2, p_b, 1e-6, "perplexity_b encode vs forward")


def test_invalidate_plan_after_data_writes():
    """Packed weights / the native plan are keyed on tensor versions, which writes through `.data` do not bump
    (ADVICE r01): `invalidate_plan()` makes such a write visible; a fresh model with the same state gives the same bits."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16, num_embeddings=64)
    sd = O.init_state_dict(O.Config(**kw), seed=3)
    m = VQVAE(**kw)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    x = torch.randn(2, 2, 32, 32, generator=torch.Generator().manual_seed(4)).to(_dev())
    before = m(x)[0].clone()
    for prm in m.parameters():
        prm.data.mul_(0.5)                       # does not bump ._version
    m.invalidate_plan()
    after = m(x)[0]
    assert not torch.equal(before, after)
    fresh = VQVAE(**kw)
    fresh.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    fresh = fresh.to(_dev()).eval()
    assert torch.equal(fresh(x)[0], after)


@pytest.mark.parametrize("precision", ["split_f16", "f32"])
def test_vqvae_full_size_properties(precision):
    """BASELINE config 2 (B=64, [2,128,512], default constructor): size-independent properties, and a TEACHER-FORCED
    oracle comparison of every stage on 8 of the 64 samples, so that a moved near-tie index at one level never hides
    the levels behind it:
      top     oracle z_t from x            -> every GPU id_t that differs from the oracle's is a certified near-tie
      bottom  oracle z_b from the GPU's id_t (embed_code -> dec_t -> cat(enc_b) -> 1x1) -> same for id_b
      decoder oracle decode_code(GPU id_t, GPU id_b) against the GPU reconstruction, <= 1e-4 of the maximum
    in the default split-f16 product mode and on the exact-fp32 matrix pipe."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 128, 512, generator=g))
    x = torch.randn(64, 2, 128, 512, generator=g)
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    m.conv_precision = precision
    xd = x.to(_dev())
    dec, diff, p_t, p_b, id_t, id_b = m(xd)
    assert dec.shape == (64, 2, 128, 512) and id_t.shape == (64, 16, 64) and id_b.shape == (64, 32, 128)
    assert torch.isfinite(dec).all()
    assert int(id_t.min()) >= 0 and int(id_t.max()) < 512 and int(id_b.min()) >= 0 and int(id_b.max()) < 512
    assert p_t.item() > 20 and p_b.item() > 20, "calibrated codebooks must be in use"
    # determinism / batch independence: the first 2 samples alone give the same codes and output
    dec2, _, _, _, id_t2, id_b2 = m(xd[:2])
    assert torch.equal(id_t2, id_t[:2]) and torch.equal(id_b2, id_b[:2])
    assert torch.equal(dec2, dec[:2])
    # decode_code(encode(x)) reproduces forward's reconstruction
    _close(m.decode_code(id_t, id_b), dec, 1e-5, "decode_code vs forward")
    # encode is idempotent on codes: quantised vectors are codebook rows
    q_t, q_b, *_ = m.encode(xd[:4])
    rows = m.quantize_b.embed.t()[id_b[:4]]
    _close(q_b.permute(0, 2, 3, 1), rows, 1e-6, "quant_b rows")
    # ---- teacher-forced oracle comparison on samples spread over the batch
    pick = torch.tensor([0, 1, 9, 18, 27, 36, 50, 63])
    xs = x[pick]
    gt, gb = id_t[pick.to(_dev())].cpu(), id_b[pick.to(_dev())].cpu()
    F = torch.nn.functional
    enc_b = O.encoder(xs, sd, "enc_b.", 4, 2)
    enc_t = O.encoder(enc_b, sd, "enc_t.", 2, 2)
    z_t = F.conv2d(enc_t, sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
    ref_t = O.quantize(z_t, sd["quantize_t.embed"])[2]
    # north_star: "bit-exact code indices".  The calibrated model is non-degenerate (oracle.calibrate_codebooks:
    # standardised pre-quantisation vectors, well-separated codes), so no vector sits on a rounding coin toss of the
    # reference's own fp32 formula: the exact-fp32 mode must EQUAL the oracle, the split-f16 mode (f16-pipe candidates,
    # fp32 decision) may move at most 2 certified near-ties.
    n_t = _certify_index_mismatches(z_t, sd["quantize_t.embed"], gt, ref_t)
    if precision == "f32":
        assert torch.equal(gt, ref_t), f"{n_t} of {ref_t.numel()} top indices differ from the oracle"
    assert n_t <= 2, f"{n_t} of {ref_t.numel()} top indices differ from the oracle"
    q_t_forced = O.embed_code(gt, sd["quantize_t.embed"]).permute(0, 3, 1, 2)
    cat = torch.cat([O.decoder(q_t_forced, sd, "dec_t.", 2, 2), enc_b], 1)
    z_b = F.conv2d(cat, sd["quantize_conv_b.weight"], sd["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
    ref_b = O.quantize(z_b, sd["quantize_b.embed"])[2]
    n_b = _certify_index_mismatches(z_b, sd["quantize_b.embed"], gb, ref_b)
    if precision == "f32":
        assert torch.equal(gb, ref_b), f"{n_b} of {ref_b.numel()} bottom indices differ from the oracle"
    assert n_b <= 2, f"{n_b} of {ref_b.numel()} bottom indices differ from the oracle"
    _close(dec[pick.to(_dev())], O.decode_code(gt, gb, sd, cfg), TOL, "dec vs oracle decode_code(GPU codes)")
    print(f"[{precision}] indices differing from the torch-CPU oracle on 8 spectrograms: top {n_t}/{ref_t.numel()}, "
          f"bottom (teacher-forced) {n_b}/{ref_b.numel()}")


def test_bench_batch_codes_certified_against_oracle():
    """bench.py's own model and batch (B = 64, default split-f16 products): the codes of ALL 64 spectrograms against the
    CPU oracle, level by level (bottom teacher-forced on the GPU's top codes) -- every code that differs must be a
    certified near-tie of the reference's fp32 distance formula (normalised float64 gap < 1e-6) and there may be only a
    handful of them, in the exact-fp32 mode as well.  bench.py prints the same
    figures as `codes_moved_vs_oracle` / `certified_near_ties`."""
    import pathlib
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    import bench
    from oracle import vqvae_oracle as O
    m, sd = bench._build_model(_dev())
    cfg = O.Config(in_channel=2)
    x = torch.randn(64, 2, 128, 512, generator=torch.Generator().manual_seed(100))
    xd = x.to(_dev())
    for mode in ("split_f16", "f32"):
        m.conv_precision = mode
        with torch.no_grad():
            out = m(xd)
        chk = O.teacher_forced_code_check(x, sd, cfg, out[4].cpu(), out[5].cpu(), eps=1e-6)
        print(f"[{mode}] {chk}")
        assert chk["samples"] == 64 and chk["of_top"] == 64 * 16 * 64 and chk["of_bottom"] == 64 * 32 * 128
        assert chk["certified_near_ties"], chk
        # (64 x 5120 vectors: a few sit within float rounding of a tie of the reference's own formula -- torch-CPU's and
        # the GPU's fp32 convolutions sum in different orders -- in the exact-fp32 mode too: 3 top codes on this batch)
        # (allowances at what is observed, VERDICT r04 item 8: split_f16 moves 4 top / 0 bottom codes of this batch, the
        # largest normalised gap is 6.2e-7)
        # observed + 1: split_f16 4 top / 0 bottom, exact fp32 3 top / 0 bottom
        lim = {"split_f16": (5, 1), "f32": (4, 1)}[mode]
        assert chk["top_moved"] <= lim[0] and chk["bottom_moved_teacher_forced"] <= lim[1], chk


def test_extract_rows(golden_dir):
    """extract_code.extract contract: one CodeRow per sample with the codes of VQVAE.encode."""
    import extract_code as E
    z = np.load(golden_dir / "vqvae_small.npz")
    m = _model_from_golden(z)
    x = torch.from_numpy(z["x"])
    names = [f"note_{i}" for i in range(x.shape[0])]
    loader = [(x, torch.arange(x.shape[0]), {"note_str": names})]
    rows = {}
    n = E.extract(loader, m, _dev(), lambda k, r: rows.__setitem__(k, r), label_encoders={"pitch": None})
    assert n == x.shape[0] and list(rows) == names
    for i, name in enumerate(names):
        r = rows[name]
        assert r.filename == name and r.attributes == {"pitch": loader[0][1][i]}
        assert r.top.dtype == np.int64 and np.array_equal(r.top, z["id_t"][i]) and np.array_equal(r.bottom, z["id_b"][i])


def test_split_bf16_precision_modes():
    """Split-bf16 products: the default decoder-only mode must leave every code index untouched and
    the reconstruction within 1e-4 of the exact-fp32 path (north_star: 1e-3); the all-layers mode may
    only move indices that are near-ties."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 64, 128, generator=g))
    x = torch.randn(4, 2, 64, 128, generator=g).to(_dev())
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    assert m.conv_precision == "split_f16"
    m.conv_precision = "f32"
    ref = m(x)
    m.conv_precision = "bf16x3_decoder"
    got = m(x)
    assert torch.equal(got[4], ref[4]) and torch.equal(got[5], ref[5])
    _close(got[0], ref[0], 1e-4, "dec (bf16x3 decoder)")
    assert not torch.equal(got[0], ref[0]), "the mode switch must actually change the arithmetic"
    m.conv_precision = "bf16x3"
    got = m(x)
    agree_t = (got[4] == ref[4]).float().mean().item()
    agree_b = (got[5] == ref[5]).float().mean().item()
    assert agree_t > 0.98 and agree_b > 0.98, (agree_t, agree_b)
    # default: six-term split where indices are decided (fp32-grade: indices may differ from the fp32 pipe's
    # only at near-ties, as any two fp32 implementations do), three-term split in the final decoder
    m.conv_precision = "split_bf16"
    got = m(x)
    agree_t = (got[4] == ref[4]).float().mean().item()
    agree_b = (got[5] == ref[5]).float().mean().item()
    assert agree_t > 0.995 and agree_b > 0.995, (agree_t, agree_b)
    # default: three-term split-f16 everywhere, fp32-grade like the six-term bf16 split
    m.conv_precision = "split_f16"
    got = m(x)
    agree_t = (got[4] == ref[4]).float().mean().item()
    agree_b = (got[5] == ref[5]).float().mean().item()
    assert agree_t > 0.995 and agree_b > 0.995, (agree_t, agree_b)
    same = (got[4] == ref[4]).all(-1).all(-1) & (got[5] == ref[5]).all(-1).all(-1)
    if same.any():
        _close(got[0][same], ref[0][same], 2e-6, "dec (split_f16, items with identical codes)")
    assert not torch.equal(got[0], ref[0]), "the mode switch must actually change the arithmetic"
    m.conv_precision = "f32"
    assert torch.equal(m(x)[0], ref[0])


def test_split_f16_products_against_fp64():
    """ISI_CONV_F16X3: fp32-grade products (checked against fp64 next to the exact-fp32 pipe and the six-term
    bf16 split) over five decades of operand magnitude inside the documented range, for the implicit-GEMM
    convolution and the fused residual block."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, C, H, W, R = 2, 128, 24, 40, 32
    w = torch.randn(C, C, 3, 3, generator=g) * 0.03
    b = torch.randn(C, generator=g) * 0.1
    w3 = torch.randn(R, C, 3, 3, generator=g) * 0.03
    b3 = torch.randn(R, generator=g) * 0.1
    w1 = torch.randn(C, R, 1, 1, generator=g) * 0.1
    b1 = torch.randn(C, generator=g) * 0.1
    pw, p3, p1 = (_ops.pack_conv_weight(t.to(dev), with_f16=True) for t in (w, w3, w1))
    F = torch.nn.functional
    for scale in (1e-3, 1e-1, 1.0, 30.0, 3000.0):
        x = torch.randn(B, C, H, W, generator=g).abs() * scale     # max ~ 4.5 scale < 16384
        xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        err = {}
        for mode in (0, 2, 3):
            y = _ops.conv2d(xd, pw, b.to(dev), C, 3, 1, 1, relu=False, bf16x3=mode).cpu().double()
            err[mode] = ((y - ref).abs().max() / ref.abs().max()).item()
        assert err[3] < 3e-6 and err[3] < 2.0 * max(err[0], err[2]) + 1e-7, (scale, err)
        # weights' pieces prepared at pack time (ISI_CONV_W16): the same bits
        assert torch.equal(_ops.conv2d(xd, pw, b.to(dev), C, 3, 1, 1, relu=False, bf16x3=4),
                           _ops.conv2d(xd, pw, b.to(dev), C, 3, 1, 1, relu=False, bf16x3=3))
        h = torch.relu(F.conv2d(x.double(), w3.double(), b3.double(), padding=1))
        ref = torch.relu(x.double() + F.conv2d(h, w1.double(), b1.double()))
        for mode in (0, 2, 3):
            y = _ops.resblock(xd, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=mode).cpu().double()
            err[mode] = ((y - ref).abs().max() / ref.abs().max()).item()
        assert err[3] < 3e-6 and err[3] < 2.0 * max(err[0], err[2]) + 1e-7, (scale, err)
        assert torch.equal(_ops.resblock(xd, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=4),
                           _ops.resblock(xd, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=3))
    # transposed convolution (4 phase matrices)
    wt = torch.randn(C, 64, 4, 4, generator=g) * 0.05
    pt = _ops.pack_convT_weight(wt.to(dev), with_f16=True)
    y3 = _ops.conv_transpose2d_k4s2(xd, pt, None, 64, relu=True, bf16x3=3)
    assert torch.equal(_ops.conv_transpose2d_k4s2(xd, pt, None, 64, relu=True, bf16x3=4), y3)
    ref = torch.relu(F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1))
    assert ((y3.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 3e-6


@pytest.mark.parametrize("B,H,W,cout", [(2, 16, 32, 64), (3, 10, 18, 32), (1, 128, 512, 64), (5, 6, 250, 64)])
def test_first_layer_kernel(B, H, W, cout):
    """conv_first_f32.hip (Conv2d(2 -> 32 / 64, k4 s2 p1) + ReLU on the NCHW input): bit-identical to the generic
    gather kernel it replaces (same k pairing and order on the exact-fp32 pipe), within 1e-6 of torch-CPU, ragged
    tiles, pair-format output, non-contiguous input view."""
    import os
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = torch.randn(B, 2, H, W, generator=g)
    w = torch.randn(cout, 2, 4, 4, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    pw = _ops.pack_conv_weight(w.to(dev))
    xd = x.to(dev)
    got = _ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True)
    with _hip.knob("ISI_NO_CONV_FIRST", 1):
        generic = _ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True)
    assert torch.equal(got, generic)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=2, padding=1))
    _close(got, ref, 2e-6, "first layer")
    assert torch.equal(_ops.conv2d(xd, pw, None, cout, 4, 2, 1, relu=False),
                       _ops.conv2d(xd, pw, torch.zeros(cout, device=dev), cout, 4, 2, 1, relu=False))
    pair = _ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True, extra_flags=_ops.PAIR_OUT)
    assert (_ops.pair_decode(pair) - got).abs().max() <= 2.0 ** -23 * got.abs().max()
    # split-f16 variant (ISI_CONV_F16X3): fp32-grade against fp64, pair output, NaN for an out-of-range input
    ref64 = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    f16 = _ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True, bf16x3=3)
    err16 = ((f16.cpu().double() - ref64).abs().max() / ref64.abs().max()).item()
    err32 = ((got.cpu().double() - ref64).abs().max() / ref64.abs().max()).item()
    assert err16 < 1e-6 and err16 < 3 * err32 + 1e-7, (err16, err32)
    assert not torch.equal(f16, got)
    pair = _ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True, bf16x3=3, extra_flags=_ops.PAIR_OUT)
    assert (_ops.pair_decode(pair) - f16).abs().max() <= 2.0 ** -23 * f16.abs().max()
    assert torch.equal(_ops.conv2d(xd, pw, b.to(dev), cout, 4, 2, 1, relu=True, bf16x3=2), got)   # six-term: exact kernel
    xbad = xd.clone()
    xbad[0, 1, 3, 5] = 5e4
    ybad = _ops.conv2d(xbad, pw, b.to(dev), cout, 4, 2, 1, relu=True, bf16x3=3)
    assert not torch.isfinite(ybad[0, :, 1:3, 2:4]).any() and torch.isfinite(ybad[0, :, 4:, :]).all()
    wide = torch.randn(B, 2, H, W + 6, generator=g).to(dev)
    view = wide[..., 3:W + 3]
    assert torch.equal(_ops.conv2d(view, pw, b.to(dev), cout, 4, 2, 1, relu=True),
                       _ops.conv2d(view.contiguous(), pw, b.to(dev), cout, 4, 2, 1, relu=True))


def test_split_f16_pair_format_activations():
    """ISI_CONV_OUT_PAIR / IN*_PAIR: a producer writes hi | lo << 16 per element, a split-f16 consumer de-interleaves
    instead of converting.  Same matrix operands: a consumer fed pairs returns the bits it returns for the fp32
    tensor; a producer's pair output decodes to its fp32 output within 2^-23; launches that cannot read pairs refuse."""
    import os
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    B, C, H, W, R = 2, 128, 12, 40, 32
    F = torch.nn.functional
    x = torch.relu(torch.randn(B, C, H, W, generator=g))
    x2 = torch.randn(B, 64, H, W, generator=g)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    x2d = x2.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xp = _ops.pair_encode(xd.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    x2p = _ops.pair_encode(x2d.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    back = _ops.pair_decode(xp)
    assert (back - xd).abs().max() <= 2.0 ** -23 * xd.abs().max()
    w = torch.randn(C, C, 3, 3, generator=g) * 0.03
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    pw = _ops.pack_conv_weight(w.to(dev), with_f16=True)
    # Pair-format sources run the LDS-DMA kernel (csrc/conv_pair_f16.hip).  Its accumulator flush (pairwise-style
    # summation) changes rounding: with ISI_CONV_FLUSH=0 it returns the bits of the register-staged kernel, with the
    # default period it must be MORE accurate against fp64.
    ref64 = torch.relu(F.conv2d(x.double(), w.double(), bias.cpu().double(), padding=1))
    ref = _ops.conv2d(xd, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4)
    flushed = _ops.conv2d(xp, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    err_old = ((ref.cpu().double() - ref64).abs().max() / ref64.abs().max()).item()
    err_new = ((flushed.cpu().double() - ref64).abs().max() / ref64.abs().max()).item()
    assert err_new < 0.6 * err_old and err_new < 4e-7, (err_new, err_old)
    with _hip.knob("ISI_CONV_FLUSH", 0):
        _pair_format_bit_identity(_ops, _hip, dev, g, xd, x2d, xp, x2p, pw, bias, ref, C, R, B)


def _pair_format_bit_identity(_ops, _hip, dev, g, xd, x2d, xp, x2p, pw, bias, ref, C, R, B):
    F = torch.nn.functional
    got = _ops.conv2d(xp, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    assert torch.equal(got, ref)
    out_p = _ops.conv2d(xp, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
    assert (_ops.pair_decode(out_p) - ref).abs().max() <= 2.0 ** -23 * ref.abs().max()
    # two sources, the first in pair format, the second fp32 (decoder conv3: cat(upsampled top, quant_b))
    w2 = torch.randn(C, C + 64, 3, 3, generator=g) * 0.03
    pw2 = _ops.pack_conv_weight(w2.to(dev), with_f16=True)
    ref = _ops.conv2d(xd, pw2, bias, C, 3, 1, 1, relu=False, x2_bchw=x2d, bf16x3=4)
    assert torch.equal(_ops.conv2d(xp, pw2, bias, C, 3, 1, 1, relu=False, x2_bchw=x2d, bf16x3=4,
                                   extra_flags=_ops.PAIR_IN0), ref)
    assert torch.equal(_ops.conv2d(xp, pw2, bias, C, 3, 1, 1, relu=False, x2_bchw=x2p, bf16x3=4,
                                   extra_flags=_ops.PAIR_IN0 | _ops.PAIR_IN1), ref)
    # strided 4x4 and transposed convolution
    w4 = torch.randn(C, C, 4, 4, generator=g) * 0.02
    p4 = _ops.pack_conv_weight(w4.to(dev), with_f16=True)
    assert torch.equal(_ops.conv2d(xp, p4, bias, C, 4, 2, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0),
                       _ops.conv2d(xd, p4, bias, C, 4, 2, 1, relu=True, bf16x3=4))
    wt = torch.randn(C, 64, 4, 4, generator=g) * 0.05
    pt = _ops.pack_convT_weight(wt.to(dev), with_f16=True)
    ref = _ops.conv_transpose2d_k4s2(xd, pt, None, 64, relu=True, bf16x3=4)
    # (pair-format input takes the fused-phase LDS-DMA kernel, csrc/convT_pair_f16.hip: another K order, same products)
    got_t = _ops.conv_transpose2d_k4s2(xp, pt, None, 64, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    assert (got_t - ref).abs().max() <= 2.0 ** -20 * ref.abs().max()
    with _hip.knob("ISI_NO_CONVT_PAIR_KERNEL", 1):
        assert torch.equal(_ops.conv_transpose2d_k4s2(xp, pt, None, 64, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0), ref)
    tp = _ops.conv_transpose2d_k4s2(xp, pt, None, 64, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT)
    assert (_ops.pair_decode(tp) - ref).abs().max() <= 2.0 ** -20 * ref.abs().max()
    # fused residual block: the skip connection reads (hi + lo) / 4
    w3 = torch.randn(R, C, 3, 3, generator=g) * 0.03
    b3 = (torch.randn(R, generator=g) * 0.1).to(dev)
    w1 = torch.randn(C, R, 1, 1, generator=g) * 0.1
    b1 = (torch.randn(C, generator=g) * 0.1).to(dev)
    p3, p1 = _ops.pack_conv_weight(w3.to(dev), with_f16=True), _ops.pack_conv_weight(w1.to(dev), with_f16=True)
    ref = _ops.resblock(xd, p3, b3, p1, b1, R, True, bf16x3=4)
    got = _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
    assert (got - ref).abs().max() <= 2.0 ** -22 * ref.abs().max()
    got = _ops.pair_decode(_ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT))
    assert (got - ref).abs().max() <= 2.0 ** -22 * ref.abs().max()
    # a first-layer (gather) convolution can WRITE pairs; launches that cannot READ them refuse
    xin = torch.randn(B, 2, 16, 32, generator=g).to(dev)
    wf = torch.randn(64, 2, 4, 4, generator=g) * 0.2
    pf = _ops.pack_conv_weight(wf.to(dev), with_f16=True)
    ref = _ops.conv2d(xin, pf, None, 64, 4, 2, 1, relu=True, bf16x3=4)
    got = _ops.pair_decode(_ops.conv2d(xin, pf, None, 64, 4, 2, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_OUT))
    assert (got - ref).abs().max() <= 2.0 ** -23 * ref.abs().max()
    for kw in (dict(bf16x3=3), dict(bf16x3=0), dict(bf16x3=2)):
        with pytest.raises(_hip.HipLibraryError):
            _ops.conv2d(xp, pw, bias, C, 3, 1, 1, relu=True, extra_flags=_ops.PAIR_IN0, **kw)
    # fp32 source, pair-format output (register-staged kernel, 2-byte stores of the pieces)
    out_p = _ops.conv2d(xd, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4, extra_flags=_ops.PAIR_OUT)
    ref_c = _ops.conv2d(xd, pw, bias, C, 3, 1, 1, relu=True, bf16x3=4)
    assert (_ops.pair_decode(out_p) - ref_c).abs().max() <= 2.0 ** -23 * ref_c.abs().max()
    with pytest.raises(_hip.HipLibraryError):
        _ops.resblock(xp, p3, b3, p1, b1, R, True, bf16x3=3, extra_flags=_ops.PAIR_IN0)


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 40, 128), (1, 9, 70, 128), (3, 33, 65, 128), (2, 17, 64, 64)])
@pytest.mark.parametrize("th", ["4", "8"])
def test_dma_residual_block_ragged_shapes(B, H, W, C, th):
    """resblock_pair_kernel (LDS-DMA staging, pair-format input; csrc/resblock_pair_f16.hip) on maps that are not
    multiples of its 4 x 64 / 8 x 64 tiles, both row counts forced: against the reference block in float64
    (encoder_decoder.py:18-35, in-place ReLU semantics) and against the register-staged kernel."""
    import os
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(100 + H + W)
    R = 32
    F = torch.nn.functional
    x = torch.relu(torch.randn(B, C, H, W, generator=g))
    w3 = torch.randn(R, C, 3, 3, generator=g) * 0.03
    b3 = torch.randn(R, generator=g) * 0.1
    w1 = torch.randn(C, R, 1, 1, generator=g) * 0.1
    b1 = torch.randn(C, generator=g) * 0.1
    ref64 = torch.relu(x.double() + F.conv2d(torch.relu(F.conv2d(x.double(), w3.double(), b3.double(), padding=1)),
                                             w1.double(), b1.double()))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xp = _ops.pair_encode(xd.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    p3, p1 = _ops.pack_conv_weight(w3.to(dev), with_f16=True), _ops.pack_conv_weight(w1.to(dev), with_f16=True)
    old = _ops.resblock(xd, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=4)
    with _hip.knob("ISI_RESPAIR_TH", int(th)):
        got = _ops.resblock(xp, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
        got_p = _ops.pair_decode(_ops.resblock(xp, p3, b3.to(dev), p1, b1.to(dev), R, True, bf16x3=4,
                                               extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT))
    scale = ref64.abs().max().item()
    assert (got.cpu().double() - ref64).abs().max().item() < 2e-6 * scale
    assert (got_p.cpu().double() - ref64).abs().max().item() < 2e-6 * scale
    assert (got - old).abs().max().item() < 2e-6 * scale


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 64, 128, 64), (1, 9, 70, 128, 64), (3, 5, 33, 64, 64),
                                            (2, 12, 130, 64, 128), (1, 1, 1, 32, 64), (2, 3, 64, 48, 64)])
@pytest.mark.parametrize("th", [4, 8])
def test_dma_transposed_convolution_ragged_shapes(B, H, W, cin, cout, th):
    """convT_pair_kernel (the four output phases of ConvTranspose2d(k4,s2,p1) fused on one LDS-DMA-staged input tile;
    csrc/convT_pair_f16.hip) on maps that are not multiples of its TH x 64 tiles, both tile heights forced, fp32 and
    pair-format output, bias + ReLU: against torch's ConvTranspose2d in float64 on the pair-rounded input
    (reference call sites vqvae/encoder_decoder.py:196-216, vqvae/vqvae.py:183-201) and against the phase-per-launch
    kernel (same products, other K order)."""
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(7 * H + W + cin)
    x = torch.relu(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cin, cout, 4, 4, generator=g) * 0.04
    bias = torch.randn(cout, generator=g) * 0.1
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xp = _ops.pair_encode(xd)
    pt = _ops.pack_convT_weight(w.to(dev), with_f16=True)
    x_seen = _ops.pair_decode(xp).cpu().double()                 # what the kernel multiplies: (hi + lo) / 4
    ref = torch.nn.functional.conv_transpose2d(x_seen, w.double(), bias.double(), stride=2, padding=1)
    with _hip.knob("ISI_CONVT_PAIR_TH", th):
        for relu in (False, True):
            want = torch.relu(ref) if relu else ref
            got = _ops.conv_transpose2d_k4s2(xp, pt, bias.to(dev), cout, relu=relu, bf16x3=4, extra_flags=_ops.PAIR_IN0)
            assert got.shape == (B, cout, 2 * H, 2 * W)
            err = ((got.cpu().double() - want).abs().max() / want.abs().max()).item()
            assert err < 2e-6, f"fp32 output, relu={relu}: {err:.2e}"
            gp = _ops.pair_decode(_ops.conv_transpose2d_k4s2(xp, pt, bias.to(dev), cout, relu=relu, bf16x3=4,
                                                             extra_flags=_ops.PAIR_IN0 | _ops.PAIR_OUT))
            assert (gp - got).abs().max() <= 2.0 ** -23 * got.abs().max(), "pair output = rounded fp32 output"
    if cin % 32 == 0 and B * H * W >= 4:      # shapes the phase-per-launch kernel reads in the pair format
        with _hip.knob("ISI_NO_CONVT_PAIR_KERNEL", 1):
            old = _ops.conv_transpose2d_k4s2(xp, pt, bias.to(dev), cout, relu=True, bf16x3=4, extra_flags=_ops.PAIR_IN0)
        assert not torch.equal(old, got), "the knob must select the other kernel"
        assert (old - got).abs().max() <= 2.0 ** -20 * got.abs().max()


@pytest.mark.parametrize("B,H,W,C0,C1,cout,k,stride", [
    (1, 5, 7, 128, 0, 128, 3, 1),        # fewer pixels than one 256-pixel tile
    (2, 19, 45, 128, 0, 128, 3, 1),      # ragged last tile
    (2, 18, 50, 64, 64, 128, 3, 1),      # two sources (the decoder's concat)
    (3, 22, 38, 64, 0, 128, 4, 2),       # strided 4x4, even map
    (2, 21, 37, 128, 0, 64, 4, 2),       # strided 4x4 on an odd map, 64 output channels
    (1, 40, 72, 64, 0, 128, 3, 1),
])
def test_dma_convolution_ragged_shapes(B, H, W, C0, C1, cout, k, stride):
    """conv_pair_kernel (LDS-DMA staging of pair-format sources; csrc/conv_pair_f16.hip) on ragged maps, one and two
    sources, both tile widths, stride 1 and 2: against torch in float64 and, with the accumulator flush off, bit for
    bit against the register-staged kernel on fp32 activations; fp32 and pair-format outputs."""
    import os
    from interactive_spectrogram_inpainting.vqvae import _ops
    dev = _dev()
    g = torch.Generator().manual_seed(7 * H + W)
    F = torch.nn.functional
    pad = 1
    x0 = torch.relu(torch.randn(B, C0, H, W, generator=g))
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(cout, C0 + C1, k, k, generator=g) * 0.03
    bias = torch.randn(cout, generator=g) * 0.1
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    ref64 = torch.relu(F.conv2d(xin.double(), w.double(), bias.double(), stride=stride, padding=pad))
    cl = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    enc = lambda t: _ops.pair_encode(t.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    x0d, x1d = cl(x0), (cl(x1) if x1 is not None else None)
    x0p, x1p = enc(x0d), (enc(x1d) if x1d is not None else None)
    pw = _ops.pack_conv_weight(w.to(dev), with_f16=True)
    flags = _ops.PAIR_IN0 | (_ops.PAIR_IN1 if x1 is not None else 0)
    kw = dict(relu=True, bf16x3=4)
    old = _ops.conv2d(x0d, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1d, **kw)
    got = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p, extra_flags=flags, **kw)
    got_p = _ops.pair_decode(_ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p,
                                         extra_flags=flags | _ops.PAIR_OUT, **kw))
    scale = ref64.abs().max().item()
    assert got.shape == ref64.shape
    assert (got.cpu().double() - ref64).abs().max().item() < 1e-6 * scale
    assert (got_p.cpu().double() - ref64).abs().max().item() < 1e-6 * scale
    with _hip.knob("ISI_CONV_FLUSH", 0):
        unflushed = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p, extra_flags=flags, **kw)
    assert torch.equal(unflushed, old)
    # the 128-row, four-wave form (two workgroups per CU) walks K and orders the terms like the 256-row kernel: same bits
    with _hip.knob("ISI_CONV_PAIR_BM", 256):
        big = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p, extra_flags=flags, **kw)
    with _hip.knob("ISI_CONV_PAIR_BM", 128):
        small = _ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p, extra_flags=flags, **kw)
        small_p = _ops.pair_decode(_ops.conv2d(x0p, pw, bias.to(dev), cout, k, stride, pad, x2_bchw=x1p,
                                               extra_flags=flags | _ops.PAIR_OUT, **kw))
    assert torch.equal(small, big) and torch.equal(big, got if True else got)
    assert (small_p.cpu().double() - ref64).abs().max().item() < 1e-6 * scale


def test_vqvae_pair_pipeline_against_fp32_activations():
    """The fused forward keeps its internal activations in the pair format when every layer can read it; with
    ISI_NO_PAIRS it runs the same arithmetic on fp32 activations: bit-identical outputs (ragged width included)."""
    import os
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=4)
    g = torch.Generator().manual_seed(9)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 64, 128, generator=g))
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    assert m.conv_precision == "split_f16"
    x = torch.randn(3, 2, 64, 136, generator=g).to(dev)      # ragged width: cropped bottom grid
    got = m(x)
    from interactive_spectrogram_inpainting import _hip
    import ctypes
    assert _hip.lib().isi_vqvae_pair_activations(ctypes.byref(m._native_weights())) == 1
    small = VQVAE(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
                  num_embeddings=64).to(dev).eval()
    small(x)
    assert _hip.lib().isi_vqvae_pair_activations(ctypes.byref(small._native_weights())) == 0   # 32-channel layers
    with _hip.knob("ISI_NO_PAIRS", 1):
        ref = m(x)
    # The convolutions' matrix operands are the same bits on both paths (with the pair kernel's accumulator flush off,
    # ISI_CONV_FLUSH=0, even the same sums); the residual blocks of the pair pipeline take their skip connection from
    # the pair pieces ((hi + lo) / 4: the value to 2^-24) instead of the fp32 tensor: rounding-level differences
    with _hip.knob("ISI_CONV_FLUSH", 0):
        unflushed = m(x)
    for got_ in (unflushed, got):
        assert (got_[4] != ref[4]).float().mean() < 0.01 and (got_[5] != ref[5]).float().mean() < 0.01
        same_ = (got_[4] == ref[4]).all(-1).all(-1) & (got_[5] == ref[5]).all(-1).all(-1)
        if same_.any():
            _close(got_[0][same_], ref[0][same_], 3e-6, "dec (pair pipeline vs fp32 activations)")
    # quantize_conv_{t,b} fused into the codebook searches (z stays in registers; csrc/vq_nearest.hip): the same
    # products in the same order as the two-launch path -- the same bits
    with _hip.knob("ISI_NO_VQ_FUSION", 1):
        unfused = m(x)
    for k_, (a, b) in enumerate(zip(got, unfused)):
        if k_ == 1:     # diff: a per-lane sum of squares whose fmas the two kernels may contract differently (1 ulp)
            assert abs(float(a) - float(b)) <= 1e-6 * abs(float(b))
        else:
            assert torch.equal(a, b)
    enc_f = m.encode(x)
    with _hip.knob("ISI_NO_VQ_FUSION", 1):
        enc_u = m.encode(x)
    for a, b in zip(enc_f, enc_u):
        assert torch.equal(a, b)

    assert torch.equal(m.decode_code(got[4], got[5]).isfinite().all(), torch.tensor(True, device=dev))
    oref = O.forward(x.cpu(), sd, cfg)
    assert (got[4].cpu() != oref[4]).float().mean() < 0.01 and (got[5].cpu() != oref[5]).float().mean() < 0.01


def test_split_f16_range_violations_are_loud():
    """Operands beyond f16's range must never produce a plausible result: an activation beyond 16384 gives
    NaN in the output (and code index -1, NaN diff in the model); a weight beyond 64 makes the model run in
    split_bf16 (warning) with the usual finite results."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae import _ops
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    C = 64
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    pw = _ops.pack_conv_weight(w.to(dev))
    x = torch.randn(1, C, 16, 16, generator=g)
    x[0, 3, 5, 7] = 7e4
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = _ops.conv2d(xd, pw, None, C, 3, 1, 1, relu=False, bf16x3=3)
    assert not torch.isfinite(y[0, :, 4:7, 6:9]).any(), "every output that reads the out-of-range value"
    assert torch.isfinite(y[0, :, 10:, :]).all(), "and nothing else"
    assert torch.isfinite(_ops.conv2d(xd, pw, None, C, 3, 1, 1, relu=False, bf16x3=2)).all()

    cfg = O.Config(in_channel=2)
    sd = O.init_state_dict(cfg, seed=2)
    m = VQVAE(in_channel=2)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    xin = torch.randn(2, 2, 32, 64, generator=g)
    xin[1] *= 1e5
    out = m(xin.to(dev))
    assert torch.isfinite(out[0][0]).all() and (out[4][0] >= 0).all() and (out[5][0] >= 0).all()
    assert (out[4][1] == -1).any() and not torch.isfinite(out[0][1]).all() and not torch.isfinite(out[1]).all()
    with pytest.raises(IndexError):
        m.decode_code(out[4], out[5])
    with torch.no_grad():
        m.enc_b.blocks[m.enc_b._conv3].weight[0, 0, 0, 0] = 100.0
    with pytest.warns(UserWarning, match="split_bf16"):
        out2 = m(xin[:1].to(dev))
    assert torch.isfinite(out2[0]).all() and (out2[4] >= 0).all()


@pytest.mark.parametrize("in_ch,B,H,W", [(3, 1, 8, 8), (1, 2, 16, 24), (2, 1, 8, 200), (3, 5, 24, 40)])
def test_vqvae_edge_shapes_against_oracle(in_ch, B, H, W):
    """Batch 1, the smallest legal map (top grid 1x1), 1- and 3-channel inputs (the reference's
    default in_channel is 3), long thin maps: ragged tiles everywhere."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    kw = dict(in_channel=in_ch, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=8,
              num_embeddings=32)
    cfg = O.Config(**kw)
    sd = O.init_state_dict(cfg, seed=31 + in_ch)
    g = torch.Generator().manual_seed(32)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, in_ch, 32, 32, generator=g))
    x = torch.randn(B, in_ch, H, W, generator=g)
    ref = O.forward(x, sd, cfg)
    m = VQVAE(**kw)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    dec, diff, p_t, p_b, id_t, id_b = m(x.to(_dev()))
    assert dec.shape == ref[0].shape
    assert torch.equal(id_t.cpu(), ref[4]) and torch.equal(id_b.cpu(), ref[5])
    _close(dec, ref[0], TOL, "dec"); _close(diff, ref[1], TOL, "diff")


def test_vqvae_rejects_what_it_cannot_compute():
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    m = VQVAE(in_channel=2, num_hidden_channels=32, num_residual_channels=8, embed_dim=16, num_embeddings=64).to(_dev()).eval()
    with pytest.raises((_hip.HipLibraryError, RuntimeError)):
        m(torch.randn(1, 2, 4, 4, device=_dev()))             # too small for the 8x down-sampling
    with pytest.raises((_hip.HipLibraryError, RuntimeError)):
        m(torch.randn(0, 2, 32, 32, device=_dev()))           # empty batch
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 3, 32, 32, device=_dev()))           # wrong channel count
    with pytest.raises(_hip.HipLibraryError):
        m(torch.randn(1, 2, 32, 32, device=_dev(), dtype=torch.float64))
    with pytest.raises(IndexError):
        m.decode_code(torch.full((1, 4, 4), 64, dtype=torch.int64, device=_dev()),
                      torch.zeros(1, 8, 8, dtype=torch.int64, device=_dev()))


def test_forward_is_graph_capturable():
    """Nothing inside the library allocates or synchronises: VQVAE.forward can be captured into a HIP graph
    (torch.cuda.CUDAGraph on the capture stream) and replayed on new input data."""
    from oracle import vqvae_oracle as O
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    cfg = O.Config(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
                   num_embeddings=64)
    sd = O.init_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(4)
    O.calibrate_codebooks(sd, cfg, torch.randn(2, 2, 32, 64, generator=g))
    m = VQVAE(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
              num_embeddings=64)
    m.load_state_dict(sd)
    m = m.to(_dev()).eval()
    x1, x2 = (torch.randn(2, 2, 32, 64, generator=g).to(_dev()) for _ in range(2))
    eager1, eager2 = m(x1), m(x2)                # also performs the one-time kernel attribute set-up
    static_x = x1.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(static_x)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = m(static_x)
    for x, eager in ((x1, eager1), (x2, eager2)):
        static_x.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], eager[0]) and torch.equal(out[4], eager[4]) and torch.equal(out[5], eager[5])
