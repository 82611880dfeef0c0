"""GPU parity tests of the audio <-> spectrogram front-end (through the C-ABI) against the
CPU specification oracle/spectrogram_oracle.py (FFT based; the HIP path is GEMM based).
PARITY UNPINNED: the reference's front-end package (GANsynth_pytorch) is absent; the
specification restates the published GANSynth representation.

Tolerances: magnitudes 1e-3 relative (north_star's fp32 bound) — measured ~1e-5; phases are
compared on the circle, and where a wrapped difference sits within rounding of +-pi the two
implementations may legitimately pick opposite signs, so a 1e-4 fraction of outliers is allowed."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _audio(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L) / 16000.0
    x = 0.05 * torch.randn(B, L, generator=g)            # every bin carries energy: phases are well defined
    for b in range(B):
        for f0 in (220.0 * (b + 1), 1234.5, 3999.0):
            x[b] += 0.3 * torch.sin(2 * math.pi * f0 * t + b)
    return x


def _helpers(n_fft, hop, mel):
    from oracle import spectrogram_oracle as S
    from interactive_spectrogram_inpainting.utils.misc import get_spectrograms_helper
    h = get_spectrograms_helper(fs_hz=16000, n_fft=n_fft, hop_length=hop, window_length=n_fft, use_mel_scale=mel,
                                mel_scale_lower_edge_hertz=0.0, mel_scale_upper_edge_hertz=8000.0,
                                mel_scale_break_frequency_hertz=700.0, mel_scale_expand_resolution_factor=1.5)
    return S, S.SpecConfig(n_fft=n_fft, hop_length=hop, window_length=n_fft), h.to(_dev())


def _circ(a, b):
    """Distance of two instantaneous frequencies (units of pi) on the circle of period 2."""
    d = (a - b).abs() % 2.0
    return torch.minimum(d, 2.0 - d)


@pytest.mark.parametrize("n_fft,hop,B,L,mel", [(256, 64, 3, 4000, False), (256, 64, 3, 4000, True),
                                                (2048, 512, 2, 16000, False), (2048, 512, 2, 16000, True),
                                                (512, 128, 1, 1000, True)])
def test_to_spectrogram_against_spec(n_fft, hop, B, L, mel):
    S, cfg, h = _helpers(n_fft, hop, mel)
    x = _audio(B, L, n_fft + L)
    ref = S.to_spectrogram(cfg, x.double(), mel).float()
    got = h.to_spectrogram(x.to(_dev())).cpu()
    assert got.shape == ref.shape == (B, 2, n_fft // 2, -(-L // hop))
    mag_err = (got[:, 0] - ref[:, 0]).abs()
    if mel:   # empty mel filters: log(0 + 1e-6) on both sides
        assert torch.equal(torch.isfinite(got[:, 0]), torch.isfinite(ref[:, 0]))
    assert float(mag_err.max()) <= 1e-3 * float(ref[:, 0].abs().max()), float(mag_err.max())
    d = _circ(got[:, 1, :, 1:], ref[:, 1, :, 1:])
    bad = (d > 2e-3).float().mean()
    assert float(bad) <= 1e-4, f"{float(bad):.2e} of the instantaneous frequencies differ"
    # first frame: the (unwrapped) phase itself, modulo the mel weights' mixing of 2 pi jumps
    d0 = _circ(got[:, 1, :, 0], ref[:, 1, :, 0]) if not mel else (got[:, 1, :, 0] - ref[:, 1, :, 0]).abs()
    assert float((d0 > 2e-3).float().mean()) <= 1e-3


@pytest.mark.parametrize("n_fft,hop,B,T,mel", [(256, 64, 2, 50, False), (256, 64, 2, 50, True),
                                                (2048, 512, 2, 32, True), (2048, 512, 1, 33, False)])
def test_to_audio_against_spec(n_fft, hop, B, T, mel):
    S, cfg, h = _helpers(n_fft, hop, mel)
    x = _audio(B, T * hop - 7, 5 * n_fft + T)
    spec = S.to_spectrogram(cfg, x.double(), mel).float()    # a realistic spectrogram (from the specification)
    ref = S.to_audio(cfg, spec.double(), mel).float()
    got = h.to_audio(spec.to(_dev())).cpu()
    assert got.shape == ref.shape == (B, T * hop)
    err = (got - ref).abs().max() / ref.abs().max()
    assert float(err) <= 1e-3, float(err)


def test_round_trip_and_api():
    S, cfg, h = _helpers(1024, 256, False)
    x = _audio(2, 8192, 3)
    y = h.to_audio(h.to_spectrogram(x.to(_dev()))).cpu()
    inner = slice(1024, 8192 - 1024)
    assert float((y[:, inner] - x[:, inner]).abs().max()) <= 5e-3   # log(|X| + 1e-6) is not exactly invertible
    assert h.fs_hz == 16000
    with pytest.raises(Exception):
        h.to_spectrogram(x)                                          # CPU tensor: no fallback
    with pytest.raises(ValueError):
        h.to_audio(torch.zeros(1, 2, 100, 4, device=_dev()))


def test_normalizer_and_masked_phase_against_spec(tmp_path):
    """DataNormalizer / make_masked_phase_transform (fused affine + mask kernel) against the CPU
    specification, their gradients against autograd, statistics measured from a loader, JSON round trip."""
    from GANsynth_pytorch.loader import make_masked_phase_transform
    from GANsynth_pytorch.normalizer import DataNormalizer, DataNormalizerStatistics
    from oracle import spectrogram_oracle as S
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    batches = [torch.randn(3, 2, 16, 24, generator=g) * 4 - 5 for _ in range(3)]
    st = S.normalizer_statistics(batches)
    norm = DataNormalizer(dataloader=[(b.to(dev), None) for b in batches])
    for k, v in st.items():
        assert abs(norm.statistics[k] - v) <= 1e-6 * max(1.0, abs(v)), k
    x = batches[1]
    y = norm.normalize(x.to(dev))
    ref = S.normalize(x, st)
    assert (y.cpu() - ref).abs().max() <= 1e-6 * ref.abs().max()
    assert y[:, 0].abs().max() <= 0.8 + 1e-5 and y[:, 1].abs().max() <= 1.0 + 1e-5
    back = norm.denormalize(y)
    assert (back.cpu() - x).abs().max() <= 1e-5 * x.abs().max()
    # masked phase, alone and fused with the de-normalisation; 3-D input like a dataset transform sees
    thr = float(x[:, 0].median())
    m = make_masked_phase_transform(thr)(x.to(dev))
    assert torch.equal(m.cpu(), S.mask_phase(x, thr))
    assert torch.equal(make_masked_phase_transform(thr)(x[0].to(dev)).cpu(), S.mask_phase(x[:1], thr)[0])
    fused = norm.denormalize(y, threshold=thr)
    ref_f = S.mask_phase(S.denormalize(ref, st), thr)
    near = (S.denormalize(ref, st)[:, 0] - thr).abs() < 1e-4          # bins whose magnitude sits on the threshold
    assert ((fused.cpu() - ref_f).abs().max(1).values[~near]).max() <= 1e-5 * x.abs().max()
    # gradients
    xd = x.to(dev).requires_grad_(True)
    w = torch.randn(x.shape, generator=g)
    (norm.denormalize(norm.normalize(xd) * 1.5, threshold=thr) * w.to(dev)).sum().backward()
    xr = x.clone().requires_grad_(True)
    (S.mask_phase(S.denormalize(S.normalize(xr, st) * 1.5, st), thr) * w).sum().backward()
    assert (xd.grad.cpu() - xr.grad).abs().max() <= 1e-5 * xr.grad.abs().max()
    # persistence
    norm.dump_statistics(tmp_path / "stats.json")
    again = DataNormalizer.load_statistics(tmp_path / "stats.json")
    assert dict(again.statistics) == dict(norm.statistics)
    assert isinstance(DataNormalizerStatistics(**norm.statistics), dict)


def test_vqvae_with_normalizer_and_output_threshold():
    """VQVAE(normalizer_statistics=..., output_spectrogram_min_magnitude=...) (vqvae.py:218-241,254-255,297-302):
    encode sees the normalised input, every decode path post-processes."""
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from oracle import spectrogram_oracle as S, vqvae_oracle as O
    dev = torch.device("cuda:0")
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16, num_embeddings=64)
    cfg = O.Config(**kw)
    sd = O.init_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 2, 32, 32, generator=g) * 3 - 4
    st = S.normalizer_statistics([x])
    O.calibrate_codebooks(sd, cfg, S.normalize(x, st))
    thr = -0.5
    m = VQVAE(normalizer_statistics=st, output_spectrogram_min_magnitude=thr, **kw)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    ref = O.forward(S.normalize(x, st), sd, cfg)
    ref_dec = S.mask_phase(S.denormalize(ref[0], st), thr)
    dec, diff, p_t, p_b, id_t, id_b = m(x.to(dev))
    assert torch.equal(id_t.cpu(), ref[4]) and torch.equal(id_b.cpu(), ref[5])
    near = (S.denormalize(ref[0], st)[:, 0] - thr).abs() < 1e-3
    err = (dec.cpu() - ref_dec).abs()
    assert err[:, 0].max() <= 1e-4 * ref_dec.abs().max() and err[:, 1][~near].max() <= 1e-4 * ref_dec.abs().max()
    assert (dec[:, 1][dec[:, 0] <= thr] == 0).all() and (dec[:, 0] <= thr).any()
    # forward decodes the straight-through value z + (e - z), decode_code the code vector e: equal up to the rounding
    # of that sum (bottleneck.py:94-101), as in the reference
    again = m.decode_code(id_t, id_b)
    keep = (S.denormalize(ref[0], st)[:, 0] - thr).abs() >= 1e-3
    assert (again[:, 0] - dec[:, 0]).abs().max() <= 2e-6 * dec.abs().max()
    assert (again[:, 1] - dec[:, 1]).abs().cpu()[keep].max() <= 2e-6 * dec.abs().max()
    # the parameters JSON carries the statistics (vqvae.py:98-122)
    assert m.normalizer_statistics == st and m.output_spectrogram_min_magnitude == thr
    # train mode: differentiable through the post-processing
    m.train()
    out, latent, *_ = m(x.to(dev))
    (out.pow(2).mean() + latent.mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("name", ["ddsp", "jukebox", "l2"])
def test_multiscale_spectral_loss_against_reference(golden_dir, name):
    """utils/losses/spectral.py classes (strided-conv STFTs + fused distance kernels) against the values and the
    audio gradients of the reference's own classes (tests/golden/spectral_loss.npz)."""
    from interactive_spectrogram_inpainting.utils.losses import spectral as S
    z = np.load(golden_dir / "spectral_loss.npz")
    dev = torch.device("cuda:0")
    kw = {"ddsp": S.DDSPMultiscaleSpectralLoss_kwargs, "jukebox": S.JukeboxMultiscaleSpectralLoss_kwargs,
          "l2": dict(n_ffts=[256, 512], window_lengths=[200, 512], overlap_ratio=0.75, loss=S.L2Loss(),
                     lin_loss_alpha=0.5, log_loss_alpha=2.0)}[name]
    m = S.MultiscaleSpectralLoss(**kw)
    p = torch.from_numpy(z["pred"]).to(dev).requires_grad_(True)
    loss = m(p, torch.from_numpy(z["target"]).to(dev))
    loss.backward()
    ref_l, ref_g = float(z[f"{name}::loss"]), torch.from_numpy(z[f"{name}::grad"])
    assert abs(loss.item() - ref_l) <= 1e-4 * abs(ref_l), (loss.item(), ref_l)
    err = (p.grad.cpu() - ref_g).abs().max() / ref_g.abs().max()
    assert err <= 1e-3, f"audio gradient: max error / max|ref| = {err:.3e}"


def test_spectral_loss_from_spectrogram_runs():
    from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper
    from interactive_spectrogram_inpainting.utils.losses.spectral import JukeboxMultiscaleSpectralLoss_fromSpectrogram
    from oracle import spectral_loss_oracle as S
    dev = torch.device("cuda:0")
    h = MelSpectrogramsHelper(16000, 2048, 512, 2048).to(dev)
    g = torch.Generator().manual_seed(5)
    a = torch.randn(2, 16000, generator=g).to(dev) * 0.1
    spec_t = h.to_spectrogram(a)
    spec_p = h.to_spectrogram(a * 0.8 + 0.01 * torch.randn(2, 16000, generator=g).to(dev))
    crit = JukeboxMultiscaleSpectralLoss_fromSpectrogram(h)
    loss = crit(spec_p, spec_t)
    ref = S.multiscale_spectral_loss(h.to_audio(spec_p).cpu(), h.to_audio(spec_t).cpu(), [2048, 1024, 512],
                                     [1200, 600, 240], 0.80, "mse", 1.0, 0.0)
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())


@pytest.mark.parametrize("mel", [False, True])
def test_to_audio_gradient_against_spec(mel):
    """Backward of to_audio (framing convolution with the inverse basis, adjoint polar / mel / running-sum kernels)
    against torch autograd through the CPU specification."""
    from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper, SpectrogramsHelper
    from oracle import spectrogram_oracle as S
    dev = torch.device("cuda:0")
    n_fft, hop = 256, 64
    cfg = S.SpecConfig(fs_hz=16000, n_fft=n_fft, hop_length=hop, window_length=n_fft)
    h = (MelSpectrogramsHelper if mel else SpectrogramsHelper)(16000, n_fft, hop, n_fft).to(dev)
    g = torch.Generator().manual_seed(9)
    audio = torch.randn(2, 40 * hop, generator=g) * 0.2
    spec = S.to_spectrogram(cfg, audio, mel).float()          # a realistic operating point
    w = torch.randn(2, 40 * hop, generator=g)
    sd = spec.to(dev).requires_grad_(True)
    out = h.to_audio(sd)
    assert out.requires_grad
    (out * w.to(dev)).sum().backward()
    sr = spec.double().requires_grad_(True)
    ref = S.to_audio(cfg, sr, mel)
    (ref * w.double()).sum().backward()
    assert (out.detach().cpu() - ref.detach().float()).abs().max() <= 2e-3 * ref.abs().max()
    gr = sr.grad.float()
    for ch in (0, 1):
        err = (sd.grad[:, ch].cpu() - gr[:, ch]).abs().max() / gr[:, ch].abs().max()
        assert err <= 5e-3, f"channel {ch}: gradient error {err:.3e}"


def test_spectral_loss_from_spectrogram_trains():
    """A `_fromSpectrogram` criterion back-propagates into the predicted spectrogram (and only into it)."""
    from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper
    from interactive_spectrogram_inpainting.utils.losses.spectral import DDSPMultiscaleSpectralLoss_fromSpectrogram
    dev = torch.device("cuda:0")
    h = MelSpectrogramsHelper(16000, 2048, 512, 2048).to(dev)
    g = torch.Generator().manual_seed(6)
    a = torch.randn(2, 16000, generator=g).to(dev) * 0.1
    target = h.to_spectrogram(a)
    pred = (target + 0.05 * torch.randn(target.shape, generator=g).to(dev)).requires_grad_(True)
    crit = DDSPMultiscaleSpectralLoss_fromSpectrogram(h)
    l0 = crit(pred, target)
    l0.backward()
    assert pred.grad is not None and torch.isfinite(pred.grad).all() and pred.grad.abs().max() > 0
    with torch.no_grad():
        stepped = pred - 1e-2 * pred.grad / pred.grad.abs().max()
    assert crit(stepped, target).item() < l0.item()         # a small step against the gradient lowers the loss
