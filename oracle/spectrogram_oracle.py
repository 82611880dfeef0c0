"""TEST INFRASTRUCTURE — CPU specification of the audio <-> spectrogram front-end.

PARITY UNPINNED.  The reference delegates this to the absent, un-pinned package
`GANsynth_pytorch` (`spectrograms_helper.{SpectrogramsHelper, MelSpectrogramsHelper}`;
call sites utils/misc.py:10-29, train_vqvae.py:61-79,392-400, sample.py:488-599,
flask_server.py:242-244,596,1016).  That package is a PyTorch port of the GANSynth
representation (Engel et al., "GANSynth: Adversarial Neural Audio Synthesis", ICLR 2019,
section 2.2 / appendix; Magenta's `specgrams_helper`), whose published algorithm is
restated here with torch's FFT as an independent reference for the GEMM-based HIP path:

  audio -> STFT (periodic Hann window of n_fft, hop, left pad n_fft - hop, DC bin dropped)
        -> log-magnitude  log(|X| + 1e-6)  and instantaneous frequency
           IF[t] = wrap(angle[t] - angle[t-1]) / pi   (IF[0] = angle[0] / pi)
  mel variant: power and UNWRAPPED phase are both projected with the triangular
           mel matrix (as many mel bins as linear bins, mel = 1127 ln(1 + f / 700));
           channel 0 = log(mel power + 1e-6), channel 1 = IF of the projected phase
  inverse: approximate inverse mel matrix  M^T diag(1 / colsum(M M^T)),  polar -> iSTFT
           with the synthesis window hann / sum_k hann^2(n + k hop)  (overlap-add).

The SonyCSL port's extra `mel_bin_width_threshold_factor` ("expand resolution") option
has no published definition and is not reproduced.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

EPS = 1e-6


@dataclass
class SpecConfig:
    fs_hz: int = 16000
    n_fft: int = 2048
    hop_length: int = 512
    window_length: int = 2048
    lower_edge_hertz: float = 0.0
    upper_edge_hertz: float = 8000.0
    mel_break_frequency_hertz: float = 700.0

    @property
    def n_bins(self) -> int:
        return self.n_fft // 2


def num_frames(cfg: SpecConfig, length: int) -> int:
    return max(1, -(-length // cfg.hop_length))


def pad_audio(cfg: SpecConfig, audio: torch.Tensor) -> torch.Tensor:
    """Left pad n_fft - hop, right pad so that ceil(L / hop) full frames exist."""
    L = audio.shape[-1]
    T = num_frames(cfg, L)
    total = (T - 1) * cfg.hop_length + cfg.n_fft
    left = cfg.n_fft - cfg.hop_length
    return torch.nn.functional.pad(audio, (left, total - left - L))


def hertz_to_mel(f, break_hz: float):
    q = 1127.0  # _MEL_HIGH_FREQUENCY_Q
    return q * torch.log1p(torch.as_tensor(f, dtype=torch.float64) / break_hz)


def mel_matrix(cfg: SpecConfig) -> torch.Tensor:
    """[n_bins linear, n_bins mel] triangular filters in the mel domain (float64)."""
    n = cfg.n_bins
    nyquist = cfg.fs_hz / 2.0
    # the spectrogram has lost its DC bin: bin i is frequency (i + 1) * fs / n_fft
    lin = torch.linspace(0.0, nyquist, n + 1, dtype=torch.float64)[1:]
    spec_mel = hertz_to_mel(lin, cfg.mel_break_frequency_hertz)[:, None]
    edges = torch.linspace(float(hertz_to_mel(cfg.lower_edge_hertz, cfg.mel_break_frequency_hertz)),
                           float(hertz_to_mel(cfg.upper_edge_hertz, cfg.mel_break_frequency_hertz)),
                           n + 2, dtype=torch.float64)
    lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    up = (spec_mel - lower) / (center - lower)
    down = (upper - spec_mel) / (upper - center)
    return torch.clamp(torch.minimum(up, down), min=0.0)


def inverse_mel_matrix(m: torch.Tensor) -> torch.Tensor:
    """[n mel, n linear]: M^T scaled so that the columns of M M^T sum to one."""
    mt = m.t()
    p = m @ mt
    s = p.sum(0)
    d = torch.where(s.abs() > 1e-8, 1.0 / s, s)
    return mt * d[None, :]


def wrap(d: torch.Tensor) -> torch.Tensor:
    """Principal value in [-pi, pi] with the sign convention of numpy.unwrap."""
    w = torch.remainder(d + math.pi, 2 * math.pi) - math.pi
    return torch.where((w == -math.pi) & (d > 0), torch.full_like(w, math.pi), w)


def instantaneous_frequency(phase: torch.Tensor) -> torch.Tensor:
    """phase [..., T] (time last) -> wrapped finite difference / pi, first frame kept."""
    d = wrap(phase[..., 1:] - phase[..., :-1])
    return torch.cat([phase[..., :1], d], -1) / math.pi


def unwrap(phase: torch.Tensor) -> torch.Tensor:
    d = wrap(phase[..., 1:] - phase[..., :-1])
    return torch.cat([phase[..., :1], phase[..., :1] + torch.cumsum(d, -1)], -1)


def stft(cfg: SpecConfig, audio: torch.Tensor) -> torch.Tensor:
    """[B, L] -> complex [B, n_bins, T] (DC dropped)."""
    x = pad_audio(cfg, audio)
    w = torch.hann_window(cfg.window_length, periodic=True, dtype=audio.dtype)
    X = torch.stft(x, cfg.n_fft, hop_length=cfg.hop_length, win_length=cfg.window_length, window=w, center=False,
                   onesided=True, return_complex=True)
    return X[:, 1:, :]


def to_spectrogram(cfg: SpecConfig, audio: torch.Tensor, mel: bool) -> torch.Tensor:
    """[B, L] -> [B, 2, n_bins, T]."""
    X = stft(cfg, audio)
    mag, ang = X.abs(), torch.angle(X)
    if not mel:
        return torch.stack([torch.log(mag + EPS), instantaneous_frequency(ang)], 1)
    M = mel_matrix(cfg).to(audio.dtype)
    power = mag * mag
    ph = unwrap(ang)
    mel_power = torch.einsum("bft,fm->bmt", power, M)
    mel_ph = torch.einsum("bft,fm->bmt", ph, M)
    return torch.stack([torch.log(mel_power + EPS), instantaneous_frequency(mel_ph)], 1)


def synthesis_window(cfg: SpecConfig, dtype=torch.float32) -> torch.Tensor:
    w = torch.hann_window(cfg.window_length, periodic=True, dtype=torch.float64)
    den = torch.zeros_like(w)
    hop, N = cfg.hop_length, cfg.window_length
    for k in range(-(N // hop), N // hop + 1):
        lo, hi = max(0, -k * hop), min(N, N - k * hop)
        if lo < hi:
            den[lo:hi] += w[lo + k * hop:hi + k * hop] ** 2
    return (w / den).to(dtype)


def to_audio(cfg: SpecConfig, spec: torch.Tensor, mel: bool) -> torch.Tensor:
    """[B, 2, n_bins, T] -> [B, T * hop]."""
    a, p = spec[:, 0], spec[:, 1]
    ph = torch.cumsum(p * math.pi, -1)
    if mel:
        Minv = inverse_mel_matrix(mel_matrix(cfg)).to(spec.dtype)
        power = torch.einsum("bmt,mf->bft", torch.exp(a), Minv)
        ph = torch.einsum("bmt,mf->bft", ph, Minv)
        mag = torch.exp(0.5 * torch.log(power.clamp_min(0) + EPS))
    else:
        mag = torch.exp(a)
    X = torch.polar(mag, ph)
    X = torch.cat([torch.zeros_like(X[:, :1]), X], 1)               # DC bin back
    B, _, T = X.shape
    frames = torch.fft.irfft(X, n=cfg.n_fft, dim=1)                  # [B, n_fft, T]
    frames = frames * synthesis_window(cfg, spec.dtype)[None, :, None]
    total = (T - 1) * cfg.hop_length + cfg.n_fft
    out = torch.zeros(B, total, dtype=spec.dtype)
    for t in range(T):
        out[:, t * cfg.hop_length:t * cfg.hop_length + cfg.n_fft] += frames[:, :, t]
    left = cfg.n_fft - cfg.hop_length
    return out[:, left:left + T * cfg.hop_length]


# ---------------------------------------------------------------- range normalisation / masked phase
# (GANsynth_pytorch.normalizer.DataNormalizer, GANsynth_pytorch.loader.make_masked_phase_transform; call
# sites vqvae.py:221-241,254-255,297-302, train_vqvae.py:646-676.  PARITY UNPINNED: source absent.)
def normalizer_statistics(specs, magnitude_margin: float = 0.8, IF_margin: float = 1.0) -> dict:
    """Affine maps sending each channel's [min, max] over `specs` (iterable of [B,2,F,T]) to [-margin, margin]."""
    lo = [min(float(s[:, c].min()) for s in specs) for c in range(2)]
    hi = [max(float(s[:, c].max()) for s in specs) for c in range(2)]
    out = {}
    for c, (ka, kb, margin) in enumerate((("s_a", "s_b", magnitude_margin), ("p_a", "p_b", IF_margin))):
        rng = hi[c] - lo[c]
        out[ka] = margin * 2.0 / rng
        out[kb] = margin * (1.0 - 2.0 * hi[c] / rng)
    return out


def normalize(spec: torch.Tensor, st: dict) -> torch.Tensor:
    a = torch.tensor([st["s_a"], st["p_a"]], dtype=spec.dtype).view(1, 2, 1, 1)
    b = torch.tensor([st["s_b"], st["p_b"]], dtype=spec.dtype).view(1, 2, 1, 1)
    return spec * a + b


def denormalize(spec: torch.Tensor, st: dict) -> torch.Tensor:
    a = torch.tensor([1.0 / st["s_a"], 1.0 / st["p_a"]], dtype=spec.dtype).view(1, 2, 1, 1)
    b = torch.tensor([-st["s_b"] / st["s_a"], -st["p_b"] / st["p_a"]], dtype=spec.dtype).view(1, 2, 1, 1)
    return spec * a + b


def mask_phase(spec: torch.Tensor, min_magnitude: float) -> torch.Tensor:
    """Instantaneous frequency zeroed where the log-magnitude is at or below `min_magnitude`."""
    out = spec.clone()
    out[:, 1] = torch.where(spec[:, 0] <= min_magnitude, torch.zeros_like(spec[:, 1]), spec[:, 1])
    return out
