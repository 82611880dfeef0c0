"""CPU restatement of the reference's multi-scale spectral loss (utils/losses/spectral.py:78-104) -- TEST
INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).  Pinned by tests/golden/spectral_loss.npz, which holds
values and gradients of the reference classes themselves (oracle/make_golden.py::spectral_loss_fixtures; the
reference's `torch.stft(...)` call predates `return_complex` and is run there with the legacy real-pair output
torch still offers)."""
import math

import torch


def multiscale_spectral_loss(audio_pred, audio_target, n_ffts, window_lengths=None, overlap_ratio=0.75, kind="l1",
                             lin_loss_alpha=1.0, log_loss_alpha=1.0, eps=1e-6):
    window_lengths = window_lengths or n_ffts
    lin, log = [], []

    def crit(a, b):
        if kind == "l1":
            return (a - b).abs().mean()
        if kind == "mse":
            return (a - b).pow(2).mean()
        return (b - a).reshape(a.shape[0], -1).norm(2, dim=-1)          # L2Loss, reference :136-143

    for n_fft, win in zip(n_ffts, window_lengths):
        hop = math.ceil((1 - overlap_ratio) * win)                      # :84
        w = torch.hann_window(win, dtype=audio_pred.dtype)
        mp, mt = (torch.stft(a, n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=False,
                             return_complex=True).abs() for a in (audio_pred, audio_target))
        if lin_loss_alpha > 0:
            lin.append(crit(mp, mt))
        if log_loss_alpha > 0:
            log.append(crit(torch.log(mp + eps), torch.log(mt + eps)))
    mean = lambda ts: sum(ts) / len(ts) if ts else 0
    return (lin_loss_alpha * mean(lin) + log_loss_alpha * mean(log)).mean()
