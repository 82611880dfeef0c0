"""Audio <-> (log-magnitude, instantaneous-frequency) spectrograms on MI355X.

Same class names, constructor keywords and methods the reference uses
(utils/misc.py:10-29, train_vqvae.py:61-79,392-400, sample.py:488-599,
flask_server.py:242-244,596,1016):

    SpectrogramsHelper(fs_hz, n_fft, hop_length, window_length)
    MelSpectrogramsHelper(..., lower_edge_hertz, upper_edge_hertz,
                          mel_break_frequency_hertz, mel_bin_width_threshold_factor)
        .to(device)  .to_spectrogram(audio [B, L]) -> [B, 2, n_fft/2, T]
        .to_audio(spec [B, 2, n_fft/2, T]) -> [B, T * hop]        .fs_hz

Arithmetic = the published GANSynth representation (specification and parity status:
oracle/spectrogram_oracle.py — the original package is absent, parity unpinned).
Compute: the windowed DFT is a convolution of the audio, viewed as [B, 1, L/hop, hop]
channels-last, with the DFT basis as a 1 x (n_fft/hop) kernel on the exact-fp32 matrix
pipe (isi_conv2d_f32); mel projections are 1x1 convolutions; polar / unwrap / integrate /
transpose stages and the overlap-add are the HBM-bound kernels of csrc/spectrogram.hip.
No CPU path: tensors must live on the GPU.  `from_wavfile` needs an audio decoder
(torchaudio / soundfile, not in this image) and is not built.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from interactive_spectrogram_inpainting import _hip
from interactive_spectrogram_inpainting.priors import _ops as _gemm
from interactive_spectrogram_inpainting.vqvae._ops import pack_conv_weight

_MEL_BREAK_FREQUENCY_HERTZ = 700.0
_MEL_HIGH_FREQUENCY_Q = 1127.0


def _s(t):
    return C.c_void_p(_hip.stream_ptr(t.device))


class SpectrogramsHelper:
    mel = False

    def __init__(self, fs_hz: int = 16000, n_fft: int = 2048, hop_length: int = 512, window_length: int = 2048,
                 device: Optional[torch.device] = None):
        if window_length != n_fft:
            raise NotImplementedError("window_length != n_fft is not used by the reference's configurations")
        if n_fft % hop_length or hop_length % 4:
            raise NotImplementedError("hop_length must divide n_fft and be a multiple of 4")
        self.fs_hz, self.n_fft, self.hop_length, self.window_length = fs_hz, n_fft, hop_length, window_length
        self.n_bins = n_fft // 2
        # floor of the log-magnitude channel, log(0 + eps): the threshold the masked-phase transform and
        # `output_spectrogram_min_magnitude` use (train_vqvae.py:710-712)
        self.safelog_eps = math.log(1e-6)
        self.device = torch.device(device) if device is not None else None
        self._built = None

    # ------------------------------------------------------------------ constants (host, float64 -> fp32)
    def _dft_bases(self):
        N, F = self.n_fft, self.n_bins
        n = torch.arange(N, dtype=torch.float64)
        k = torch.arange(1, F + 1, dtype=torch.float64)                     # DC bin dropped
        w = torch.hann_window(N, periodic=True, dtype=torch.float64)
        arg = 2.0 * math.pi * k[:, None] * n[None, :] / N
        fwd = torch.cat([torch.cos(arg), -torch.sin(arg)], 0) * w[None, :]   # [2F, N]: X = sum_n x[n] w[n] e^{-i arg}
        # inverse real DFT of a spectrum without DC, times the overlap-add synthesis window
        den = torch.zeros(N, dtype=torch.float64)
        hop = self.hop_length
        for j in range(-(N // hop), N // hop + 1):
            lo, hi = max(0, -j * hop), min(N, N - j * hop)
            if lo < hi:
                den[lo:hi] += w[lo + j * hop:hi + j * hop] ** 2
        ws = w / den
        scale = torch.full((F,), 2.0 / N, dtype=torch.float64)
        scale[-1] = 1.0 / N                                                  # Nyquist bin counts once
        inv = torch.cat([torch.cos(arg) * scale[:, None], -torch.sin(arg) * scale[:, None]], 0) * ws[None, :]  # [2F, N]
        return fwd.float(), inv.float()

    def _build(self, device):
        if self._built is not None and self._built["device"] == device:
            return self._built
        fwd, inv = self._dft_bases()
        N, F, hop = self.n_fft, self.n_bins, self.hop_length
        kw = N // hop
        # conv weight [Cout = 2F, Cin = hop, 1, kw]: tap j, channel c = sample j * hop + c of the frame
        w_conv = fwd.reshape(2 * F, kw, hop).permute(0, 2, 1).reshape(2 * F, hop, 1, kw).contiguous().to(device)
        wi_conv = inv.reshape(2 * F, kw, hop).permute(0, 2, 1).reshape(2 * F, hop, 1, kw).contiguous().to(device)
        built = {"device": device, "stft_w": pack_conv_weight(w_conv),
                 "istft_w": _gemm.pack_linear_weight(inv.t().contiguous().to(device)),   # Linear [N out, 2F in]
                 "istft_conv_w": pack_conv_weight(wi_conv)}   # its adjoint, as the framing convolution (backward)
        built.update(self._build_extra(device))
        self._built = built
        return built

    def _build_extra(self, device):
        return {}

    def to(self, device):
        self.device = torch.device(device)
        return self

    # ------------------------------------------------------------------ forward
    def _stft(self, audio: torch.Tensor):
        """[B, L] -> (X [B, T, 2F] re|im, T)."""
        _hip.require_gpu(audio, "audio")
        c = self._build(audio.device)
        B, L = audio.shape
        N, hop, F = self.n_fft, self.hop_length, self.n_bins
        T = max(1, -(-L // hop))
        total = (T - 1) * hop + N
        left = N - hop
        x = torch.nn.functional.pad(audio.float(), (left, total - left - L)).contiguous()
        out = torch.empty(B, T, 2 * F, dtype=torch.float32, device=audio.device)
        s0 = _hip.isi_src(x.data_ptr(), hop, total, 1, total, hop)
        dst = _hip.isi_dst(out.data_ptr(), T * 2 * F, 1, T * 2 * F, 2 * F)
        rc = _hip.lib().isi_conv2d_f32(C.byref(s0), None, c["stft_w"].data_ptr(), None, None, C.byref(dst),
                                       B, 1, total // hop, 2 * F, 1, N // hop, 1, 0, _gemm._PREC_FLAG[_gemm.LINEAR_PRECISION],
                                       _s(audio))
        _hip.check(rc, "isi_conv2d_f32 (stft)")
        return out, T

    def to_spectrogram(self, audio: torch.Tensor) -> torch.Tensor:
        X, T = self._stft(audio)
        B, F = X.shape[0], self.n_bins
        L = _hip.lib()
        a = torch.empty(B, T, F, dtype=torch.float32, device=X.device)
        ph = torch.empty_like(a)
        _hip.check(L.isi_spec_polar_f32(X.data_ptr(), a.data_ptr(), ph.data_ptr(), B, T, F, int(self.mel), _s(X)),
                   "isi_spec_polar_f32")
        a, ph = self._project(a, ph)
        out = torch.empty(B, 2, F, T, dtype=torch.float32, device=X.device)
        _hip.check(L.isi_spec_finish_f32(a.data_ptr(), ph.data_ptr(), out.data_ptr(), B, T, F, int(self.mel), _s(X)),
                   "isi_spec_finish_f32")
        return out

    def _project(self, a, ph):
        return a, ph

    def _unproject(self, a, ph):
        return a, ph

    def _unproject_bwd(self, da, dph):
        return da, dph

    # ------------------------------------------------------------------ inverse
    def to_audio(self, spec: torch.Tensor) -> torch.Tensor:
        """[B, 2, F, T] -> [B, T * hop]; differentiable w.r.t. the spectrogram (the `_fromSpectrogram` losses of
        utils/losses/spectral.py train through it)."""
        if torch.is_grad_enabled() and spec.requires_grad:
            return _ToAudioFunction.apply(spec, self)
        return self._to_audio(spec)[0]

    def _to_audio(self, spec: torch.Tensor, keep: bool = False):
        _hip.require_gpu(spec, "spectrogram")
        c = self._build(spec.device)
        spec = spec.float().contiguous()
        B, two, F, T = spec.shape
        if two != 2 or F != self.n_bins:
            raise ValueError(f"expected [B, 2, {self.n_bins}, T], got {tuple(spec.shape)}")
        L = _hip.lib()
        a = torch.empty(B, T, F, dtype=torch.float32, device=spec.device)
        ph = torch.empty_like(a)
        _hip.check(L.isi_spec_inverse_prepare_f32(spec.data_ptr(), a.data_ptr(), ph.data_ptr(), B, T, F, _s(spec)),
                   "isi_spec_inverse_prepare_f32")
        a, ph = self._unproject(a, ph)
        X = torch.empty(B, T, 2 * F, dtype=torch.float32, device=spec.device)
        _hip.check(L.isi_spec_to_stft_f32(a.data_ptr(), ph.data_ptr(), X.data_ptr(), B * T, F, int(self.mel), _s(spec)),
                   "isi_spec_to_stft_f32")
        frames = _gemm.linear(X, c["istft_w"], None, self.n_fft)            # [B, T, n_fft], synthesis window folded in
        n_out = T * self.hop_length
        audio = torch.empty(B, n_out, dtype=torch.float32, device=spec.device)
        _hip.check(L.isi_overlap_add_f32(frames.data_ptr(), audio.data_ptr(), B, T, self.n_fft, self.hop_length,
                                         self.n_fft - self.hop_length, n_out, _s(spec)), "isi_overlap_add_f32")
        return audio, ((spec, a, ph) if keep else None)

    def _to_audio_backward(self, saved, d_audio: torch.Tensor) -> torch.Tensor:
        spec, a, ph = saved
        c = self._build(spec.device)
        B, _, F, T = spec.shape
        N, hop = self.n_fft, self.hop_length
        L = _hip.lib()
        # d frames[b, t, k] = d audio[b, t hop + k - left]; d X = d frames . inv^T : the framing convolution of the
        # padded gradient with the inverse basis (same form as the forward STFT)
        left, total = N - hop, (T - 1) * hop + N
        g = torch.nn.functional.pad(d_audio.float(), (left, total - left - T * hop)).contiguous()
        dX = torch.empty(B, T, 2 * F, dtype=torch.float32, device=spec.device)
        s0 = _hip.isi_src(g.data_ptr(), hop, total, 1, total, hop)
        dst = _hip.isi_dst(dX.data_ptr(), T * 2 * F, 1, T * 2 * F, 2 * F)
        _hip.check(L.isi_conv2d_f32(C.byref(s0), None, c["istft_conv_w"].data_ptr(), None, None, C.byref(dst),
                                    B, 1, total // hop, 2 * F, 1, N // hop, 1, 0,
                                    _gemm._PREC_FLAG[_gemm.LINEAR_PRECISION], _s(spec)), "isi_conv2d_f32 (istft backward)")
        da, dph = torch.empty_like(a), torch.empty_like(ph)
        _hip.check(L.isi_spec_to_stft_bwd_f32(a.data_ptr(), ph.data_ptr(), dX.data_ptr(), da.data_ptr(), dph.data_ptr(),
                                              B * T, F, int(self.mel), _s(spec)), "isi_spec_to_stft_bwd_f32")
        da, dph = self._unproject_bwd(da, dph)
        dspec = torch.empty_like(spec)
        _hip.check(L.isi_spec_inverse_prepare_bwd_f32(spec.data_ptr(), da.contiguous().data_ptr(),
                                                      dph.contiguous().data_ptr(), dspec.data_ptr(), B, T, F, _s(spec)),
                   "isi_spec_inverse_prepare_bwd_f32")
        return dspec

    def from_wavfile(self, *args, **kwargs):
        raise NotImplementedError("decoding audio files needs torchaudio / soundfile, which this image lacks")


class _ToAudioFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, helper):
        audio, saved = helper._to_audio(spec, keep=True)
        ctx.helper = helper
        ctx.save_for_backward(*saved)
        return audio

    @staticmethod
    def backward(ctx, d_audio):
        return ctx.helper._to_audio_backward(ctx.saved_tensors, d_audio.contiguous()), None


class MelSpectrogramsHelper(SpectrogramsHelper):
    mel = True

    def __init__(self, *args, lower_edge_hertz: float = 0.0, upper_edge_hertz: float = 8000.0,
                 mel_break_frequency_hertz: float = _MEL_BREAK_FREQUENCY_HERTZ,
                 mel_bin_width_threshold_factor: Optional[float] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lower_edge_hertz, self.upper_edge_hertz = lower_edge_hertz, upper_edge_hertz
        self.mel_break_frequency_hertz = mel_break_frequency_hertz
        # the absent package's "expand resolution" option has no published definition: accepted, not applied
        self.mel_bin_width_threshold_factor = mel_bin_width_threshold_factor

    def _mel_matrix(self) -> torch.Tensor:
        n, nyq, brk = self.n_bins, self.fs_hz / 2.0, self.mel_break_frequency_hertz
        to_mel = lambda f: _MEL_HIGH_FREQUENCY_Q * torch.log1p(torch.as_tensor(f, dtype=torch.float64) / brk)  # noqa: E731
        spec_mel = to_mel(torch.linspace(0.0, nyq, n + 1, dtype=torch.float64)[1:])[:, None]
        edges = torch.linspace(float(to_mel(self.lower_edge_hertz)), float(to_mel(self.upper_edge_hertz)), n + 2,
                               dtype=torch.float64)
        lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
        return torch.clamp(torch.minimum((spec_mel - lower) / (center - lower), (upper - spec_mel) / (upper - center)),
                           min=0.0)

    def _build_extra(self, device):
        m = self._mel_matrix()                       # [linear, mel]
        mt = m.t()
        s = (m @ mt).sum(0)
        minv = mt * torch.where(s.abs() > 1e-8, 1.0 / s, s)[None, :]   # [mel, linear]
        return {"mel_w": _gemm.pack_linear_weight(m.t().contiguous().float().to(device)),      # Linear [mel out, linear in]
                "mel_inv_w": _gemm.pack_linear_weight(minv.t().contiguous().float().to(device)),
                "mel_inv_wT": _gemm.pack_linear_weight(minv.contiguous().float().to(device))}     # adjoint (backward)

    def _project(self, a, ph):
        c = self._build(a.device)
        return _gemm.linear(a, c["mel_w"], None, self.n_bins), _gemm.linear(ph, c["mel_w"], None, self.n_bins)

    def _unproject(self, a, ph):
        c = self._build(a.device)
        return _gemm.linear(a, c["mel_inv_w"], None, self.n_bins), _gemm.linear(ph, c["mel_inv_w"], None, self.n_bins)

    def _unproject_bwd(self, da, dph):
        c = self._build(da.device)
        return _gemm.linear(da, c["mel_inv_wT"], None, self.n_bins), _gemm.linear(dph, c["mel_inv_wT"], None, self.n_bins)
