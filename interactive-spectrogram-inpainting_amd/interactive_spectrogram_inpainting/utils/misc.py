"""`get_spectrograms_helper` / `expand_path` of the reference (utils/misc.py:10-33)."""
from __future__ import annotations

import pathlib
from typing import Union

from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper, SpectrogramsHelper


def get_spectrograms_helper(**kwargs) -> SpectrogramsHelper:
    """Helper built from the VQ-VAE training parameters (the keys of the reference's
    command-line JSON: fs_hz, n_fft, hop_length, window_length, use_mel_scale, mel_scale_*)."""
    spectrogram_parameters = {k: kwargs[k] for k in ('fs_hz', 'n_fft', 'hop_length', 'window_length')}
    if kwargs['use_mel_scale']:
        return MelSpectrogramsHelper(
            **spectrogram_parameters,
            lower_edge_hertz=kwargs['mel_scale_lower_edge_hertz'],
            upper_edge_hertz=kwargs['mel_scale_upper_edge_hertz'],
            mel_break_frequency_hertz=kwargs['mel_scale_break_frequency_hertz'],
            mel_bin_width_threshold_factor=kwargs['mel_scale_expand_resolution_factor'])
    return SpectrogramsHelper(**spectrogram_parameters)


def expand_path(p: Union[str, pathlib.Path]) -> pathlib.Path:
    return pathlib.Path(p).expanduser().absolute()
