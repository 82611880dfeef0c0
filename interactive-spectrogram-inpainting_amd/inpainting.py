"""Interactive inpainting operations on MI355X: the compute behind the reference's Flask
routes (flask_server.py), free of HTTP so that they can be called, tested and timed directly.

  make_time_indexes   positional-embedding time indexes of a model window inside a longer
                      codemap (flask_server.py:670-682)
  timerange_change    regenerate the masked zone of the top (and then bottom) or of the bottom
                      codemap inside the window starting at `start_index_top`
                      (flask_server.py:685-870)
  erase               attenuate the log-magnitude under the mask and re-encode (:873-931)
  generate            new top + bottom codemaps from scratch (:376-443)
  codes_to_audio      VQ-VAE decode + spectrogram inversion (:1003-1021)

The heavy work is `sample.sample_model` (encoder pass + KV-cached native decoding loop),
`VQVAE.decode_code / encode` and `SpectrogramsHelper.to_audio`: all HIP kernels.
"""
from __future__ import annotations

from typing import List, Mapping, Optional, Tuple

import torch

from sample import sample_model


def make_time_indexes(start_index: int, codemap_duration: int, transformer_duration: int) -> List[int]:
    """Index 0 marks the attack, index `transformer_duration - 1` the release; the frames in between
    share the remaining `transformer_duration - 2` indexes in equal runs (the last index takes the
    remainder).  Returns the slice seen by a window starting at `start_index`."""
    inner = transformer_duration - 2           # indexes 1 .. inner available for the sustain
    frames = codemap_duration - 2              # frames to label with them
    run = frames // inner
    full = [0]
    for i in range(1, inner):
        full += [i] * run
    full += [inner] * (frames - (len(full) - 1))
    full.append(transformer_duration - 1)
    return full[start_index:start_index + transformer_duration]


def _windows(top_code, bottom_code, transformer_top, transformer_bottom, start_index_top):
    ratio_t = transformer_bottom.shape[1] // transformer_top.shape[1]
    s_top, e_top = start_index_top, start_index_top + transformer_top.shape[1]
    s_bot = ratio_t * start_index_top
    e_bot = s_bot + transformer_bottom.shape[1]
    return (s_top, e_top), (s_bot, e_bot), ratio_t


@torch.no_grad()
def timerange_change(transformer_top, transformer_bottom, top_code: torch.Tensor, bottom_code: torch.Tensor,
                     mask: torch.Tensor, layer: str, start_index_top: int, temperature: float,
                     class_conditioning_top: Mapping[str, torch.Tensor],
                     class_conditioning_bottom: Mapping[str, torch.Tensor], device,
                     uniform_sampling: bool = False, generator: Optional[torch.Generator] = None,
                     **sampling_kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
    """top_code [1,F_t,T], bottom_code [1,F_b,T_b] (T may exceed the models' duration), mask bool
    [1,F,W] in the resolution of `layer` over the model window.  Returns the updated (top, bottom)."""
    (s_top, e_top), (s_bot, e_bot), ratio_t = _windows(top_code, bottom_code, transformer_top, transformer_bottom,
                                                        start_index_top)
    top_frame = top_code[..., s_top:e_top]
    bottom_frame = bottom_code[..., s_bot:e_bot]
    mask = mask.to(device)
    ti_top = make_time_indexes(s_top, top_code.shape[-1], transformer_top.shape[-1])
    ti_bottom = make_time_indexes(s_bot, bottom_code.shape[-1], transformer_bottom.shape[-1])
    common = dict(device=device, batch_size=1, temperature=temperature, generator=generator, **sampling_kwargs)

    def resample(model, condition, initial, m, cls, ti_src, ti_tgt):
        if uniform_sampling:
            rnd = torch.randint(0, model.n_class_target, initial.shape, generator=generator).to(initial.device)
            return torch.where(m, rnd, initial)
        return sample_model(model=model, condition=condition, codemap_size=model.shape, class_conditioning=cls,
                            initial_code=initial, mask=m, time_indexes_source=ti_src, time_indexes_target=ti_tgt,
                            **common)

    top_code, bottom_code = top_code.clone(), bottom_code.clone()
    if layer == 'bottom':
        bottom_code[..., s_bot:e_bot] = resample(transformer_bottom, top_frame, bottom_frame, mask,
                                                 class_conditioning_bottom, ti_top, ti_bottom)
    elif layer == 'top':
        condition = top_frame if transformer_top.self_conditional_model else None
        new_top = resample(transformer_top, condition, top_frame, mask, class_conditioning_top, ti_top, ti_top)
        top_code[..., s_top:e_top] = new_top
        ratio_f = transformer_bottom.shape[0] // transformer_top.shape[0]
        mask_bottom = mask.repeat_interleave(ratio_f, -2).repeat_interleave(ratio_t, -1)
        bottom_code[..., s_bot:e_bot] = resample(transformer_bottom, new_top, bottom_frame, mask_bottom,
                                                 class_conditioning_bottom, ti_top, ti_bottom)
    else:
        raise ValueError(f"unknown layer {layer}")
    return top_code, bottom_code


@torch.no_grad()
def generate(transformer_top, transformer_bottom, temperature: float,
             class_conditioning_top: Mapping[str, torch.Tensor],
             class_conditioning_bottom: Mapping[str, torch.Tensor], device,
             generator: Optional[torch.Generator] = None, **sampling_kwargs):
    top = sample_model(model=transformer_top, device=device, batch_size=1, codemap_size=transformer_top.shape,
                       temperature=temperature, class_conditioning=class_conditioning_top, generator=generator,
                       **sampling_kwargs)
    bottom = sample_model(model=transformer_bottom, device=device, condition=top, batch_size=1,
                          codemap_size=transformer_bottom.shape, temperature=temperature,
                          class_conditioning=class_conditioning_bottom, generator=generator, **sampling_kwargs)
    return top, bottom


@torch.no_grad()
def erase(vqvae, top_code: torch.Tensor, bottom_code: torch.Tensor, mask: torch.Tensor, amplitude: float,
          start_index_top: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Subtract 200 * amplitude from the decoded log-magnitude under the (up-sampled, placed) mask and
    re-encode; mask bool [F_t, W] in top resolution."""
    spec = vqvae.decode_code(top_code, bottom_code)[0]
    logmag, IF = spec[0], spec[1]
    top = top_code[0]
    up_f, up_t = logmag.shape[0] // top.shape[0], logmag.shape[1] // top.shape[1]
    up = mask.to(logmag.device).float().flip(0).repeat_interleave(up_f, 0).repeat_interleave(up_t, 1).flip(0)
    amp = 200.0 * amplitude * up
    before = torch.zeros(logmag.shape[0], up_t * start_index_top, device=logmag.device)
    after = torch.zeros(logmag.shape[0], max(0, up_t * (top.shape[1] - (start_index_top + mask.shape[1]))),
                        device=logmag.device)
    amp = torch.cat([before, amp, after], 1)
    x = torch.stack([logmag - amp, IF], 0).unsqueeze(0).contiguous()
    _, _, _, new_top, new_bottom, *_ = vqvae.encode(x)
    return new_top, new_bottom


@torch.no_grad()
def codes_to_audio(vqvae, spectrograms_helper, top_code: torch.Tensor, bottom_code: torch.Tensor) -> torch.Tensor:
    return spectrograms_helper.to_audio(vqvae.decode_code(top_code, bottom_code))


@torch.no_grad()
def top_conditioned_sample(vqvae, transformer_bottom, spectrograms_helper, top_code: torch.Tensor, temperature: float,
                           class_conditioning_bottom, device, top_k_sampling_k: int = 0, top_p_sampling_p: float = 0.0,
                           generator=None, uniforms=None):
    """The compute of `/top-conditioned-sample` (flask_server.py:1049-1110): ONE top codemap, a batch of bottom codemaps
    drawn from the bottom prior -- one per entry of the per-row class conditioning (the route passes a pitch range:
    `make_conditioning_tensors({'pitch': (lo, hi), ...})`, sample.py:68-101) --, decoded and turned into audio.
    The batch goes through the native batched decoder in one call (rows of a stage as one launch / matrix tiles beyond 16
    sequences).  Returns (bottom_code [N, F_b, T_b], audio [N, samples])."""
    from sample import sample_model
    n = max(int(torch.as_tensor(v).numel()) for v in class_conditioning_bottom.values())
    top = top_code.to(device).expand(n, -1, -1)
    bottom = sample_model(transformer_bottom, device, n, transformer_bottom.shape, temperature, condition=top,
                          class_conditioning=class_conditioning_bottom, top_k_sampling_k=top_k_sampling_k,
                          top_p_sampling_p=top_p_sampling_p, generator=generator, uniforms=uniforms)
    audio = spectrograms_helper.to_audio(vqvae.decode_code(top, bottom))
    return bottom, audio


def adapt_duration(duration_n: int, fs_hz: int, max_sound_duration_s: float, top_resolution_n: int, top_duration: int) -> int:
    """flask_server.py:602-621: trim to the maximal duration, round to the resolution of the VQ-VAE's top level, at least
    one transformer window."""
    duration_n = min(max_sound_duration_s * fs_hz, duration_n)
    return int(top_resolution_n * max(top_duration, round(duration_n / top_resolution_n)))


@torch.no_grad()
def top_resolution_n(vqvae, transformer_top, transformer_bottom, spectrograms_helper, device) -> int:
    """Samples of audio per column of the top codemap (flask_server.py:582-599): from a decoded dummy map."""
    dummy_top = torch.zeros((1,) + tuple(transformer_top.shape), dtype=torch.long, device=device)
    dummy_bottom = torch.zeros((1,) + tuple(transformer_bottom.shape), dtype=torch.long, device=device)
    audio = spectrograms_helper.to_audio(vqvae.decode_code(dummy_top, dummy_bottom))
    return audio.shape[-1] // transformer_top.shape[1]


@torch.no_grad()
def analyze_audio(vqvae, spectrograms_helper, audio: torch.Tensor, duration_n: int, device):
    """The compute of `/analyze-audio` (flask_server.py:624-667): mono audio [samples] at the models' rate, trimmed /
    zero-padded to `duration_n` (what `from_wavfile(path, duration_n=...)` does), -> spectrogram -> `VQVAE.encode` ->
    (top_code, bottom_code)."""
    x = audio.to(device=device, dtype=torch.float32).reshape(-1)[:duration_n]
    if x.numel() < duration_n:
        x = torch.nn.functional.pad(x, (0, duration_n - x.numel()))
    spec = spectrograms_helper.to_spectrogram(x.unsqueeze(0))
    _, _, _, top_code, bottom_code, *_ = vqvae.encode(spec)
    return top_code, bottom_code
