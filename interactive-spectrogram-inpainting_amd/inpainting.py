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
  top_conditioned_sample / analyze_audio    (:1049-1110 / :624-667)
  sample_from_database   a stored (top, bottom) pair whose attributes meet the request's constraints,
                      cut / padded to the requested duration (:314-372, 446-514)
  spectrogram_png     the decoded log-mel magnitude as a viridis PNG (:103-143, 1024-1046)

The heavy work is `sample.sample_model` (encoder pass + KV-cached native decoding loop),
`VQVAE.decode_code / encode` and `SpectrogramsHelper.to_audio`: all HIP kernels.
"""
from __future__ import annotations

import struct
import zlib
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


def resize_codemaps_repeat_last(top_code: torch.Tensor, bottom_code: torch.Tensor, duration_top: int):
    """flask_server.py:314-330: cut to `duration_top` columns of the top map (and the matching number of the bottom map);
    a shorter map is continued with its last column."""
    ratio = bottom_code.shape[-1] // top_code.shape[-1]

    def resize(codemap, duration):
        codemap = codemap[..., :duration]
        if codemap.shape[-1] < duration:
            codemap = torch.cat([codemap, codemap[..., -1:].expand(*codemap.shape[:-1], duration - codemap.shape[-1])], -1)
        return codemap
    return resize(top_code, duration_top), resize(bottom_code, ratio * duration_top)


def sample_from_database(codes_dataset, label_encoders_per_modality: Mapping[str, object], duration_top: int,
                         attribute_constraints: Mapping[str, object], generator: Optional[torch.Generator] = None):
    """The lookup of `/sample-from-dataset` (flask_server.py:333-372): items of the code database
    (`utils.datasets.lmdb_dataset.LMDBDataset`: (top [F_t,T_t], bottom [F_b,T_b], {class name: tensor [1]})) are visited in
    a random order until one meets every constraint -- equality on the decoded attributes, `pitch_class` = pitch % 12 and
    `octave` = pitch // 12 derived from the pitch (the reference stores the octave under the `pitch_class` key, :351-353,
    so its octave constraint can never be met; the evident intent is built here).  The reference draws with replacement
    for ever; a database without a match raises LookupError here.  Returns ((top [1,F_t,T], bottom [1,F_b,T_b]), attributes)."""
    for index in torch.randperm(len(codes_dataset), generator=generator).tolist():
        top_code, bottom_code, encoded = codes_dataset[index]
        attributes = {key: label_encoders_per_modality[key].inverse_transform([int(torch.as_tensor(value).reshape(-1)[0])])[0]
                      for key, value in encoded.items()}
        if 'pitch' in attributes:
            attributes['pitch_class'] = int(attributes['pitch']) % 12
            attributes['octave'] = int(attributes['pitch']) // 12
        if all(key in attributes and attributes[key] == wanted for key, wanted in attribute_constraints.items()):
            top_code, bottom_code = torch.as_tensor(top_code).unsqueeze(0), torch.as_tensor(bottom_code).unsqueeze(0)
            return resize_codemaps_repeat_last(top_code, bottom_code, duration_top), attributes
    raise LookupError(f"no item of the code database meets {dict(attribute_constraints)}")


# nine samples of the viridis colour map (0, 1/8, .. 1), interpolated linearly: the map the reference draws with
_VIRIDIS = ((68, 1, 84), (71, 44, 122), (59, 81, 139), (44, 113, 142), (33, 144, 141), (39, 173, 129), (92, 200, 99),
            (170, 220, 50), (253, 231, 37))


def _png_bytes(rgb: torch.Tensor) -> bytes:
    """[H, W, 3] uint8 -> PNG (8-bit truecolour, one IDAT chunk; standard library only)."""
    h, w, _ = rgb.shape
    rows = torch.cat([torch.zeros(h, 1, dtype=torch.uint8), rgb.reshape(h, w * 3)], 1)      # filter type 0 per scanline

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(rows.numpy().tobytes(), 6)) + chunk(b"IEND", b""))


@torch.no_grad()
def spectrogram_png(vqvae, top_code: torch.Tensor, bottom_code: torch.Tensor, upsampling_factor: int = 1) -> bytes:
    """The compute of `/get-spectrogram-image` (flask_server.py:103-143, 1024-1046): decode the codes, take the
    log-magnitude channel of the first item, upsample it bilinearly, and draw it full-frame in viridis with the lowest
    frequency at the bottom and the colour range spanning the map's own minimum .. maximum (what `specshow` does).  The
    reference rasterises through matplotlib at 2400 x 1600; here one pixel per (upsampled) spectrogram bin."""
    s = vqvae.decode_code(top_code, bottom_code)[0, 0].float()
    if upsampling_factor > 1:
        s = torch.nn.functional.interpolate(s[None, None], mode='bilinear', scale_factor=upsampling_factor)[0, 0]
    s = s.flip(0).cpu()
    lo, hi = float(s.min()), float(s.max())
    t = ((s - lo) / (hi - lo) if hi > lo else torch.zeros_like(s)).clamp(0, 1) * (len(_VIRIDIS) - 1)
    i0 = t.floor().long().clamp(max=len(_VIRIDIS) - 2)
    lut = torch.tensor(_VIRIDIS, dtype=torch.float32)
    frac = (t - i0).unsqueeze(-1)
    rgb = (lut[i0] * (1 - frac) + lut[i0 + 1] * frac).round().to(torch.uint8)
    return _png_bytes(rgb)
