"""HTTP surface of the interactive inpainting service (reference flask_server.py): same routes,
query arguments and JSON schema for the operations built on MI355X

  POST /timerange-change?layer=&temperature=&start_index_top=&uniform_sampling=&pitch=&instrument_family_str=
       body {top_code: int[F_t][T], bottom_code: int[F_b][T_b], mask: bool[F][W],
             top_conditioning, bottom_conditioning: {modality: value[F][T]}}          (:685-870,934-1000)
  GET/POST /generate?pitch=&instrument_family_str=&temperature=                         (:376-443)
  POST /erase?eraser_amplitude=&start_index_top=                                        (:873-931)
  POST /get-audio   -> audio/wav (16-bit PCM written with the standard library)         (:1003-1021)
  POST /analyze-audio?pitch=&instrument_family_str=   multipart file `audio` (RIFF / PCM 16-bit or float32, mono or
       averaged to mono, at the models' sampling rate: read with the standard library, no resampler)      (:624-667)
  POST /top-conditioned-sample?instrument_family_str=&min_pitch=&max_pitch=&temperature=&top_p=&top_k=
       body {top_code, bottom_code} -> application/zip of `{family}-{pitch}.wav`, one per pitch             (:1049-1115)
  GET/POST /sample-from-dataset?duration_top=&pitch=&pitch_class=&octave=&instrument_family_str=
       a stored codemap pair meeting the constraints (`codes_dataset`: the LMDB code database or any
       sequence of its items; 404 when nothing matches -- the reference searches for ever)                 (:333-372,446-514)
  GET/POST /test-generate?pitch=&instrument_family_str=   uniformly random codemaps                         (:517-552)
  POST /get-spectrogram-image   body {top_code, bottom_code} -> image/png (viridis, standard library)       (:1024-1046)
  response {top_code, bottom_code, top_conditioning, bottom_conditioning}               (:991-1000)

A thin adapter: parsing / serialisation here, all compute in `inpainting.py`.  The models (and the
code database) are handed to `create_app` already loaded: checkpoint paths are deployment plumbing.
Global (non-local) class conditioning only, like the reference's default deployment.
"""
from __future__ import annotations

import io
import struct
import zipfile
from typing import Mapping, Optional

import flask
import torch

import inpainting


def _wav_bytes(audio: torch.Tensor, fs_hz: int) -> bytes:
    pcm = (audio.clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().numpy().tobytes()
    header = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, fs_hz, fs_hz * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    return header + pcm


def _read_wav(data: bytes):
    """RIFF/WAVE bytes -> (float32 mono tensor in [-1, 1], sampling rate): PCM 16-bit or IEEE float32, the two formats
    the reference's front end uploads (the reference decodes with torchaudio, absent here)."""
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif tag == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError("WAVE file without fmt / data chunk")
    code, channels, rate, _, _, bits = fmt
    if code == 1 and bits == 16:
        x = torch.frombuffer(bytearray(payload[:len(payload) // 2 * 2]), dtype=torch.int16).float() / 32768.0
    elif code == 3 and bits == 32:
        x = torch.frombuffer(bytearray(payload[:len(payload) // 4 * 4]), dtype=torch.float32).clone()
    else:
        raise ValueError(f"unsupported WAVE encoding (format {code}, {bits} bits)")
    if channels > 1:
        x = x[:x.numel() // channels * channels].reshape(-1, channels).mean(1)
    return x, rate


def create_app(vqvae, transformer_top, transformer_bottom, label_encoders_per_modality: Mapping[str, object],
               device, spectrograms_helper=None, top_k: int = 0, top_p: float = 0.0,
               seed: Optional[int] = None, max_sound_duration_s: float = 4.0, codes_dataset=None,
               spectrograms_upsampling_factor: int = 1) -> flask.Flask:
    """`label_encoders_per_modality[name].transform([value]) -> [class index]` (sklearn LabelEncoder
    in the reference, `utils/datasets/label_encoders.py`)."""
    app = flask.Flask(__name__)
    device = torch.device(device)
    generator = torch.Generator().manual_seed(seed) if seed is not None else None
    sampling = dict(top_k_sampling_k=top_k, top_p_sampling_p=top_p)

    def conditioning_tensors(args) -> Mapping[str, torch.Tensor]:
        values = {'pitch': args.get('pitch', type=int), 'instrument_family_str': str(args.get('instrument_family_str'))}
        return {name: torch.as_tensor(label_encoders_per_modality[name].transform([value])).long().reshape(1, 1)
                for name, value in values.items()}, values

    def codes(json_data):
        return (torch.LongTensor(json_data['top_code']).unsqueeze(0).to(device),
                torch.LongTensor(json_data['bottom_code']).unsqueeze(0).to(device))

    def respond(top_code, bottom_code, conditioning_top, conditioning_bottom):
        return flask.jsonify({'top_code': top_code[0].int().cpu().numpy().tolist(),
                              'bottom_code': bottom_code[0].int().cpu().numpy().tolist(),
                              'top_conditioning': conditioning_top, 'bottom_conditioning': conditioning_bottom})

    def matrix(shape, value):
        return [[value] * shape[1]] * shape[0]

    @app.route('/generate', methods=['GET', 'POST'])
    def generate():
        args = flask.request.args
        cls, values = conditioning_tensors(args)
        top, bottom = inpainting.generate(transformer_top, transformer_bottom, float(args.get('temperature')), cls, cls,
                                          device, generator=generator, **sampling)
        return respond(top, bottom, {k: matrix(transformer_top.shape, v) for k, v in values.items()},
                       {k: matrix(transformer_bottom.shape, v) for k, v in values.items()})

    @app.route('/timerange-change', methods=['POST'])
    def timerange_change():
        args = flask.request.args
        json_data = flask.request.get_json(force=True)
        cls, values = conditioning_tensors(args)
        top_code, bottom_code = codes(json_data)
        mask = torch.BoolTensor(json_data['mask']).unsqueeze(0)
        layer = str(args.get('layer'))
        uniform = str(args.get('uniform_sampling', default='False')).lower() in ('1', 'true', 'yes', 'y', 'on', 't')
        new_top, new_bottom = inpainting.timerange_change(
            transformer_top, transformer_bottom, top_code, bottom_code, mask, layer,
            args.get('start_index_top', type=int), args.get('temperature', type=float), cls, cls, device,
            uniform_sampling=uniform, generator=generator, **sampling)
        cond_top, cond_bottom = json_data.get('top_conditioning'), json_data.get('bottom_conditioning')
        if layer == 'top' and cond_bottom is not None:
            # the regenerated zone now carries the requested classes (flask_server.py:845-853)
            ratio_f = transformer_bottom.shape[0] // transformer_top.shape[0]
            ratio_t = transformer_bottom.shape[1] // transformer_top.shape[1]
            m = mask[0].repeat_interleave(ratio_f, 0).repeat_interleave(ratio_t, 1).tolist()
            cond_bottom = {name: [[values[name] if mv else pv for pv, mv in zip(row, mrow)]
                                  for row, mrow in zip(rows, m)] for name, rows in cond_bottom.items()}
        return respond(new_top, new_bottom, cond_top, cond_bottom)

    @app.route('/erase', methods=['POST'])
    def erase():
        args = flask.request.args
        json_data = flask.request.get_json(force=True)
        top_code, bottom_code = codes(json_data)
        new_top, new_bottom = inpainting.erase(vqvae, top_code, bottom_code, torch.BoolTensor(json_data['mask']),
                                               float(args.get('eraser_amplitude')), int(args.get('start_index_top')))
        return respond(new_top, new_bottom, json_data.get('top_conditioning'), json_data.get('bottom_conditioning'))

    @app.route('/get-audio', methods=['POST'])
    def get_audio():
        if spectrograms_helper is None:
            flask.abort(501)
        top_code, bottom_code = codes(flask.request.get_json(force=True))
        audio = inpainting.codes_to_audio(vqvae, spectrograms_helper, top_code, bottom_code)[0]
        return flask.send_file(io.BytesIO(_wav_bytes(audio, spectrograms_helper.fs_hz)), mimetype="audio/wav",
                               max_age=0)

    @app.route('/analyze-audio', methods=['POST'])
    def analyze_audio():
        if spectrograms_helper is None:
            flask.abort(501)
        args = flask.request.args
        values = {'pitch': args.get('pitch', type=int), 'instrument_family_str': str(args.get('instrument_family_str'))}
        try:
            audio, rate = _read_wav(flask.request.files['audio'].read())
        except (KeyError, ValueError, struct.error) as e:
            flask.abort(400, description=str(e))
        if rate != spectrograms_helper.fs_hz:
            flask.abort(400, description=f"audio at {rate} Hz: the models run at {spectrograms_helper.fs_hz} Hz (no resampler here)")
        res_n = inpainting.top_resolution_n(vqvae, transformer_top, transformer_bottom, spectrograms_helper, device)
        duration_n = inpainting.adapt_duration(audio.numel(), spectrograms_helper.fs_hz, max_sound_duration_s, res_n,
                                               transformer_top.shape[1])
        top_code, bottom_code = inpainting.analyze_audio(vqvae, spectrograms_helper, audio, duration_n, device)
        return respond(top_code, bottom_code, {k: matrix(transformer_top.shape, v) for k, v in values.items()},
                       {k: matrix(transformer_bottom.shape, v) for k, v in values.items()})

    @app.route('/top-conditioned-sample', methods=['POST'])
    def top_conditioned_sample():
        if spectrograms_helper is None:
            flask.abort(501)
        args = flask.request.args
        top_code, _ = codes(flask.request.get_json(force=True))
        family = str(args.get('instrument_family_str'))
        lo, hi = args.get('min_pitch', type=int), args.get('max_pitch', type=int)
        if lo is None or hi is None or not lo < hi:
            flask.abort(400, description="Provide increasing range for range conditioning")
        cls = {'pitch': torch.as_tensor(label_encoders_per_modality['pitch'].transform(list(range(lo, hi)))).long(),
               'instrument_family_str': torch.as_tensor(
                   label_encoders_per_modality['instrument_family_str'].transform([family])).long()}
        _, audio = inpainting.top_conditioned_sample(
            vqvae, transformer_bottom, spectrograms_helper, top_code, float(args.get('temperature')), cls, device,
            top_k_sampling_k=int(args.get('top_k') or 0), top_p_sampling_p=float(args.get('top_p') or 0.0),
            generator=generator)
        buf = io.BytesIO()
        with zipfile.ZipFile(buf, 'w') as zf:        # (in memory: the reference writes sample files and a zip to its upload folder)
            for pitch, row in zip(range(lo, hi), audio):
                zf.writestr(f'{family}-{pitch}.wav', _wav_bytes(row, spectrograms_helper.fs_hz))
        buf.seek(0)
        return flask.send_file(buf, mimetype="application/zip", max_age=0)

    @app.route('/sample-from-dataset', methods=['GET', 'POST'])
    def sample_from_dataset():
        if codes_dataset is None:
            flask.abort(501)
        args = flask.request.args
        constraints = {}
        pitch = args.get('pitch', type=int, default=None)
        if pitch is not None:
            constraints['pitch'] = pitch
        pitch_class = args.get('pitch_class', type=int, default=None)
        if pitch_class is not None and 0 <= pitch_class <= 12:
            constraints['pitch_class'] = pitch_class
        octave = args.get('octave', type=int, default=None)
        if octave is not None and octave >= 0:
            constraints['octave'] = octave
        family = args.get('instrument_family_str', type=str, default=None)
        if family is not None:
            constraints['instrument_family_str'] = family
        try:
            (top, bottom), attributes = inpainting.sample_from_database(
                codes_dataset, label_encoders_per_modality, args.get('duration_top', type=int), constraints, generator)
        except LookupError as e:
            flask.abort(404, description=str(e))
        values = {'pitch': int(attributes['pitch']), 'instrument_family_str': str(attributes['instrument_family_str'])}
        return respond(top, bottom, {k: matrix(top.shape[1:], v) for k, v in values.items()},
                       {k: matrix(bottom.shape[1:], v) for k, v in values.items()})

    @app.route('/test-generate', methods=['GET', 'POST'])
    def test_generate():
        args = flask.request.args
        values = {'pitch': int(args.get('pitch')), 'instrument_family_str': str(args.get('instrument_family_str'))}
        top = torch.randint(0, vqvae.n_embed_t, (1,) + tuple(transformer_top.shape), generator=generator)
        bottom = torch.randint(0, vqvae.n_embed_b, (1,) + tuple(transformer_bottom.shape), generator=generator)
        return respond(top, bottom, {k: matrix(transformer_top.shape, v) for k, v in values.items()},
                       {k: matrix(transformer_bottom.shape, v) for k, v in values.items()})

    @app.route('/get-spectrogram-image', methods=['POST'])
    def get_spectrogram_image():
        top_code, bottom_code = codes(flask.request.get_json(force=True))
        png = inpainting.spectrogram_png(vqvae, top_code, bottom_code, spectrograms_upsampling_factor)
        return flask.send_file(io.BytesIO(png), mimetype="image/png", max_age=0)

    return app
