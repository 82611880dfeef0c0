"""Autoregressive / inpainting sampling of codemaps on MI355X.

Drop-in for `sample_model` and `top_k_top_p_filtering` of the reference's
`sample.py:36-65,131-347` (the CLI, audio and PNG writing of that script are
harness code outside the compute path).  Contract kept:
positions with `mask == False` keep `initial_code`; masked positions are sampled
left to right in the target helper's order; `temperature` divides the logits;
the result is an int64 codemap `[B, F, T]`.

Differences by design: the decoder runs incrementally on cached keys / values
(same logits as the reference's full pass per token, O(S) instead of O(S^2) layer
passes); filtering + softmax + the categorical draw are one kernel on the single
row that is consumed; the draw uses uniforms from `generator` (or given `uniforms`)
instead of torch.multinomial's stream.
"""
from __future__ import annotations

from typing import Iterable, Mapping, Optional, Union

import torch

from interactive_spectrogram_inpainting.priors import _ops
from interactive_spectrogram_inpainting.priors._decode import NativeSampler
from interactive_spectrogram_inpainting.priors.transformer import Seq2SeqInputKind, VQNSynthTransformer


def top_k_top_p_filtering(logits: torch.Tensor, top_k: int = 0, top_p: float = 0.0,
                          filter_value: float = -float('Inf')) -> torch.Tensor:
    """Filter logits `[..., n_class]` in place (like the reference) with top-k and/or
    nucleus filtering; runs `isi_sample_row_f32`'s filtering stage on every row."""
    if filter_value != -float('Inf'):
        raise NotImplementedError("only filter_value = -inf is built")
    rows = logits.reshape(-1, logits.shape[-1])
    u = torch.zeros(rows.shape[0], device=logits.device)
    _, filt = _ops.sample_rows(rows, 1.0, top_k, top_p, u, return_filtered=True)
    logits.copy_(filt.reshape(logits.shape))
    return logits


@torch.no_grad()
def sample_model(model: VQNSynthTransformer, device: Union[torch.device, str], batch_size: int,
                 codemap_size: Iterable[int], temperature: float,
                 condition: Optional[torch.Tensor] = None, constraint: Optional[torch.Tensor] = None,
                 class_conditioning: Mapping[str, Iterable[int]] = {},
                 initial_code: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                 local_class_conditioning_map=None,
                 time_indexes_source: Optional[Iterable[int]] = None,
                 time_indexes_target: Optional[Iterable[int]] = None,
                 top_k_sampling_k: int = 0, top_p_sampling_p: float = 0.0,
                 progressbar_decorator=None, use_predictive_sampling: bool = False,
                 generator: Optional[torch.Generator] = None,
                 uniforms: Optional[torch.Tensor] = None,
                 gumbel_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    if constraint is not None:
        raise NotImplementedError
    device = torch.device(device)
    model.eval()
    if batch_size > 256:
        # the native loop decodes up to 256 sequences at a time: larger requests run chunk by chunk (rows are
        # independent; every chunk gets its own slice of the per-row inputs and of the uniforms)
        if uniforms is None:
            uniforms = torch.rand(model.target_transformer_sequence_length, batch_size, generator=generator)

        def rows(t, lo, hi):
            if t is None or not torch.is_tensor(t) or t.dim() == 0 or t.shape[0] != batch_size:
                return t
            return t[lo:hi]

        parts = []
        for lo in range(0, batch_size, 256):
            hi = min(batch_size, lo + 256)
            cls = {k: (torch.as_tensor(v).reshape(-1)[lo:hi] if torch.as_tensor(v).numel() == batch_size else v)
                   for k, v in class_conditioning.items()}
            # every per-row input is sliced (a mask with a batch dimension included); options are passed through
            parts.append(sample_model(
                model, device, hi - lo, codemap_size, temperature, condition=rows(condition, lo, hi),
                class_conditioning=cls, initial_code=rows(initial_code, lo, hi), mask=rows(mask, lo, hi),
                time_indexes_source=time_indexes_source, time_indexes_target=time_indexes_target,
                top_k_sampling_k=top_k_sampling_k, top_p_sampling_p=top_p_sampling_p,
                progressbar_decorator=progressbar_decorator, use_predictive_sampling=use_predictive_sampling,
                uniforms=uniforms[:, lo:hi], gumbel_noise=rows(gumbel_noise, lo, hi)))
        return torch.cat(parts, 0)
    if initial_code is None:
        fill = model.mask_token_index if model.self_conditional_model else 0
        codemap = torch.full([batch_size] + list(codemap_size), fill, dtype=torch.int64, device=device)
    else:
        codemap = initial_code.to(device)
    cls = {}
    for name, value in class_conditioning.items():
        value = torch.as_tensor(value).long().reshape(-1)
        cls[name] = (value.expand(batch_size) if value.numel() == 1 else value).reshape(batch_size, 1).to(device)
    if model.self_conditional_model:
        condition = codemap
    if mask is not None:
        mask = mask.to(device)
    # With no initial_code the reference fills the map with the mask token, which its
    # target embedding cannot index (sample.py:167-171 vs priors/transformer.py:286: an
    # IndexError there).  Those target rows are always overwritten by samples before the
    # decoder reads them, so the target side is built from in-range placeholders.
    target_codemap = codemap.clamp(max=model.n_class_target - 1) if initial_code is None else codemap
    source_seq, target_seq = model.to_sequences(
        target_codemap, condition.to(device), class_conditioning=cls, mask=mask,
        time_indexes_source=time_indexes_source, time_indexes_target=time_indexes_target)

    S = model.target_transformer_sequence_length
    start_len = model.target_start_symbol.shape[1]
    code_seq = model.target_codemaps_helper.to_sequence(codemap).clone()
    if mask is not None:
        mask_seq = model.target_codemaps_helper.to_sequence(mask).reshape(-1, S)[0].cpu().numpy()
    else:
        mask_seq = [True] * S
    if use_predictive_sampling:
        return _predictive_sampling(model, source_seq, target_seq, code_seq, mask_seq, start_len, temperature,
                                    top_k_sampling_k, top_p_sampling_p, gumbel_noise, progressbar_decorator)
    if uniforms is None:
        uniforms = torch.rand(S, batch_size, generator=generator)
    uniforms = uniforms.to(device=device, dtype=torch.float32)

    # encoder memory once (anti-causal for the self-conditional top prior)
    src = source_seq.transpose(0, 1).contiguous()
    memory, *_ = model.transformer.encoder(src, mask='anticausal' if model.self_conditional_model else None)
    x_seq = target_seq.transpose(0, 1).contiguous()          # [S_t, B, d]; rows are rewritten as we sample
    if not code_seq.is_contiguous():
        code_seq = code_seq.contiguous()
    n_pos = S + start_len - 1
    # Token i is drawn from decoder position i + start_len - 1.  Positions behind the last masked token are
    # never read; positions before the first one only contribute keys / values, which one batched causal
    # pass over that prefix provides (an inpainting request masks a window, not the whole map).
    masked = [i for i, mk in enumerate(mask_seq) if mk]
    if not masked:
        return model.target_codemaps_helper.to_time_frequency_map(code_seq).long()
    p_first, n_pos = masked[0] + start_len - 1, min(n_pos, masked[-1] + start_len)
    # the whole loop natively: no per-token return to Python, no host sync
    sampler = NativeSampler(model, memory, x_seq, code_seq, mask_seq, uniforms)
    if p_first < 8:
        p_first = 0                                        # a few rows: not worth a batched pass
    sampler.prefill(p_first)
    chunk = n_pos if progressbar_decorator is None else 64
    starts = range(p_first, n_pos, chunk)
    if progressbar_decorator is not None:
        starts = progressbar_decorator(starts)
    for p0 in starts:
        sampler.run(p0, min(n_pos, p0 + chunk), temperature, top_k_sampling_k, top_p_sampling_p)
    return model.target_codemaps_helper.to_time_frequency_map(code_seq).long()


def _predictive_sampling(model, source_seq, target_seq, code_seq, mask_seq, start_len, temperature, top_k, top_p,
                         gumbel_noise, progressbar_decorator):
    """Predictive sampling of the reference (sample.py:251-261,268-342): with the Gumbel reparametrisation
    `argmax(log p + g)` and the noise `g` drawn up front, every full pass also FORECASTS all later masked tokens; a
    step whose forecast already equalled the token the previous pass put there is skipped.  The sampled map is that of
    sequential Gumbel-max sampling with the same noise.  One full decoder pass per non-skipped step (the encoder
    memory is cached), on the HIP forward kernels; the KV-cached loop does not apply -- a pass rewrites every later
    input row.  Differences from the reference: its boolean index `[1, S]` on `[B, S]` tensors only works at batch 1,
    here the mask broadcasts over the batch; `gumbel_noise` ([B, S, n_class]) may be passed in (tests)."""
    device = target_seq.device
    B, S = code_seq.shape
    eff = model.embeddings_effective_dim
    if gumbel_noise is None:
        gumbel_noise = torch.distributions.gumbel.Gumbel(torch.zeros(B, S, model.n_class_target), 1).sample()
    gumbel_noise = gumbel_noise.to(device=device, dtype=torch.float32)
    mask_t = torch.as_tensor(list(mask_seq), dtype=torch.bool, device=device)          # [S]
    positions = torch.arange(S, device=device)
    source_start = model.source_start_symbol.shape[1] if hasattr(model, "source_start_symbol") else start_len
    memory = None
    sample = None
    prediction_was_correct = False
    correct_predictions = 0
    previous = code_seq
    steps = enumerate(mask_seq)
    if progressbar_decorator is not None:
        steps = progressbar_decorator(steps)
    for i, is_masked in steps:
        if not is_masked:
            continue
        if sample is not None and prediction_was_correct:
            prediction_was_correct = bool(torch.all(sample[:, i] == previous[:, i]))
            if prediction_was_correct:
                correct_predictions += 1
                continue
        logits, memory = model(target_seq, source_seq, memory=memory)
        logits = top_k_top_p_filtering(logits / temperature, top_k=top_k, top_p=top_p)
        probabilities = torch.softmax(logits, dim=-1)
        sample = torch.argmax(torch.log(probabilities) + gumbel_noise, dim=-1)        # [B, S]
        prediction_was_correct = bool(torch.all(sample[:, i] == code_seq[:, i]))
        previous = code_seq.clone()
        update = (mask_t & (positions >= i)).unsqueeze(0).expand(B, S)                 # causal and inpainting mask
        code_seq[update] = sample[update]
        embedded = model.embed_data(sample, Seq2SeqInputKind.Target)                   # [B, S, eff]
        tgt_view = target_seq[:, start_len:, :eff]
        tgt_view[update] = embedded[update]
        if model.self_conditional_model:
            src_view = source_seq[:, source_start:, :eff]
            src_view[update] = embedded[update]
            # the cached memory stays valid: the top encoder's attention is anti-causal (reference comment)
    model.predictive_sampling_correct_ratio = correct_predictions / max(1, len(mask_seq))
    return model.target_codemaps_helper.to_time_frequency_map(code_seq).long()
