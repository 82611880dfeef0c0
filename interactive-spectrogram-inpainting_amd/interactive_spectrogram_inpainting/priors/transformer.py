"""Autoregressive priors over the VQ-VAE codemaps, MI355X-native.

Drop-in for the reference's `priors/transformer.py:24-872`
(`VQNSynthTransformer`, `SelfAttentiveVQTransformer`, `UpsamplingVQTransformer`):
same constructor keywords and JSON persistence, same parameter names / shapes
(pinned against the reference's own state dict in tests), same public methods
(`to_sequences`, `prepare_data`, `embed_data`, `forward`, `causal_mask`) and the
attributes the sampling code reads (`shape`, `mask_token_index`,
`self_conditional_model`, `source_start_symbol`, `target_start_symbol`,
`embeddings_effective_dim`, `target_codemaps_helper`, `n_class_target`, ...).

Every FLOP runs in libisi_hip.so: the token-embedding Linear is folded once per
weight version into a `[n_class, 496]` table (one GEMM), sequences are assembled
by gathers, the layers are `VQCPCB.transformer.transformer_custom` (this
repository's HIP implementation), the logits head is a GEMM.
"""
from __future__ import annotations

import collections
import json
import os
import pathlib
from enum import Enum, auto
from typing import Iterable, Mapping, Optional, Tuple, Union

import torch
from torch import nn

from VQCPCB.transformer.transformer_custom import (
    TransformerCustom, TransformerDecoderCustom, TransformerEncoderCustom,
    TransformerDecoderLayerCustom, TransformerEncoderLayerCustom,
    TransformerAlignedDecoderLayerCustom, _LinearParams)

from .. import _hip
from . import _train
from .codemaps_helpers import CodemapsHelper, SimpleCodemapsHelper, ZigZagCodemapsHelper



DEFERRED_INDEX_CHECK = os.environ.get("ISI_DEFERRED_INDEX_CHECK", "1") != "0"


class _IndexGuard:
    """Out-of-range verdicts of `embed_data` calls whose read-back is deferred: one pinned byte and one event per call."""

    def __init__(self):
        self._pending = collections.deque()

    def _raise_if(self, flag: torch.Tensor) -> None:
        if bool(flag.item()):
            self._pending.clear()
            raise IndexError("index out of range in self (seen by a deferred check of an earlier training step)")

    def submit(self, bad: torch.Tensor) -> None:
        while self._pending and self._pending[0][1].query():      # verdicts that have arrived: no waiting
            self._raise_if(self._pending.popleft()[0])
        host = torch.empty((), dtype=torch.bool, pin_memory=True)
        host.copy_(bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((host, ev))

    def flush(self) -> None:
        while self._pending:
            host, ev = self._pending.popleft()
            ev.synchronize()
            self._raise_if(host)


class Seq2SeqInputKind(Enum):
    Source = auto()
    Target = auto()


class VQNSynthTransformer(nn.Module):
    source_codemaps_helper: CodemapsHelper
    target_codemaps_helper: CodemapsHelper

    @property
    def use_inpainting_mask_on_source(self) -> bool:
        raise NotImplementedError("subclass this")

    def __init__(
        self,
        shape: Iterable[int],
        n_class: int,
        channel: int,
        kernel_size: int,
        n_block: int,
        n_res_block: int,
        res_channel: int,
        attention: bool = True,
        dropout: float = 0.1,
        n_cond_res_block: int = 0,
        cond_res_channel: int = 0,
        cond_res_kernel: int = 3,
        n_out_res_block: int = 0,
        predict_frequencies_first: bool = False,
        predict_low_frequencies_first: bool = True,
        d_model: int = 512,
        embeddings_dim: int = 32,
        positional_embeddings_dim: int = 16,
        use_relative_transformer: bool = False,
        class_conditioning_num_classes_per_modality: Optional[Mapping[str, int]] = None,
        class_conditioning_embedding_dim_per_modality: Optional[Mapping[str, int]] = None,
        class_conditioning_prepend_to_dummy_input: bool = False,
        local_class_conditioning: bool = False,
        positional_class_conditioning: bool = False,
        add_mask_token_to_symbols: bool = False,
        conditional_model: bool = False,
        self_conditional_model: bool = False,
        use_aligned_decoder: bool = False,
        condition_shape: Optional[Tuple[int, int]] = None,
        conditional_model_num_encoder_layers: int = 6,
        conditional_model_num_decoder_layers: int = 8,
        conditional_model_nhead: int = 8,
        unconditional_model_num_encoder_layers: int = 6,
        unconditional_model_nhead: int = 8,
        use_identity_memory_mask: bool = False,
        use_lstm_DEBUG: bool = False,
        disable_start_symbol_DEBUG: bool = False,
    ):
        if local_class_conditioning:
            raise NotImplementedError("Depecrated in favor of positional class conditioning")
        if not (use_relative_transformer and conditional_model):
            raise NotImplementedError("only the relative, conditional (seq2seq) configuration used by the "
                                      "reference's Self-attentive / Upsampling priors is built")
        if use_relative_transformer and not predict_frequencies_first:
            raise NotImplementedError("Relative positioning only implemented along time")
        if use_lstm_DEBUG:
            raise NotImplementedError("TODO(theis), debug mode with simple LSTM layers")
        shape = list(shape)
        if self_conditional_model:
            assert condition_shape is None or list(condition_shape) == shape
        assert condition_shape is not None or self_conditional_model

        self.shape = shape
        self.conditional_model = conditional_model
        self.self_conditional_model = self_conditional_model
        self.use_relative_transformer = use_relative_transformer
        self.condition_shape = list(shape) if self_conditional_model else list(condition_shape)
        self.local_class_conditioning = local_class_conditioning
        self.positional_class_conditioning = positional_class_conditioning
        self.n_class = n_class
        self.channel = channel
        self.kernel_size = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
        self.n_block = n_block
        self.n_res_block = n_res_block
        self.res_channel = res_channel
        self.dropout = dropout
        self.n_cond_res_block = n_cond_res_block
        self.cond_res_channel = cond_res_channel
        self.cond_res_kernel = cond_res_kernel
        self.n_out_res_block = n_out_res_block
        self.predict_frequencies_first = predict_frequencies_first
        self.predict_low_frequencies_first = predict_low_frequencies_first
        self.d_model = d_model
        self.embeddings_dim = embeddings_dim
        self.positional_embeddings_dim = 2 * (positional_embeddings_dim // 2)
        self.class_conditioning_num_classes_per_modality = class_conditioning_num_classes_per_modality
        self.class_conditioning_embedding_dim_per_modality = class_conditioning_embedding_dim_per_modality
        self.class_conditioning_prepend_to_dummy_input = class_conditioning_prepend_to_dummy_input
        self.conditional_model_num_encoder_layers = conditional_model_num_encoder_layers
        self.conditional_model_nhead = conditional_model_nhead
        self.conditional_model_num_decoder_layers = conditional_model_num_decoder_layers
        self.use_identity_memory_mask = use_identity_memory_mask
        self.use_aligned_decoder = use_aligned_decoder
        self.use_lstm_DEBUG = use_lstm_DEBUG
        self.disable_start_symbol_DEBUG = disable_start_symbol_DEBUG
        self._instantiation_parameters = self.__dict__.copy()

        super().__init__()

        if self.use_inpainting_mask_on_source:
            self.n_class_source = self.n_class + 1
            self.mask_token_index = self.n_class_source - 1
            self.n_class_target = self.n_class
        else:
            self.n_class_target = self.n_class_source = self.n_class

        if class_conditioning_num_classes_per_modality is not None:
            self.class_conditioning_num_modalities = len(class_conditioning_embedding_dim_per_modality)
            self.class_conditioning_total_dim = sum(class_conditioning_embedding_dim_per_modality.values())
            if not (class_conditioning_prepend_to_dummy_input or positional_class_conditioning):
                raise NotImplementedError     # no start positions would be defined (reference :304)
        else:
            self.class_conditioning_num_modalities = 0
            self.class_conditioning_total_dim = 0

        self.source_frequencies, self.source_duration = self.condition_shape
        self.source_num_channels, self.source_num_events = 1, self.source_frequencies * self.source_duration
        self.source_transformer_sequence_length = self.source_frequencies * self.source_duration
        self.target_frequencies, self.target_duration = self.shape
        self.target_transformer_sequence_length = self.target_frequencies * self.target_duration
        self.target_events_per_source_patch = ((self.target_duration // self.source_duration)
                                               * (self.target_frequencies // self.source_frequencies))
        self.target_num_channels = self.target_events_per_source_patch
        self.target_num_events = self.target_transformer_sequence_length // self.target_num_channels
        self.output_sizes = (-1, self.target_frequencies, self.target_duration, self.n_class_target)

        pe = self.positional_embeddings_dim // 2
        self.source_positional_embeddings_frequency = nn.Parameter(torch.randn(1, self.source_frequencies, 1, pe))
        self.source_positional_embeddings_time = nn.Parameter(torch.randn(1, 1, self.source_duration, pe))
        self.target_positional_embeddings_time = None
        self.target_positional_embeddings_patch = nn.Parameter(torch.randn(
            1, self.target_frequencies // self.source_frequencies, self.target_duration // self.source_duration, pe))
        self.target_positional_embeddings_frequency = nn.Parameter(torch.randn(1, self.target_frequencies, 1, pe))

        if self.embeddings_dim is None:
            self.embeddings_dim = self.d_model - self.positional_embeddings_dim
        self.source_embed = nn.Embedding(self.n_class_source, self.embeddings_dim)
        self.embeddings_effective_dim = self.d_model - self.positional_embeddings_dim
        if self.positional_class_conditioning:      # class embeddings ride on every position (:270-271)
            self.embeddings_effective_dim -= self.class_conditioning_total_dim
        self.source_embeddings_linear = _LinearParams(self.embeddings_dim, self.embeddings_effective_dim)
        self.target_embeddings_linear = _LinearParams(self.embeddings_dim, self.embeddings_effective_dim)
        self.target_embed = nn.Embedding(self.n_class_target, self.embeddings_dim)
        self.project_transformer_outputs_to_logits = _LinearParams(self.d_model, self.n_class_target)

        self.class_conditioning_embedding_layers = nn.ModuleDict()
        self.class_conditioning_class_to_index_per_modality = {}
        self.class_conditioning_start_positions_per_modality = {}
        if class_conditioning_num_classes_per_modality is not None:
            pos = 0
            for (name, n_cls), dim in zip(class_conditioning_num_classes_per_modality.items(),
                                          class_conditioning_embedding_dim_per_modality.values()):
                self.class_conditioning_embedding_layers[name] = nn.Embedding(n_cls, dim)
            for name, dim in class_conditioning_embedding_dim_per_modality.items():
                self.class_conditioning_start_positions_per_modality[name] = pos
                pos += dim
        self.class_conditioning_total_dim_with_positions = (self.class_conditioning_total_dim
                                                            + self.positional_embeddings_dim)

        self.source_start_symbol_dim = self.d_model
        if self.positional_class_conditioning:      # the class embeddings are appended to the start symbols too (:331,345)
            self.source_start_symbol_dim -= self.class_conditioning_total_dim
        self.source_start_symbol = nn.Parameter(torch.randn(1, 1, self.source_start_symbol_dim))
        self.source_num_events_with_start_symbol = self.source_num_events + 1
        self.source_transformer_sequence_length_with_start_symbol = self.source_transformer_sequence_length + 1
        self.target_start_symbol_dim = self.d_model
        if self.positional_class_conditioning:
            self.target_start_symbol_dim -= self.class_conditioning_total_dim
        self.target_start_symbol = nn.Parameter(
            torch.randn(1, self.target_events_per_source_patch, self.target_start_symbol_dim))
        self.target_num_events_with_start_symbol = self.target_num_events + 1
        self.target_transformer_sequence_length_with_start_symbol = (
            self.target_num_events_with_start_symbol * self.target_num_channels)

        encoder_layer = TransformerEncoderLayerCustom(
            d_model=self.d_model, nhead=self.conditional_model_nhead,
            attention_bias_type='relative_attention',
            num_channels=self.source_num_channels, num_events=self.source_num_events_with_start_symbol)
        relative_encoder = TransformerEncoderCustom(encoder_layer=encoder_layer,
                                                    num_layers=self.conditional_model_num_encoder_layers)
        attention_bias_type_cross = 'no_bias' if self.use_identity_memory_mask else 'relative_attention_target_source'
        decoder_impl = TransformerAlignedDecoderLayerCustom if self.use_aligned_decoder else TransformerDecoderLayerCustom
        decoder_layer = decoder_impl(
            d_model=self.d_model, nhead=self.conditional_model_nhead,
            attention_bias_type_self='relative_attention',
            attention_bias_type_cross=attention_bias_type_cross,
            num_channels_encoder=self.source_num_channels,
            num_events_encoder=self.source_num_events_with_start_symbol,
            num_channels_decoder=self.target_num_channels,
            num_events_decoder=self.target_num_events_with_start_symbol)
        custom_decoder = TransformerDecoderCustom(decoder_layer=decoder_layer,
                                                  num_layers=self.conditional_model_num_decoder_layers)
        self.transformer = TransformerCustom(nhead=self.conditional_model_nhead, custom_encoder=relative_encoder,
                                             custom_decoder=custom_decoder, d_model=self.d_model)
        self._tables = {}
        self._index_guard = _IndexGuard()

    # ---------------------------------------------------------------- embeddings
    def _embedding_table(self, kind: Seq2SeqInputKind) -> torch.Tensor:
        """[n_class, effective_dim] = Linear(Embedding.weight): the per-token result of
        `embed_data` for every symbol, computed by one GEMM per weight version."""
        if kind == Seq2SeqInputKind.Source:
            emb, lin = self.source_embed, self.source_embeddings_linear
        elif kind == Seq2SeqInputKind.Target and self.conditional_model:
            emb, lin = self.target_embed, self.target_embeddings_linear
        else:
            raise ValueError(f"Unexpected value {kind} for kind option")
        key = (kind, _hip.version_of(emb.weight), emb.weight.data_ptr(), lin.weight._version, lin.weight.data_ptr(),
               lin.bias._version)
        if self._differentiable():
            return lin.run(emb.weight)  # recorded: gradients reach the embedding and the linear layer
        hit = self._tables.get(kind)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, lin.run(emb.weight.detach()))
            self._tables[kind] = hit
        return hit[1]

    def _differentiable(self) -> bool:
        """The training path records autograd history; inference (eval / no_grad) uses cached tables."""
        return self.training and torch.is_grad_enabled()

    def embed_data(self, input: torch.Tensor, kind: Seq2SeqInputKind) -> torch.Tensor:
        table = self._embedding_table(kind)
        if input.numel():
            if input.is_cuda and torch.cuda.is_current_stream_capturing():
                # a step recorded into a HIP graph: the eager steps in front of the capture did the checking, the
                # replays run on clamped indices (GraphedTrainingStep checks the batches it is handed)
                input = input.clamp(0, table.shape[0] - 1)
                lo = hi = None
            else:
                lo, hi = torch.aminmax(input)          # one launch
            if lo is None:
                pass
            elif self._differentiable() and input.is_cuda and DEFERRED_INDEX_CHECK:
                # A training step must not JOIN the device at its head (the read-back below made the host wait for the previous
                # step's backward and optimizer: ~12 ms of a 30-ms step during which nothing was enqueued).  The verdict
                # travels to pinned memory behind the launches and is looked at by a later call / `check_indices()`; the
                # gather runs on clamped indices meanwhile.  (The reference on a GPU fails the same way: nn.Embedding's
                # device-side assert surfaces at a later synchronisation, priors/transformer.py:539-560.)
                self._index_guard.submit((lo < 0) | (hi >= table.shape[0]))
                input = input.clamp(0, table.shape[0] - 1)
            elif int(lo) < 0 or int(hi) >= table.shape[0]:      # the read-back is this call's only synchronisation
                raise IndexError("index out of range in self")  # what nn.Embedding raises in the reference
        if table.requires_grad:
            return _train.EmbeddingRowsFn.apply(table, input)
        return table[input]

    def check_indices(self) -> None:
        """Waits for the index checks of the training steps enqueued so far and raises the reference's IndexError if one of
        them saw a symbol outside its embedding table (training loops call it where they read the loss back)."""
        self._index_guard.flush()

    def state_dict(self, *args, **kwargs):
        # a checkpoint must not be written over a pending verdict: the deferred checks are read first
        guard = getattr(self, "_index_guard", None)
        if guard is not None:
            guard.flush()
        return super().state_dict(*args, **kwargs)

    def _get_combined_positional_embeddings(self, kind: Seq2SeqInputKind) -> torch.Tensor:
        if kind == Seq2SeqInputKind.Source:
            freq = self.source_positional_embeddings_frequency.repeat(1, 1, self.source_duration, 1)
            return torch.cat([freq, freq], dim=3)  # relative mode: priors/transformer.py:470-472
        elif kind == Seq2SeqInputKind.Target and self.conditional_model:
            freq = self.target_positional_embeddings_frequency.repeat(1, 1, self.target_duration, 1)
            patch = self.target_positional_embeddings_patch.repeat(1, self.source_frequencies,
                                                                   self.source_duration, 1)
            return torch.cat([freq, patch], dim=3)
        raise ValueError(f"Unexpected value {kind} for kind option")

    @property
    def combined_positional_embeddings_source(self) -> torch.Tensor:
        return self._get_combined_positional_embeddings(Seq2SeqInputKind.Source)

    @property
    def combined_positional_embeddings_target(self) -> torch.Tensor:
        return self._get_combined_positional_embeddings(Seq2SeqInputKind.Target)

    @property
    def causal_mask(self) -> torch.Tensor:
        n = (self.target_transformer_sequence_length_with_start_symbol if self.conditional_model
             else self.source_transformer_sequence_length_with_start_symbol)
        allowed = torch.ones(n, n).tril() == 1
        return torch.zeros(n, n).masked_fill(~allowed, float('-inf'))

    @property
    def identity_memory_mask(self) -> torch.Tensor:
        n = self.source_transformer_sequence_length_with_start_symbol
        return torch.zeros(n, n).masked_fill(torch.eye(n) == 0, float('-inf'))

    # ---------------------------------------------------------------- sequences
    def _helper(self, kind: str) -> CodemapsHelper:
        if kind == 'source':
            return self.source_codemaps_helper
        if kind == 'target':
            return self.target_codemaps_helper
        raise ValueError(f"Unexpected value {kind} for kind option")

    def to_time_frequency_map(self, sequence: torch.Tensor, kind: str,
                              permute_output_as_logits: bool = False) -> torch.Tensor:
        """What the reference's training loop calls on the model (train_autoregressive_model.py:233-234,
        268-269); the arithmetic lives in the codemaps helpers (codemaps_helpers.py:16-243)."""
        return self._helper(kind).to_time_frequency_map(sequence, permute_output_as_logits=permute_output_as_logits)

    def flatten_map(self, codemap: torch.Tensor, kind: str) -> torch.Tensor:
        return self._helper(kind).to_sequence(codemap)

    def to_sequences(self, input: torch.Tensor, condition: Optional[torch.Tensor] = None,
                     class_conditioning: Mapping[str, torch.Tensor] = {},
                     mask: Optional[torch.Tensor] = None,
                     time_indexes_source: Optional[Iterable[int]] = None,
                     time_indexes_target: Optional[Iterable[int]] = None):
        source_sequence = self.source_codemaps_helper.to_sequence(condition)
        mask_sequence = None
        if mask is not None and self.use_inpainting_mask_on_source:
            mask_sequence = self.source_codemaps_helper.to_sequence(mask)
        source_sequence, _ = self.prepare_data(source_sequence, kind=Seq2SeqInputKind.Source,
                                               class_conditioning=class_conditioning, mask=mask_sequence,
                                               time_indexes=time_indexes_source)
        target_sequence = self.target_codemaps_helper.to_sequence(input)
        target_sequence, _ = self.prepare_data(target_sequence, kind=Seq2SeqInputKind.Target,
                                               class_conditioning=class_conditioning,
                                               time_indexes=time_indexes_target)
        return source_sequence, target_sequence

    def prepare_data(self, sequence: torch.Tensor, kind: Seq2SeqInputKind,
                     class_conditioning: Mapping[str, torch.Tensor] = {},
                     mask: Optional[torch.Tensor] = None, time_indexes: Optional[Iterable[int]] = None):
        with torch.set_grad_enabled(self._differentiable()):
            return self._prepare_data(sequence, kind, class_conditioning, mask, time_indexes)

    def _prepare_data(self, sequence, kind, class_conditioning, mask, time_indexes):
        if mask is not None:
            sequence = sequence.masked_fill(mask, self.mask_token_index)
        embedded = self.embed_data(sequence, kind=kind)
        with_positions = self.add_positions_to_sequence(embedded, kind=kind, embedding_dim=2,
                                                        time_indexes=time_indexes)
        if self.positional_class_conditioning:      # :555-560
            with_positions = self.add_class_conditioning_to_sequence(with_positions, class_conditioning)
        prepared = self.add_start_symbol(with_positions, kind=kind, class_conditioning=class_conditioning,
                                         sequence_dim=1)
        return prepared, (0, 2, 1)

    def add_positions_to_sequence(self, sequence: torch.Tensor, kind: Seq2SeqInputKind, embedding_dim: int,
                                  time_indexes: Optional[Iterable[int]]):
        batch_size = sequence.shape[0]
        if kind == Seq2SeqInputKind.Source:
            pos, helper = self.combined_positional_embeddings_source, self.source_codemaps_helper
            F_, T_ = self.source_frequencies, self.source_duration
        elif kind == Seq2SeqInputKind.Target:
            pos, helper = self.combined_positional_embeddings_target, self.target_codemaps_helper
            F_, T_ = self.target_frequencies, self.target_duration
        else:
            raise ValueError(f"Unexpected value {kind} for kind option")
        pos = pos.reshape(1, F_, T_, -1)
        if time_indexes is not None:
            pos = pos[:, :, list(time_indexes), :]
        pos_seq = helper.to_sequence(pos.to(sequence.device)).expand(batch_size, -1, -1)
        return torch.cat([sequence, pos_seq], dim=embedding_dim)

    def add_class_conditioning_to_sequence(self, sequence_with_positions: torch.Tensor,
                                           class_conditioning: Mapping[str, torch.Tensor]) -> torch.Tensor:
        """The class embeddings appended to every row (reference priors/transformer.py:620-638)."""
        emb = torch.zeros(*sequence_with_positions.shape[:2], self.class_conditioning_total_dim,
                          device=sequence_with_positions.device, dtype=sequence_with_positions.dtype)
        for name, cls in class_conditioning.items():
            e = self.class_conditioning_embedding_layers[name].weight[cls]       # [B, 1, dim]
            p0 = self.class_conditioning_start_positions_per_modality[name]
            emb[:, :, p0:p0 + e.shape[2]] = e
        return torch.cat([sequence_with_positions, emb], dim=-1)

    def add_start_symbol(self, sequence_with_positions: torch.Tensor, kind: Seq2SeqInputKind,
                         class_conditioning: Mapping[str, torch.Tensor], sequence_dim: int):
        batch_size = sequence_with_positions.shape[0]
        start = self.source_start_symbol if kind == Seq2SeqInputKind.Source else self.target_start_symbol
        start = start.repeat(batch_size, 1, 1)
        if self.positional_class_conditioning:      # :660-663
            start = self.add_class_conditioning_to_sequence(start, class_conditioning)
            return torch.cat([start, sequence_with_positions], dim=sequence_dim)
        for name, cls in class_conditioning.items():
            emb = self.class_conditioning_embedding_layers[name].weight[cls].squeeze(1)
            p0 = self.class_conditioning_start_positions_per_modality[name]
            start[:, :, p0:p0 + emb.shape[1]] = emb.unsqueeze(1)
        return torch.cat([start, sequence_with_positions], dim=sequence_dim)

    # ---------------------------------------------------------------- forward
    def forward(self, input: torch.Tensor, condition: Optional[torch.Tensor] = None,
                class_condition: Optional[torch.Tensor] = None, memory: Optional[torch.Tensor] = None):
        """input = prepared target sequence [B,S_t,d], condition = prepared source [B,S_s,d]
        (priors/transformer.py:720-795).  Returns (logits [B,S,n_class], memory [S_s,B,d]).
        Differentiable in `.train()` mode (priors/_train.py); inference never records history."""
        with torch.set_grad_enabled(self._differentiable()):
            return self._forward(input, condition, class_condition, memory)

    def _forward(self, input, condition, class_condition, memory):
        if class_condition is not None:
            raise NotImplementedError("local class conditioning is deprecated in the reference")
        tgt = input.transpose(0, 1).contiguous()
        if memory is None:
            src = condition.transpose(0, 1).contiguous()
            memory, *_ = self.transformer.encoder(src, mask='anticausal' if self.self_conditional_model else None)
        memory_mask = self.identity_memory_mask.to(input.device) if self.use_identity_memory_mask else None
        out, *_ = self.transformer.decoder(tgt, memory, tgt_mask='causal', memory_mask=memory_mask)
        start_len = self.target_start_symbol.shape[1]
        out = out[start_len - 1:-1].transpose(0, 1)
        logits = self.project_transformer_outputs_to_logits.run(out.contiguous())
        return logits, memory

    # ---------------------------------------------------------------- persistence
    @classmethod
    def from_parameters_and_weights(cls, parameters_json_path: pathlib.Path,
                                    model_weights_checkpoint_path: pathlib.Path,
                                    device: Union[str, torch.device] = 'cpu') -> 'VQNSynthTransformer':
        with open(parameters_json_path, 'r') as f:
            model = cls(**json.load(f))
        sd = torch.load(model_weights_checkpoint_path, map_location=device)
        if 'model' in sd.keys():
            sd = sd['model']
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        model.load_state_dict(sd)
        return model

    def store_instantiation_parameters(self, path: pathlib.Path) -> None:
        with open(path, 'w') as f:
            json.dump(self._instantiation_parameters, f, indent=4)


class SelfAttentiveVQTransformer(VQNSynthTransformer):
    """Top prior: regenerates masked codes of a map given the rest of the same map
    (priors/transformer.py:832-845)."""

    @property
    def use_inpainting_mask_on_source(self) -> bool:
        return True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.source_codemaps_helper = self.target_codemaps_helper = SimpleCodemapsHelper(
            self.source_frequencies, self.source_duration)


class UpsamplingVQTransformer(VQNSynthTransformer):
    """Bottom prior: zig-zag patch order aligned with the conditioning top map
    (priors/transformer.py:848-872)."""

    @property
    def use_inpainting_mask_on_source(self) -> bool:
        return False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.source_codemaps_helper = SimpleCodemapsHelper(self.source_frequencies, self.source_duration)
        self.target_codemaps_helper = ZigZagCodemapsHelper(
            self.target_frequencies, self.target_duration,
            self.target_frequencies // self.source_frequencies,
            self.target_duration // self.source_duration)
