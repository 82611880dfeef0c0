"""Training / validation epoch of the transformer prior on MI355X
(reference train_autoregressive_model.py:119-372 `run_model`).

Same loop semantics: per batch of `(top, bottom, class_conditioning)` code maps

  hier == 'top', self-conditional : source = target = top, inpainting mask from `mask_sampler`
                                    applied to the source side              (:178-205)
  hier == 'bottom'                : target = bottom, condition = top         (:218-231)
  logits -> time-frequency map with the class dim on axis 1                  (:233-234)
  loss = criterion(logits_map, target)  (LabelSmoothingLoss)                 (:254)
  backward, optional clip_grad_norm_, optimizer.step(), scheduler.step()     (:256-263)
  accuracy = mean(argmax == target); satisfied-constraints count             (:265-273)
  returns (loss_sum, total_accuracy, num_samples)  sample-weighted           (:372)

What differs by design: the reference wraps the model in `nn.DataParallel` (:145);
here data parallelism is one process per GPU (torch.distributed "nccl" = RCCL) with
`GradBucketReducer` averaging gradients bucket by bucket while the backward is still
running.  Forward and backward of every heavy operator are HIP kernels
(priors/_train.py); plotting / TensorBoard are out of scope.
"""
from __future__ import annotations

import argparse
import time
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from interactive_spectrogram_inpainting.priors.sequence_mask import SequenceMask
from interactive_spectrogram_inpainting.priors.transformer import VQNSynthTransformer
from interactive_spectrogram_inpainting.utils.distributed import GradBucketReducer, is_distributed
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam


def num_satisfied_constraints(predicted: torch.Tensor, condition: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Positions where the prediction keeps the unmasked (given) codes (reference :104-116)."""
    correct = torch.eq(predicted, condition).float()
    return torch.masked_fill(correct, mask, 1).sum()


def run_model(args, epoch: int, loader: Iterable, model: VQNSynthTransformer, optimizer, scheduler, device,
              criterion, tensorboard_writer=None, is_training: bool = True,
              mask_sampler: Optional[SequenceMask] = None, clip_grad_norm: Optional[float] = None,
              reducer: Optional[GradBucketReducer] = None):
    """One epoch.  `args.hier` in {'top', 'bottom'}.  With torch.distributed initialised each rank
    runs its own shard of the data and `reducer` (built once over model.parameters()) averages the
    gradients; the returned sums are this rank's."""
    hier = args.hier
    model.train(is_training)
    loss_sum, total_accuracy, num_samples = 0.0, 0.0, 0
    satisfied_total = 0.0 if model.self_conditional_model else None
    if is_training and reducer is None and is_distributed() and dist.get_world_size() > 1:
        raise RuntimeError("distributed training needs a GradBucketReducer")

    for top, bottom, class_conditioning_tensors in loader:
        if is_training:
            if reducer is not None:
                reducer.zero()
            else:
                optimizer.zero_grad(set_to_none=True)
        class_conditioning_tensors = {k: v.to(device, non_blocking=True).view(-1, 1)
                                      for k, v in class_conditioning_tensors.items()}
        top = top.to(device, non_blocking=True)
        mask = None
        with torch.set_grad_enabled(is_training):
            if hier == 'top':
                if not model.self_conditional_model:
                    raise NotImplementedError("the unconditional top model is not built (see priors/transformer.py)")
                kind, source, target = 'target', top, top
                # the sampler draws in sequence order [B,S]; the wrapper wants the time-frequency layout
                mask = model.to_time_frequency_map(mask_sampler.sample_mask(top.shape[0]).to(device), kind='source')
                source_sequence, target_sequence = model.to_sequences(
                    target, condition=source, class_conditioning=class_conditioning_tensors, mask=mask)
            elif hier == 'bottom':
                kind, target = 'target', bottom.to(device, non_blocking=True)
                source_sequence, target_sequence = model.to_sequences(
                    target, condition=top, class_conditioning=class_conditioning_tensors)
            else:
                raise ValueError(f"unknown hierarchy level {hier}")
            logits_sequence, _ = model(target_sequence, condition=source_sequence)
            logits_map = model.to_time_frequency_map(logits_sequence, kind=kind, permute_output_as_logits=True)
            loss = criterion(logits_map, target)

        if is_training:
            loss.backward()
            if reducer is not None:
                reducer.finish()
            if clip_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm)
            optimizer.step()
            if scheduler is not None:
                scheduler.step()

        with torch.no_grad():
            pred = logits_map.argmax(1)
            accuracy = (pred == target).float().mean()
            batch = top.shape[0]
            if model.self_conditional_model:
                satisfied_total += float(num_satisfied_constraints(pred, source, mask))
            loss_sum += float(loss) * batch          # (joins the device: the step's deferred index verdict has arrived too)
            total_accuracy += float(accuracy) * batch
            num_samples += batch
        if is_training:
            # the training path gathers on clamped indices and defers the range verdict (priors/transformer.py embed_data):
            # raise the reference's IndexError here, where the loss was read back anyway -- before the next optimizer step
            # trains on another clamped symbol
            model.check_indices()
    if tensorboard_writer is not None and not is_training:
        tensorboard_writer.add_scalar(f'code_prediction-validation_{hier}/mean_loss', loss_sum / max(1, num_samples), epoch)
        tensorboard_writer.add_scalar(f'code_prediction-validation_{hier}/mean_accuracy',
                                      total_accuracy / max(1, num_samples), epoch)
    model.check_indices()       # (end of the epoch: nothing pending may outlive it, e.g. into a checkpoint)
    run_model.last_satisfied_constraints = satisfied_total
    return loss_sum, total_accuracy, num_samples


class SyntheticCodes(torch.utils.data.Dataset):
    """Random code maps of the shapes `extract_code.py` writes (top [F_t,T_t], bottom [F_b,T_b])
    with NSynth-like class labels; stands in for the LMDB database (lmdb is not in this image)."""

    def __init__(self, n: int, top_shape, bottom_shape, n_class: int, classes_per_modality, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        self.top = torch.randint(0, n_class, (n, *top_shape), generator=g)
        self.bottom = torch.randint(0, n_class, (n, *bottom_shape), generator=g)
        self.cls = {k: torch.randint(0, v, (n,), generator=g) for k, v in classes_per_modality.items()}

    def __len__(self):
        return self.top.shape[0]

    def __getitem__(self, i):
        return self.top[i], self.bottom[i], {k: v[i] for k, v in self.cls.items()}


def main(argv=None):
    from interactive_spectrogram_inpainting.priors.sequence_mask import UniformProbabilityBernoulliSequenceMask
    from interactive_spectrogram_inpainting.priors.transformer import (SelfAttentiveVQTransformer,
                                                                      UpsamplingVQTransformer)
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    from interactive_spectrogram_inpainting.utils.training.scheduler import CycleScheduler

    ap = argparse.ArgumentParser(description="train the transformer prior on synthetic code maps")
    ap.add_argument('--hier', default='top', choices=['top', 'bottom'])
    ap.add_argument('--batch_size', type=int, default=8)
    ap.add_argument('--num_batches', type=int, default=4)
    ap.add_argument('--num_epochs', type=int, default=1)
    ap.add_argument('--lr', type=float, default=3e-4)
    ap.add_argument('--label_smoothing', type=float, default=0.0)
    ap.add_argument('--clip_grad_norm', type=float, default=None)
    ap.add_argument('--n_class', type=int, default=512)
    ap.add_argument('--top_shape', type=int, nargs=2, default=[32, 32])
    ap.add_argument('--num_encoder_layers', type=int, default=6)
    ap.add_argument('--num_decoder_layers', type=int, default=8)
    ap.add_argument('--database_path', default=None,
                    help="code database written by extract_code.py (LMDB, needs the `lmdb` package); "
                         "default: synthetic code maps")
    args = ap.parse_args(argv)

    distributed = 'RANK' in __import__('os').environ
    if distributed:
        dist.init_process_group('nccl')
        torch.cuda.set_device(int(__import__('os').environ.get('LOCAL_RANK', 0)))
    device = torch.device('cuda', torch.cuda.current_device())
    classes = {'pitch': 61, 'instrument_family_str': 11}
    common = dict(n_class=args.n_class, channel=8, kernel_size=5, n_block=1, n_res_block=1, res_channel=8,
                  use_relative_transformer=True, predict_frequencies_first=True, conditional_model=True,
                  class_conditioning_prepend_to_dummy_input=True,
                  class_conditioning_num_classes_per_modality=classes,
                  class_conditioning_embedding_dim_per_modality={k: 16 for k in classes},
                  conditional_model_num_encoder_layers=args.num_encoder_layers,
                  conditional_model_num_decoder_layers=args.num_decoder_layers)
    top_shape = list(args.top_shape)
    bottom_shape = [2 * top_shape[0], 2 * top_shape[1]]
    torch.manual_seed(2)
    if args.hier == 'top':
        model = SelfAttentiveVQTransformer(shape=top_shape, condition_shape=top_shape, self_conditional_model=True,
                                           add_mask_token_to_symbols=True, **common)
    else:
        model = UpsamplingVQTransformer(shape=bottom_shape, condition_shape=top_shape, **common)
    model = model.to(device)
    if args.database_path is not None:   # train_autoregressive_model.py:330-352 of the reference
        from interactive_spectrogram_inpainting.utils.datasets.lmdb_dataset import LMDBDataset
        data = LMDBDataset(args.database_path, classes_for_conditioning=list(classes))
        sampler_d = torch.utils.data.distributed.DistributedSampler(data) if distributed else None
        loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=sampler_d is None, sampler=sampler_d)
    else:
        data = SyntheticCodes(args.batch_size * args.num_batches, top_shape, bottom_shape, args.n_class, classes,
                              seed=dist.get_rank() if distributed else 0)
        loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=False)
    optimizer = make_adam(model.parameters(), lr=args.lr)
    scheduler = CycleScheduler(optimizer, args.lr, n_iter=len(loader) * args.num_epochs)
    criterion = LabelSmoothingLoss(args.n_class, args.label_smoothing, dim=1)
    reducer = GradBucketReducer(model.parameters()) if distributed else None
    sampler = UniformProbabilityBernoulliSequenceMask(
        low=0.0, high=1.0, sequence_duration=model.source_transformer_sequence_length,
        mask_token_index=model.mask_token_index) if args.hier == 'top' else None
    for epoch in range(args.num_epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        loss_sum, acc_sum, n = run_model(args, epoch, loader, model, optimizer, scheduler, device, criterion,
                                         is_training=True, mask_sampler=sampler, clip_grad_norm=args.clip_grad_norm,
                                         reducer=reducer)
        torch.cuda.synchronize()
        if not distributed or dist.get_rank() == 0:
            print(f"epoch {epoch + 1}: loss {loss_sum / n:.5f} acc {acc_sum / n:.5f} "
                  f"{n / (time.time() - t0):.2f} samples/s/rank")
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
