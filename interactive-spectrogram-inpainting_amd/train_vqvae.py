#!/usr/bin/env python3
"""VQ-VAE training / evaluation loops on MI355X.

Counterpart of the loop semantics of the reference's `train_vqvae.py:133-370`
 -- not of its CLI, dataloaders, tensorboard or audio dumps:

  loss = criterion(out, img) + latent_loss_weight (0.25) * latent_loss.mean()    (:177-179)
  Adam(lr 3e-4) step, optional clip_grad_norm_, optional scheduler step          (:181-192,777)
  per-sample-weighted running means of reconstruction / latent loss / perplexities
  evaluate(): sample-weighted sums, one packed all-reduce over ranks, / world     (:327-365)

Differences by design: metrics are accumulated ON DEVICE and read once per epoch
(the reference calls .item() five times per step), and data parallelism does not
wrap the model in DDP: the model's own backward all-reduces gradient buckets (RCCL)
while it is still running, and the EMA codebook statistics are all-reduced so that
every rank keeps the same codebook (vqvae/_train.py).

Run (one process per GPU):
  python train_vqvae.py --synthetic 64 --batch-size 8 --epochs 1
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_vqvae.py --synthetic 4096 --batch-size 64
"""
from __future__ import annotations

import argparse
import os
import pathlib
import sys
import time
from typing import Callable, Dict, Iterable, Optional, Tuple

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch import nn  # noqa: E402

from interactive_spectrogram_inpainting.utils.distributed import (  # noqa: E402
    DistributedEvalSampler, DistributedTrainSampler, assert_same_step_count, is_distributed, is_master_process)
from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE  # noqa: E402
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam  # noqa: E402


class RunningMeans:
    """Sample-weighted sums kept on the device: no host sync inside the step."""

    NAMES = ("reconstruction_loss", "latent_loss", "perplexity_t", "perplexity_b")

    def __init__(self, device):
        self.sums = torch.zeros(len(self.NAMES), dtype=torch.float64, device=device)
        self.count = 0

    def update(self, batch_size: int, *values: torch.Tensor) -> None:
        vals = torch.stack([v.detach().reshape(-1)[0].double() for v in values])
        self.sums += vals * batch_size
        self.count += batch_size

    def means(self) -> Dict[str, float]:
        m = (self.sums / max(1, self.count)).tolist()
        return dict(zip(self.NAMES, m))


class GraphedVQVAEStep:
    """The loop body of `train` below recorded into HIP graph segments (utils/training/graphed_step.py) and replayed per
    batch: the host's ~200 launches per step become one call per segment (data parallel: segments are cut at the EMA
    messages and the gradient buckets).  The recording's eager warm-up steps would train on the first batch; the model,
    its codebook buffers and the optimizer's moments are therefore put back IN PLACE afterwards (the graph holds their
    addresses), so the replayed epoch follows the eager epoch's trajectory.  Needs a capturable optimizer
    (`make_adam(..., capturable=True)`), no host-side scheduler, batches of one shape."""

    def __init__(self, model: VQVAE, reconstruction_criterion: Callable, optimizer, img: torch.Tensor,
                 latent_loss_weight: float, clip_grad_norm: Optional[float]):
        from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
        self.outs: Dict[str, torch.Tensor] = {}

        def step(x):
            model.zero_grad()
            out, latent_loss, perplexity_t, perplexity_b, *_ = model(x)
            reconstruction_loss = reconstruction_criterion(out, x)
            latent_loss = latent_loss.mean()
            loss = reconstruction_loss + latent_loss_weight * latent_loss
            loss.backward()
            if clip_grad_norm is not None:
                nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm)
            optimizer.step()
            # (detached views of the static result tensors: a kept autograd graph would pin the warm-up steps' gradient
            # accumulators to their stream)
            self.outs = {"reconstruction_loss": reconstruction_loss.detach(), "latent_loss": latent_loss.detach(),
                         "perplexity_t": perplexity_t.detach(), "perplexity_b": perplexity_b.detach()}
            return loss.detach()
        saved = [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())]
        had_state = {id(p) for p in optimizer.state}
        saved_opt = {id(p): {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
                     for p, st in optimizer.state.items()}
        self.graphed = GraphedTrainingStep(step, (img.clone(),), warmup=2)
        with torch.no_grad():
            for t, sv in zip(list(model.parameters()) + list(model.buffers()), saved):
                t.copy_(sv)
            for p_, st in optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(saved_opt[id(p_)][k]) if id(p_) in had_state else v.zero_()
        from interactive_spectrogram_inpainting import _hip
        _hip._on_optimizer_step()      # (the values moved without a version bump: packed-weight caches are stale)
        for q in (model.quantize_t, model.quantize_b):
            if hasattr(q, "_packed_key"):
                q._packed_key = None

    def __call__(self, img: torch.Tensor):
        self.graphed(img)
        return self.outs

    def finish(self) -> None:
        self.graphed.finish()


def train(epoch: int, loader: Iterable, model: VQVAE, reconstruction_criterion: Callable,
          optimizer: torch.optim.Optimizer, scheduler=None, device="cuda", latent_loss_weight: float = 0.25,
          clip_grad_norm: Optional[float] = None, dry_run: bool = False, hip_graph: bool = False) -> Dict[str, float]:
    model.train()
    stats = RunningMeans(device)
    if hasattr(loader, "__len__") and not dry_run:
        assert_same_step_count(len(loader), torch.device(device) if is_distributed() and
                               dist.get_backend() == "nccl" else None)
    if hip_graph:
        if scheduler is not None:
            raise NotImplementedError("a recorded step bakes the learning rate in: no host-side scheduler with hip_graph")
        # the recorded step is kept ACROSS epochs on the model (every train() call used to pay two eager warm-up
        # steps, a re-capture, a new graph pool and a clone / restore of all parameters and optimizer state); it is bound to
        # (optimizer, criterion, batch shape, loss weights): anything else records afresh.  finish() always runs for a step
        # that is dropped, also when the loop raises.
        key = (id(optimizer), id(reconstruction_criterion), latent_loss_weight, clip_grad_norm)
        cached = getattr(model, "_graphed_train_step", None)
        graphed = cached[1] if cached is not None and cached[0][:4] == key else None
        if cached is not None and graphed is None:
            cached[1].finish()
            model._graphed_train_step = None
        ok = False
        try:
            for batch_index, (img, *_) in enumerate(loader):
                img = img.to(device, non_blocking=True)
                if graphed is not None and tuple(img.shape) != cached[0][4]:
                    if batch_index == 0:      # a different batch shape from the first batch on: record afresh
                        graphed.finish()
                        graphed, model._graphed_train_step = None, None
                    else:                     # a ragged last batch (loader without drop_last): one eager step, like the other path
                        model.zero_grad()
                        out, latent_loss, perplexity_t, perplexity_b, *_ = model(img)
                        reconstruction_loss = reconstruction_criterion(out, img)
                        latent_loss = latent_loss.mean()
                        (reconstruction_loss + latent_loss_weight * latent_loss).backward()
                        if clip_grad_norm is not None:
                            nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm)
                        optimizer.step()
                        stats.update(img.shape[0], reconstruction_loss, latent_loss, perplexity_t, perplexity_b)
                        continue
                if graphed is None:
                    graphed = GraphedVQVAEStep(model, reconstruction_criterion, optimizer, img, latent_loss_weight, clip_grad_norm)
                    cached = (key + (tuple(img.shape),), graphed)
                    model._graphed_train_step = cached
                o = graphed(img)
                stats.update(img.shape[0], o["reconstruction_loss"], o["latent_loss"], o["perplexity_t"], o["perplexity_b"])
                if dry_run:
                    break
            ok = True
        finally:
            # finish() waits for the replays, raises a pending range / index verdict and marks the packed-weight caches stale
            # (the replays moved the parameters without touching their version counters): after EVERY epoch -- evaluate() must
            # see the new weights -- and also when the loop raised.  The recording itself stays valid and is replayed by the
            # next epoch; only a loop that raised drops it.
            if graphed is not None:
                graphed.finish()
                if not ok:
                    model._graphed_train_step = None
        return stats.means()
    for batch_index, (img, *_) in enumerate(loader):
        model.zero_grad()
        img = img.to(device, non_blocking=True)
        out, latent_loss, perplexity_t, perplexity_b, *_ = model(img)
        reconstruction_loss = reconstruction_criterion(out, img)
        latent_loss = latent_loss.mean()
        loss = reconstruction_loss + latent_loss_weight * latent_loss
        loss.backward()
        if clip_grad_norm is not None:
            nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        stats.update(img.shape[0], reconstruction_loss, latent_loss, perplexity_t, perplexity_b)
        if dry_run:
            break
    return stats.means()


@torch.no_grad()
def evaluate(loader: Iterable, model: VQVAE, reconstruction_criterion: Callable, device="cuda",
             latent_loss_weight: float = 0.25, dry_run: bool = False) -> Tuple[float, Dict[str, float]]:
    """Sample-weighted averages over the (sharded) validation set; one packed all-reduce
    (sum, then / world size, like train_vqvae.py:342-365)."""
    model.eval()
    stats = RunningMeans(device)
    for img, *_ in loader:
        img = img.to(device, non_blocking=True)
        out, latent_loss, perplexity_t, perplexity_b, *_ = model(img)
        # the reference adds latent_loss.sum() per batch UNWEIGHTED and divides by the number of
        # samples at the end (train_vqvae.py:333,351): reproduce that by pre-dividing by the batch size
        stats.update(img.shape[0], reconstruction_criterion(out, img), latent_loss.sum() / img.shape[0],
                     perplexity_t, perplexity_b)
        if dry_run:
            break
    packed = stats.sums / max(1, stats.count)
    if is_distributed():
        dist.all_reduce(packed)
        packed = packed / dist.get_world_size()
    means = dict(zip(RunningMeans.NAMES, packed.tolist()))
    validation_loss = means["reconstruction_loss"] + latent_loss_weight * means["latent_loss"]
    return validation_loss, means


class SyntheticSpectrograms(torch.utils.data.Dataset):
    """NSynth-shape stand-in: [2, 128, 512] fp32 (log-mel magnitude, mel-IF in [-1,1])."""

    def __init__(self, n: int, shape=(2, 128, 512), seed: int = 0):
        self.n, self.shape, self.seed = n, shape, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        x = torch.randn(*self.shape, generator=g)
        x[1].tanh_()
        return (x,)


def get_reconstruction_criterion(criterion_id: str, spectrograms_helper=None):
    """reference train_vqvae.py:82-98: 'MSE' on the spectrograms, or a multi-scale spectral loss on the audio both
    spectrograms invert to ('Jukebox' / 'DDSP'; needs the helper that defines the inversion)."""
    if criterion_id == 'MSE':
        # `nn.MSELoss()` on the HIP library (same value, fixed summation order; no memset node in a recorded step)
        from interactive_spectrogram_inpainting.utils.losses.mse import MSELoss
        return MSELoss()
    from interactive_spectrogram_inpainting.utils.losses.spectral import (
        DDSPMultiscaleSpectralLoss_fromSpectrogram, JukeboxMultiscaleSpectralLoss_fromSpectrogram)
    if criterion_id in ('Jukebox', 'JukeboxMultiscaleSpectralLoss'):
        assert spectrograms_helper is not None
        return JukeboxMultiscaleSpectralLoss_fromSpectrogram(spectrograms_helper)
    if criterion_id in ('DDSP', 'DDSPMultiscaleSpectralLoss'):
        assert spectrograms_helper is not None
        return DDSPMultiscaleSpectralLoss_fromSpectrogram(spectrograms_helper)
    raise ValueError("Unexpected reconstruction criterion identifier " + criterion_id)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=64, help="number of synthetic spectrograms")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--clip-grad-norm", type=float, default=None)
    ap.add_argument("--latent-loss-weight", type=float, default=0.25)
    ap.add_argument("--reconstruction-criterion", default="MSE",
                    help="MSE | Jukebox | DDSP (spectral losses invert both spectrograms with the mel helper)")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--hip-graph", action="store_true",
                    help="record the training step into HIP graph segments and replay it per batch (GraphedVQVAEStep)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "training runs on MI355X GPUs only (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    torch.manual_seed(1)   # identical initial weights on every rank
    model = VQVAE(in_channel=2).to(device)
    optimizer = make_adam(model.parameters(), lr=args.lr, **({"capturable": True} if args.hip_graph else {}))
    helper = None
    if args.reconstruction_criterion != "MSE":
        from GANsynth_pytorch.spectrograms_helper import SpectrogramsHelper
        # the synthetic spectrograms are [2, 128, 512]: 128 bins <-> n_fft 256, hop 64
        helper = SpectrogramsHelper(16000, 256, 64, 256).to(device)
    criterion = get_reconstruction_criterion(args.reconstruction_criterion, helper)
    data = SyntheticSpectrograms(args.synthetic)
    # training: even shards (every step holds collectives); evaluation: nothing added, nothing dropped
    sampler = DistributedTrainSampler(data, shuffle=True, seed=20200117) if world > 1 else None
    loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, sampler=sampler, shuffle=sampler is None,
                                         drop_last=True, num_workers=0)
    eval_sampler = DistributedEvalSampler(data, shuffle=False) if world > 1 else None
    eval_loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, sampler=eval_sampler, shuffle=False,
                                              drop_last=False, num_workers=0)
    for epoch in range(args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        means = train(epoch, loader, model, criterion, optimizer, device=device,
                      latent_loss_weight=args.latent_loss_weight, clip_grad_norm=args.clip_grad_norm,
                      dry_run=args.dry_run, hip_graph=args.hip_graph)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        val_loss, val = evaluate(eval_loader, model, criterion, device=device,
                                 latent_loss_weight=args.latent_loss_weight, dry_run=args.dry_run)
        if is_master_process():
            print(f"epoch {epoch}: train {means} ({len(loader) * args.batch_size * world / dt:.1f} spectrograms/s) "
                  f"| validation loss {val_loss:.5f} {val}", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
