"""Data-parallel glue (reference `utils/distributed.py:1-22`): one process per
GPU, RCCL via torch.distributed's "nccl" backend on MI355X, gloo on CPU."""
from __future__ import annotations

import math
from typing import Iterator, Optional, Sized

import torch
import torch.distributed as dist
from torch.utils.data import Sampler

from .training.graphed_step import host_boundary, launch, recording


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def is_master_process() -> bool:
    return not is_distributed() or dist.get_rank() == 0


class DistributedEvalSampler(Sampler[int]):
    """Shards a dataset over ranks without adding or dropping samples
    (rank r gets indices r, r+n, r+2n, ...), unlike DistributedSampler which
    pads to a multiple of the world size.  `shuffle` uses the same
    seed + epoch permutation on every rank."""

    def __init__(self, dataset: Sized, num_replicas: Optional[int] = None, rank: Optional[int] = None,
                 shuffle: bool = True, seed: int = 0):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if is_distributed() else 1
        if rank is None:
            rank = dist.get_rank() if is_distributed() else 0
        if not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.total_size = len(dataset)
        self.num_samples = self.total_size // num_replicas + int(rank < self.total_size % num_replicas)

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.total_size, generator=g).tolist()
        else:
            order = list(range(self.total_size))
        return iter(order[self.rank:self.total_size:self.num_replicas])

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch


class DistributedTrainSampler(DistributedEvalSampler):
    """Shards for TRAINING: every rank gets exactly `total // world` samples of the epoch's (shared) permutation, the
    remainder is dropped for that epoch.  Every training step issues collectives (gradient buckets, EMA statistics),
    so all ranks must run the same number of steps: `DistributedEvalSampler` leaves the first `total % world` ranks
    one sample -- possibly one whole batch -- longer, which deadlocks or mixes mismatched buffers."""

    def __init__(self, dataset: Sized, num_replicas: Optional[int] = None, rank: Optional[int] = None,
                 shuffle: bool = True, seed: int = 0):
        super().__init__(dataset, num_replicas, rank, shuffle, seed)
        self.num_samples = self.total_size // self.num_replicas

    def __iter__(self) -> Iterator[int]:
        return iter(list(super().__iter__())[:self.num_samples])


def assert_same_step_count(n_steps: int, device: Optional[torch.device] = None) -> int:
    """Raises on every rank when the ranks disagree on the number of steps of the coming epoch (each step holds
    collectives); returns the common count.  One small all-reduce per epoch."""
    if not is_distributed():
        return n_steps
    t = torch.tensor([n_steps, -n_steps], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    lo, hi = int(t[0]), -int(t[1])
    if lo != hi:
        raise RuntimeError(f"data-parallel ranks would run between {lo} and {hi} training steps this epoch: use an even "
                           f"sampler (DistributedTrainSampler) for training")
    return n_steps


class GradBucketReducer:
    """Data-parallel gradient averaging for an autograd-driven model (the prior): replaces
    the reference's `nn.DataParallel` (train_autoregressive_model.py:145) with one process
    per GPU.  Parameter gradients live in ONE flat fp32 buffer (each `p.grad` is a view);
    consecutive parameters form buckets, and as soon as autograd has accumulated every
    gradient of a bucket (post-accumulate hooks; the backward reaches the last layers
    first) the bucket is all-reduced asynchronously (RCCL on the GPU, gloo on CPU),
    overlapping the rest of the backward.  `finish()` reduces whatever is left (parameters
    that received no gradient keep zeros), waits, and divides by the world size."""

    def __init__(self, params, bucket_mb: float = 32.0, force_collectives: bool = False):
        self.params = [p for p in params if p.requires_grad]
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.world = dist.get_world_size() if is_distributed() else 1
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        # buckets are filled from the LAST parameter backwards (the order the backward produces them)
        self.buckets = []      # [start, end) element ranges
        self.bucket_of = {}
        off = sum(sizes)
        end, count = off, 0
        for i in range(len(self.params) - 1, -1, -1):
            off -= sizes[i]
            p = self.params[i]
            p.grad = self.flat[off:off + sizes[i]].view_as(p)
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            if end - off >= cap or i == 0:
                self.buckets.append([off, end, count])
                end, count = off, 0
        self.left = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []
        # (tests / single-GPU measurements: run the bucket collectives although there is nobody to exchange with)
        self.force_collectives = bool(force_collectives) and is_distributed()
        if self.world > 1 or self.force_collectives:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def zero(self) -> None:
        """Call instead of `zero_grad()` (which would detach the views)."""
        launch(self.flat.zero_)          # (`launch`: plain call on the GPU; the CPU tests' op-list recording notes it)
        self.left = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        if not recording():
            self.handles = []      # (a recorded step's handle list belongs to its replays' boundary calls)

    def _launch(self, b: int) -> None:
        s, e, _ = self.buckets[b]
        self.launched[b] = True
        bucket = self.flat[s:e]

        def go():
            self.handles.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True))
        # a host call into RCCL: eagerly it happens now (from the gradient hook, while the backward goes on); in a recorded
        # step the graph segment ends here and every replay makes the call between two segments (graphed_step.py)
        host_boundary(go)

    def _on_grad(self, p) -> None:
        b = self.bucket_of[id(p)]
        self.left[b] -= 1
        if self.left[b] == 0 and not self.launched[b]:
            self._launch(b)

    def finish(self) -> None:
        if self.world == 1 and not self.force_collectives:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)

        def wait_all():
            for h in self.handles:
                h.wait()
            self.handles.clear()
        host_boundary(wait_all)
        launch(lambda: self.flat.div_(self.world))


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """Slowest rank's value (used for step timing); identity when not distributed."""
    if not is_distributed():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
