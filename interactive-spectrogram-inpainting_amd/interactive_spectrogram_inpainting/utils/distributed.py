"""Data-parallel glue (reference `utils/distributed.py:1-22`): one process per
GPU, RCCL via torch.distributed's "nccl" backend on MI355X, gloo on CPU."""
from __future__ import annotations

import math
from typing import Iterator, Optional, Sized

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


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


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """Slowest rank's value (used for step timing); identity when not distributed."""
    if not is_distributed():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
