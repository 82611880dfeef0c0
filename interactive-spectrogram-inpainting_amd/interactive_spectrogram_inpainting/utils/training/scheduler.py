"""Learning-rate schedules of the reference's training scripts
(utils/training/scheduler.py): `CycleScheduler` (train_vqvae.py:781-785,
train_autoregressive_model.py:654-658) and `get_cosine_schedule_with_warmup`
(:659-664).  Host-side only; written as closed forms of the step count and
pinned against LR traces recorded from the reference classes
(tests/golden/schedulers.npz)."""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

from torch.optim import lr_scheduler


def _ramp(kind: str, start: float, end: float, t: float) -> float:
    """Interpolate start -> end at progress t in (0, 1]."""
    if kind == 'linear':
        return start + t * (end - start)
    if kind == 'cos':
        return end + 0.5 * (start - end) * (math.cos(math.pi * t) + 1.0)
    raise KeyError(kind)


class CycleScheduler:
    """One-cycle policy: lr goes lr_max/divider -> lr_max over the first
    `warmup_proportion` of `n_iter` steps, then down to lr_max/divider/1e4; Adam's
    beta1 (or SGD momentum) moves the opposite way between `momentum[0]` and
    `momentum[1]`.  After n_iter steps the cycle restarts.  `step()` returns
    `(lr, momentum)` and writes them into every param group."""

    def __init__(self, optimizer, lr_max: float, n_iter: int, momentum: Optional[Tuple[float, float]] = (0.95, 0.85),
                 divider: float = 25, warmup_proportion: float = 0.3, phase: Sequence[str] = ('linear', 'cos')):
        self.optimizer = optimizer
        self.lr_max, self.n_iter, self.momentum = lr_max, n_iter, momentum
        self.lr_min = lr_max / divider
        self.n_up = int(n_iter * warmup_proportion)
        self.n_down = n_iter - self.n_up
        if self.n_up <= 0 or self.n_down <= 0:
            raise ValueError("CycleScheduler needs at least one warm-up and one annealing step")
        self.kinds = tuple(phase)
        self.count = 0  # steps taken in the current cycle

    def values_at(self, count: int) -> Tuple[float, Optional[float]]:
        """(lr, momentum) set by the count-th step of a cycle, count in 1..n_iter."""
        if count <= self.n_up:
            t, kind = count / self.n_up, self.kinds[0]
            lr = _ramp(kind, self.lr_min, self.lr_max, t)
            mom = _ramp(kind, self.momentum[0], self.momentum[1], t) if self.momentum is not None else None
        else:
            t, kind = (count - self.n_up) / self.n_down, self.kinds[1]
            lr = _ramp(kind, self.lr_max, self.lr_min / 1e4, t)
            mom = _ramp(kind, self.momentum[1], self.momentum[0], t) if self.momentum is not None else None
        return lr, mom

    def step(self):
        self.count += 1
        lr, mom = self.values_at(self.count)
        for group in self.optimizer.param_groups:
            group['lr'] = lr
            if mom is not None:
                if 'betas' in group:
                    group['betas'] = (mom, group['betas'][1])
                else:
                    group['momentum'] = mom
        if self.count >= self.n_iter:
            self.count = 0
        return lr, mom

    def state_dict(self):
        return {'count': self.count}

    def load_state_dict(self, state):
        self.count = int(state['count'])


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                    num_cycles: float = 0.5, last_epoch: int = -1):
    """Linear warm-up 0 -> 1 over `num_warmup_steps`, then cosine decay over the rest
    (`num_cycles` half-periods... 0.5 = down to 0 once), as a LambdaLR factor."""
    warm = max(1, num_warmup_steps)
    rest = max(1, num_training_steps - num_warmup_steps)

    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return step / warm
        return max(0.0, 0.5 * (1.0 + math.cos(2.0 * math.pi * num_cycles * (step - num_warmup_steps) / rest)))

    return lr_scheduler.LambdaLR(optimizer, factor, last_epoch)
