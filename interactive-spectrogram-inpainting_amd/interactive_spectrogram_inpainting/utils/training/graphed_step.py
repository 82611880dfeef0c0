"""A whole training step -- zero_grad, forward, loss, backward, optimizer -- recorded ONCE into a HIP graph and replayed.

Why: a step of the top prior is ~1300 kernel launches issued through autograd Functions and ctypes; enqueueing them takes the
host 26-30 ms, as long as the GPU needs to run them, so eager steps are bound by whichever of the two is slower on the day
(tools/bench_prior_train.py prints both).  The library allocates nothing and never synchronises, every workspace comes from
torch's caching allocator, so the step records as it stands (`torch.cuda.graph`, i.e. hipStreamBeginCapture on the stream
the library launches on) and a replay costs the host one call.

What a recorded step must not do, and how each case is handled:
  * read anything back -- the index range check of `embed_data` and the lagged weight-range monitor skip themselves while a
    stream is being captured (priors/transformer.py, priors/_ops.py: they ran in the eager warm-up steps; this class checks
    the index range of every batch it is handed instead, asynchronously);
  * draw host random numbers per step -- the fused dropouts' seeds are launch constants of the recording; a device-resident
    counter added to every seed (include/isi_hip.h: isi_set_dropout_seed_base) is advanced by the first node of the graph;
  * take new tensors as inputs -- batches are copied into the static tensors the recording used.
Losses / outputs of EARLIER eager steps must not be alive when the step is recorded (their autograd graphs pin gradient
accumulators to the eager stream, which breaks the capture).  The optimizer must be capturable (`make_adam(..., capturable=True)`).  Caches keyed on parameter versions (packed weights,
embedding tables) are refreshed by launches INSIDE the recording; `finish()` invalidates them for eager code that follows.

The reference trains eagerly under nn.DataParallel (train_autoregressive_model.py:145,441-520); this is the MI355X-side
replacement of its per-step host work, not a change of arithmetic: a replayed step launches exactly the kernels of the eager
step, in the same order (tests/test_prior_train_gpu.py::test_graphed_training_step_equals_eager).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch

from ... import _hip
from ...priors import _ops

_SEED_STRIDE = 0x5851F42D4C957F2D      # odd: the counter visits every residue before it repeats


class GraphedTrainingStep:
    """`step_fn(*static_inputs) -> loss` runs zero_grad / forward / backward / optimizer.step on the static inputs.

    >>> g = GraphedTrainingStep(step_fn, (code, mask), warmup=3)
    >>> loss = g(code_batch, mask_batch)       # copies the batch into the static tensors, replays, returns the loss tensor
    """

    def __init__(self, step_fn: Callable[..., torch.Tensor], static_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 index_limits: Optional[Dict[int, int]] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainingStep needs the GPU (HIP graph capture)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # GradBucketReducer launches its all-reduces from autograd hooks on a side stream: not recorded here
            raise NotImplementedError("GraphedTrainingStep records single-process steps; data-parallel steps run eagerly")
        self.static_inputs = list(static_inputs)
        self.index_limits = dict(index_limits or {})      # input position -> exclusive upper bound of its symbols
        self._pending = []
        dev = self.static_inputs[0].device if self.static_inputs else torch.device("cuda")
        self.seed_base = torch.zeros((), dtype=torch.int64, device=dev)
        _ops.set_dropout_seed_base(self.seed_base)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # eager steps: lazy initialisation, range checks, allocator warm-up
            for _ in range(max(1, warmup)):
                self.seed_base.add_(_SEED_STRIDE)
                step_fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.seed_base.add_(_SEED_STRIDE)
            self.loss = step_fn(*self.static_inputs)
        self.replays = 0

    def __call__(self, *batch: torch.Tensor) -> torch.Tensor:
        if len(batch) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} tensors, got {len(batch)}")
        for i, (dst, src) in enumerate(zip(self.static_inputs, batch)):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("a replayed step takes batches of the recorded shape and dtype")
                dst.copy_(src, non_blocking=True)
            if i in self.index_limits:                      # the check embed_data skipped while recording
                lo, hi = torch.aminmax(dst)
                bad = torch.empty((), dtype=torch.bool, pin_memory=True)
                bad.copy_((lo < 0) | (hi >= self.index_limits[i]), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending.append((bad, ev))
        self.graph.replay()
        self.replays += 1
        while self._pending and self._pending[0][1].query():
            self._raise_if(self._pending.pop(0)[0])
        return self.loss

    def _raise_if(self, flag: torch.Tensor) -> None:
        if bool(flag.item()):
            self._pending.clear()
            raise IndexError("index out of range in self (a batch of an earlier replayed step)")

    def finish(self) -> None:
        """Waits for the replays, raises a pending IndexError, detaches the seed counter and marks every cache derived from
        parameter values stale (the replays changed the parameters without moving their version counters)."""
        torch.cuda.synchronize()
        while self._pending:
            self._raise_if(self._pending.pop(0)[0])
        _ops.set_dropout_seed_base(None)
        _hip._on_optimizer_step()
