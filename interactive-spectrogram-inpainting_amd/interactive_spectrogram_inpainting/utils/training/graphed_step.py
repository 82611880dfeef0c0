"""A whole training step -- zero_grad, forward, loss, backward, optimizer -- recorded ONCE into HIP graphs and replayed.

Why: a step of the top prior is ~1300 kernel launches issued through autograd Functions and ctypes (the VQ-VAE's: ~200);
enqueueing them takes the host 17-30 ms (2-4 ms), as long as or longer than the GPU needs to run them, so eager steps are
bound by whichever of the two is slower on the day -- and with one process per GPU, eight such hosts share one machine.
The library allocates nothing and never synchronises, every workspace comes from torch's caching allocator, so the step
records as it stands (hipStreamBeginCapture on the stream the library launches on) and a replay costs the host one call
per segment.

Data parallelism (round 5).  A collective is a HOST call into RCCL; instead of relying on RCCL's own graph support the
recording is CUT wherever the step reaches one: `host_boundary(fn)` ends the current graph segment, remembers `fn`, and starts
the next segment.  A replay launches segment 0, calls fn_0 (e.g. `dist.all_reduce(bucket, async_op=True)`: RCCL's stream
waits for the segment, the next segment does not wait for RCCL), launches segment 1, ... and the boundary in front of the
optimizer waits for every handle -- exactly the eager step's overlap of bucket all-reduces with the remaining backward
(utils/distributed.py GradBucketReducer, vqvae/_train.py Grads), at ~10 host calls per step instead of ~1300.  Outside a
recording `host_boundary(fn)` just calls `fn()`: the library code is written once.  Boundaries may be reached from autograd's
device thread (gradient hooks, the VQ-VAE's hand-written backward): the capture runs in the relaxed mode, which is what allows
ending it from another thread than the one that began it.

What a recorded step must not do, and how each case is handled:
  * read anything back -- the index range check of `embed_data` and the lagged weight-range monitor skip themselves while a
    stream is being captured (priors/transformer.py, priors/_ops.py: they ran in the eager warm-up steps; this class checks
    the index range of every batch it is handed instead, asynchronously, and re-examines the weights' split-f16 range
    every `range_check_every` replays with the same pinned-flag pattern: a verdict that differs from the recording's raises);
  * draw host random numbers per step -- the fused dropouts' seeds are launch constants of the recording; a device-resident
    counter added to every seed (include/isi_hip.h: isi_set_dropout_seed_base) is advanced by the first node of the graph;
  * take new tensors as inputs -- batches are copied into the static tensors the recording used;
  * run a collective inside a segment -- see above.
Losses / outputs of EARLIER eager steps must not be alive when the step is recorded (their autograd graphs pin gradient
accumulators to the eager stream, which breaks the capture).  The optimizer must be capturable (`make_adam(..., capturable=True)`).
Caches keyed on parameter versions (packed weights, embedding tables, codebooks) are refreshed by launches INSIDE the recording;
`finish()` invalidates them for eager code that follows.

The reference trains eagerly under nn.DataParallel / DistributedDataParallel (train_autoregressive_model.py:145,441-520,
train_vqvae.py:168-192,770-775); this is the MI355X-side replacement of its per-step host work, not a change of arithmetic: a
replayed step launches exactly the kernels of the eager step, in the same order
(tests/test_prior_train_gpu.py::test_graphed_training_step_equals_eager, tests/test_train_gpu.py::test_graphed_vqvae_step...).

The segment / boundary bookkeeping is independent of HIP: `OpListBackend` records explicit host closures instead of stream
captures, which is how the CPU tests run the data-parallel replay logic with gloo (tests/test_host_helpers.py).
"""
from __future__ import annotations

import threading
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch

_SEED_STRIDE = 0x5851F42D4C957F2D      # odd: the counter visits every residue before it repeats

_ACTIVE = None                          # the recording in progress (one per process: a recording owns the device)
_ACTIVE_LOCK = threading.Lock()


def host_boundary(fn: Callable[[], None]) -> None:
    """Runs the host call `fn` (a collective, a wait) at this point of the step.  Eagerly: now.  While a step is being
    recorded: the current graph segment ends here, `fn` is remembered for the replays (it is NOT called during the recording:
    captured kernels have not run, there is nothing real to exchange; every rank records the same sequence, so the collectives
    of the replays still match), and the next segment begins."""
    rec = _ACTIVE
    if rec is None:
        fn()
    else:
        rec.cut(fn)


def recording() -> bool:
    return _ACTIVE is not None


def launch(fn: Callable[[], None]) -> None:
    """Device work as an explicit closure: runs `fn` now -- except while a recording on the op-list backend (CPU tests) is
    active, which only notes it in the current segment, like a stream capture that records a kernel without running it.
    GPU code needs no wrapper (the capture records the library's launches as they are made); the data-parallel helpers use
    it for their few torch calls so that the CPU tests can replay them."""
    rec = _ACTIVE
    if rec is not None and isinstance(rec.backend, OpListBackend):
        rec.backend.record(fn)
        return
    fn()


class HipGraphBackend:
    """Segments = HIP graphs captured back to back on one side stream, sharing one memory pool (tensors of segment k stay
    valid in segment k + 1 because the segments are always replayed in recording order)."""

    def __init__(self, device: torch.device):
        self.device = device
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device)
        self._ctx = None

    def begin(self) -> None:
        import gc
        torch.cuda.synchronize(self.device)
        gc.collect()
        # caches that would otherwise drop dead entries / rebuild host-built tables INSIDE the capture (priors/_ops.py)
        from ...priors import _ops
        _ops._WT_GROUP.prepare_for_capture()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        self._begin_segment()

    def _begin_segment(self) -> None:
        g = torch.cuda.CUDAGraph()
        # relaxed: a boundary reached from autograd's device thread ends the capture from another thread than began it
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        self.graphs.append(g)

    def cut(self) -> None:
        self.graphs[-1].capture_end()
        self._begin_segment()

    def end(self) -> None:
        self.graphs[-1].capture_end()
        self._ctx.__exit__(None, None, None)
        self._ctx = None
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def abort(self) -> None:
        try:
            if self.graphs:
                self.graphs[-1].capture_end()
        except Exception:
            pass
        if self._ctx is not None:
            self._ctx.__exit__(None, None, None)
            self._ctx = None

    def n_segments(self) -> int:
        return len(self.graphs)

    def replay(self, i: int) -> None:
        self.graphs[i].replay()


class OpListBackend:
    """Segments = lists of host closures (`launch(fn)`): the same begin / cut / end / replay protocol without a GPU."""

    def __init__(self):
        self.segments: List[List[Callable[[], None]]] = []

    def begin(self) -> None:
        self.segments.append([])

    def record(self, fn) -> None:
        self.segments[-1].append(fn)

    def cut(self) -> None:
        self.segments.append([])

    def end(self) -> None:
        pass

    def abort(self) -> None:
        pass

    def n_segments(self) -> int:
        return len(self.segments)

    def replay(self, i: int) -> None:
        for fn in self.segments[i]:
            fn()


class SegmentedRecording:
    """begin() ... cut(fn) ... cut(fn) ... end(); replay() = segment 0, fn 0, segment 1, fn 1, ..., last segment."""

    def __init__(self, backend):
        self.backend = backend
        self.between: List[Callable[[], None]] = []

    def __enter__(self):
        global _ACTIVE
        with _ACTIVE_LOCK:
            if _ACTIVE is not None:
                raise RuntimeError("a training step is already being recorded in this process")
            self.backend.begin()
            _ACTIVE = self
        return self

    def cut(self, fn: Callable[[], None]) -> None:
        self.backend.cut()
        self.between.append(fn)

    def __exit__(self, exc_type, exc, tb):
        global _ACTIVE
        with _ACTIVE_LOCK:
            _ACTIVE = None
        if exc_type is None:
            self.backend.end()
        else:
            self.backend.abort()
        return False

    def replay(self) -> None:
        n = self.backend.n_segments()
        for i in range(n):
            self.backend.replay(i)
            if i < n - 1:
                self.between[i]()


class GraphedTrainingStep:
    """`step_fn(*static_inputs) -> loss` runs zero_grad / forward / backward / optimizer.step on the static inputs.

    >>> g = GraphedTrainingStep(step_fn, (code, mask), warmup=3)
    >>> loss = g(code_batch, mask_batch)       # copies the batch into the static tensors, replays, returns the loss tensor

    Data-parallel steps record like single-process ones: their collectives sit behind `host_boundary` (module docstring).
    `range_params`: weights whose split-f16 operand range (|w| < 64, checked against half of it) is re-examined every
    `range_check_every` replays; `backend`: tests only."""

    def __init__(self, step_fn: Callable[..., torch.Tensor], static_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 index_limits: Optional[Dict[int, int]] = None, range_params: Optional[Iterable[torch.Tensor]] = None,
                 range_check_every: int = 256, backend=None):
        self.static_inputs = list(static_inputs)
        self.index_limits = dict(index_limits or {})      # input position -> exclusive upper bound of its symbols
        self._pending = []
        self._range_pending = []
        self.replays = 0
        self.host_ms_last = 0.0
        self.range_check_every = max(1, int(range_check_every))
        self._on_gpu = backend is None
        if self._on_gpu:
            if not torch.cuda.is_available():
                raise RuntimeError("GraphedTrainingStep needs the GPU (HIP graph capture)")
            from ...priors import _ops
            dev = self.static_inputs[0].device if self.static_inputs else torch.device("cuda", torch.cuda.current_device())
            self.device = dev
            self.seed_base = torch.zeros((), dtype=torch.int64, device=dev)
            _ops.set_dropout_seed_base(self.seed_base)
            backend = HipGraphBackend(dev)
        self.range_params = [p for p in (range_params or []) if p.dim() >= 2]
        try:
            if self._on_gpu:
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):                   # eager steps: lazy initialisation, range checks, allocator warm-up
                    for _ in range(max(1, warmup)):
                        self.seed_base.add_(_SEED_STRIDE)
                        step_fn(*self.static_inputs)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
            else:
                for _ in range(max(1, warmup)):
                    step_fn(*self.static_inputs)
            self._range_verdict0 = self._range_verdict().cpu() if self.range_params else None
            self.recording = SegmentedRecording(backend)
            with self.recording:
                if self._on_gpu:
                    self.seed_base.add_(_SEED_STRIDE)
                self.loss = step_fn(*self.static_inputs)
        except BaseException:
            #  a failed warm-up / capture must not leave the process-global seed base registered: later eager
            # dropouts would keep adding a counter nobody advances
            if self._on_gpu:
                from ...priors import _ops
                _ops.set_dropout_seed_base(None)
            raise

    # ------------------------------------------------------------------ checks that cannot live inside a recording
    def _range_verdict(self) -> torch.Tensor:
        """Per weight: is max |w| below HALF the split-f16 operand limit (priors/_ops.py WeightRange)?"""
        from ...priors import _ops
        norms = torch._foreach_norm([p.detach() for p in self.range_params], float("inf"))
        return torch.stack(norms) < 0.5 * _ops._F16_WEIGHT_LIMIT

    def _submit_range_check(self) -> None:
        flag = torch.empty((), dtype=torch.bool, pin_memory=True)
        changed = (self._range_verdict() != self._range_verdict0.to(self.device)).any()
        flag.copy_(changed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._range_pending.append((flag, ev))

    def _raise_if_range(self, flag: torch.Tensor) -> None:
        if bool(flag.item()):
            self._range_pending.clear()
            raise RuntimeError("a weight crossed the split-f16 operand range (|w| >= 32) since this step was recorded: the "
                               "recording's product modes no longer fit it -- record the step again")

    def _raise_if(self, flag: torch.Tensor) -> None:
        if bool(flag.item()):
            self._pending.clear()
            raise IndexError("index out of range in self (a batch of an earlier replayed step)")

    def __call__(self, *batch: torch.Tensor) -> torch.Tensor:
        t0 = time.perf_counter()
        if len(batch) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} tensors, got {len(batch)}")
        for i, (dst, src) in enumerate(zip(self.static_inputs, batch)):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("a replayed step takes batches of the recorded shape and dtype")
                dst.copy_(src, non_blocking=True)
            if i in self.index_limits and self._on_gpu:     # the check embed_data skipped while recording
                lo, hi = torch.aminmax(dst)
                bad = torch.empty((), dtype=torch.bool, pin_memory=True)
                bad.copy_((lo < 0) | (hi >= self.index_limits[i]), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending.append((bad, ev))
        self.recording.replay()
        self.replays += 1
        if self._on_gpu:
            if self.range_params and self.replays % self.range_check_every == 0:
                self._submit_range_check()
            while self._pending and self._pending[0][1].query():
                self._raise_if(self._pending.pop(0)[0])
            while self._range_pending and self._range_pending[0][1].query():
                self._raise_if_range(self._range_pending.pop(0)[0])
        self.host_ms_last = (time.perf_counter() - t0) * 1e3
        return self.loss

    @property
    def n_segments(self) -> int:
        return self.recording.backend.n_segments()

    def finish(self) -> None:
        """Waits for the replays, raises a pending IndexError / range verdict, detaches the seed counter and marks every cache
        derived from parameter values stale (the replays changed the parameters without moving their version counters)."""
        if not self._on_gpu:
            return
        from ... import _hip
        from ...priors import _ops
        torch.cuda.synchronize()
        try:
            while self._pending:
                self._raise_if(self._pending.pop(0)[0])
            if self.range_params:
                self._submit_range_check()
                torch.cuda.synchronize()
            while self._range_pending:
                self._raise_if_range(self._range_pending.pop(0)[0])
        finally:
            _ops.set_dropout_seed_base(None)
            _hip._on_optimizer_step()
