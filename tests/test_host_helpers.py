"""CPU tests of the host-side pieces around the hot path: codemap orderings
(pinned by the reference's own outputs and its only test's round-trip
property), mask samplers, and the 2-process (gloo) sharding logic."""
import numpy as np
import pytest
import torch


def _free_port() -> int:
    """A TCP port nobody listens on right now (asked from the OS): a fixed or seeded-random rendezvous port that happens to be
    taken would leave the spawned ranks waiting for the store's time-out."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]



def test_codemap_orderings_match_reference(golden_dir):
    from interactive_spectrogram_inpainting.priors.codemaps_helpers import (
        SimpleCodemapsHelper, ZigZagCodemapsHelper)
    z = np.load(golden_dir / "codemaps.npz")
    checked = 0
    for key in z.files:
        parts = key.split("_")
        if parts[0] == "simple":
            F, T = map(int, parts[1].split("x"))
            h = SimpleCodemapsHelper(F, T)
            cm = torch.arange(F * T).reshape(1, F, T)
            assert torch.equal(h.to_sequence(cm), torch.from_numpy(z[key])), key
            checked += 1
        elif parts[0] == "zigzag":
            F, T = map(int, parts[1].split("x"))
            pf, pt = map(int, parts[2].split("x"))
            h = ZigZagCodemapsHelper(F, T, pf, pt)
            cm = torch.arange(F * T).reshape(1, F, T)
            seq = h.to_sequence(cm)
            assert torch.equal(seq, torch.from_numpy(z[key])), key
            assert torch.equal(h.to_time_frequency_map(seq), cm)
            checked += 1
        elif parts[0] == "zigzag4d" and parts[1] != "logits":
            F, T = map(int, parts[1].split("x"))
            pf, pt = map(int, parts[2].split("x"))
            h = ZigZagCodemapsHelper(F, T, pf, pt)
            cm4 = torch.arange(2 * F * T * 3).reshape(2, F, T, 3)
            seq4 = h.to_sequence(cm4)
            assert torch.equal(seq4, torch.from_numpy(z[key])), key
            assert torch.equal(h.to_time_frequency_map(seq4), cm4)
            logits = h.to_time_frequency_map(seq4, permute_output_as_logits=True)
            assert torch.equal(logits, torch.from_numpy(z[f"zigzag4d_logits_{parts[1]}_{parts[2]}"]))
            checked += 1
    assert checked >= 12


def test_codemap_properties_of_reference_test():
    """tests/check_relative_transformer.py:59-123 of the reference: round trips on
    arange maps with a trailing embedding dim of 3, and the first
    target_events_per_source_patch entries of the zig-zag order."""
    from interactive_spectrogram_inpainting.priors.codemaps_helpers import (
        SimpleCodemapsHelper, ZigZagCodemapsHelper)
    for cond, tgt in zip([[32, 4], [64, 8], [128, 16]], [[64, 8], [128, 16], [256, 32]]):
        pf, pt = tgt[0] // cond[0], tgt[1] // cond[1]
        src_h, tgt_h = SimpleCodemapsHelper(*cond), ZigZagCodemapsHelper(*tgt, pf, pt)
        for h, (F, T) in ((src_h, cond), (tgt_h, tgt)):
            cm = torch.arange(2 * F * T * 3).reshape(2, F, T, 3)
            assert torch.equal(h.to_time_frequency_map(h.to_sequence(cm)), cm)
        cm = torch.arange(tgt[0] * tgt[1]).reshape(1, *tgt)
        first = tgt_h.to_sequence(cm)[0, :pf * pt]
        # reference check (:106-119): arange(pf)[:,None] + arange(pt)[None,:]*duration, flattened
        expect = (torch.arange(pf)[:, None] * tgt[1] + torch.arange(pt)[None, :]).t().flatten()
        assert torch.equal(first, expect)


def test_sequence_masks():
    from interactive_spectrogram_inpainting.priors import sequence_mask as M
    torch.manual_seed(0)
    m = M.BernoulliSequenceMask(0.3, 1000, 512).sample_mask(8)
    assert m.shape == (8, 1000) and m.dtype == torch.bool and 0.25 < m.float().mean() < 0.35
    m = M.UniformMaskedAmountSequenceMask(0.5, 64, 512).sample_mask(5)
    counts = m.sum(1)
    assert (counts == counts[0]).all() and 32 <= counts[0] <= 64
    x = torch.zeros(3, 64, dtype=torch.int64)
    y = M.UniformProbabilityBernoulliSequenceMask(0.2, 0.9, 64, 512).apply_mask(x)
    assert set(y.unique().tolist()) <= {0, 512}


def _worker(rank, world, port, q):
    import os
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactive_spectrogram_inpainting.utils.distributed import (
        DistributedEvalSampler, is_distributed, is_master_process, max_over_ranks)
    assert is_distributed() and is_master_process() == (rank == 0)
    ds = list(range(11))
    s = DistributedEvalSampler(ds, shuffle=False)
    idx = list(s)
    assert len(idx) == len(s)
    slow = max_over_ranks(1.0 + rank)
    q.put((rank, idx, slow))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_sharding_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = {r: idx for r, idx, _ in out}
    assert sorted(shards[0] + shards[1]) == list(range(11))       # nothing added, nothing dropped
    assert not set(shards[0]) & set(shards[1]) and len(shards[0]) == 6 and len(shards[1]) == 5
    assert all(abs(slow - 2.0) < 1e-12 for _, _, slow in out)        # max over ranks


def _grads_worker(rank, world, port, q):
    import os
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactive_spectrogram_inpainting.vqvae._train import Grads
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3), torch.nn.Linear(3, 2))
    g = Grads(model, n_buckets=3)
    # the backward produces gradients from the last layer to the first
    for p in reversed(list(model.parameters())):
        g.set(p, torch.full_like(p, float(rank + 1)))
    views = g.finish()
    q.put((rank, [float(v.mean()) for v in views], g.n_collectives))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_gloo():
    """Data-parallel gradient exchange of the training step (vqvae/_train.py::Grads):
    buckets are all-reduced as soon as complete and averaged over the 2 ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grads_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, means, n_handles in out:
        assert all(abs(m - 1.5) < 1e-6 for m in means), means      # (1 + 2) / 2 on every rank
        assert n_handles >= 2                                        # more than one bucket was exchanged


def test_lr_schedules_match_reference_traces(golden_dir):
    """CycleScheduler / get_cosine_schedule_with_warmup (utils/training/scheduler.py) against LR and
    beta1 traces recorded from the reference classes (oracle/make_golden.py::scheduler_fixtures)."""
    from interactive_spectrogram_inpainting.utils.training.scheduler import (
        CycleScheduler, get_cosine_schedule_with_warmup)
    z = np.load(golden_dir / "schedulers.npz")
    for tag, kw in (("a", {}), ("b", dict(divider=10, warmup_proportion=0.45)),
                    ("c", dict(momentum=None, phase=("cos", "linear")))):
        n_iter = int(z[f"cycle_{tag}::args"][0])
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = CycleScheduler(opt, 3e-4, n_iter=n_iter, **kw)
        lrs, moms = [], []
        for _ in range(len(z[f"cycle_{tag}::lr"])):
            lr, _mom = sch.step()
            assert lr == opt.param_groups[0]["lr"]
            lrs.append(lr)
            moms.append(opt.param_groups[0]["betas"][0])
        np.testing.assert_allclose(lrs, z[f"cycle_{tag}::lr"], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(moms, z[f"cycle_{tag}::beta1"], rtol=1e-12)
    for tag in "abc":
        warm, total, cycles = z[f"cosine_{tag}::args"]
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
        sch = get_cosine_schedule_with_warmup(opt, int(warm), int(total), num_cycles=float(cycles))
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(len(z[f"cosine_{tag}::lr"]) - 1):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(lrs, z[f"cosine_{tag}::lr"], rtol=1e-12, atol=1e-18)
    # checkpointable
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    a = CycleScheduler(opt, 1e-3, n_iter=10)
    for _ in range(4):
        a.step()
    b = CycleScheduler(opt, 1e-3, n_iter=10)
    b.load_state_dict(a.state_dict())
    assert a.step() == b.step()


def _reducer_worker(rank, world, port, q):
    import os
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactive_spectrogram_inpainting.utils.distributed import GradBucketReducer
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.ones(4))       # a parameter the loss never touches
    params = list(model.parameters()) + [unused]
    red = GradBucketReducer(params, bucket_mb=1e-4)  # ~26 floats per bucket: several buckets
    out = []
    for step in range(2):                            # buffers are reused across steps
        red.zero()
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(6, 5, generator=g)
        model(x).square().sum().backward()
        red.finish()
        out.append([p.grad.flatten().tolist() for p in params])
    q.put((rank, out, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_grad_bucket_reducer_two_process_gloo(world):
    """Autograd-driven DP of the prior: bucketed async all-reduce == mean of the per-rank gradients, at world size 2 and
    at the node's 8 (bucket boundaries at the world size the driver's scaling run uses)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, out, nb = q.get(timeout=240)
        got[rank] = out
        assert nb >= 2
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 2))
    for step in range(2):
        grads = []
        for rank in range(world):
            model.zero_grad()
            g = torch.Generator().manual_seed(10 * step + rank)
            model(torch.randn(6, 5, generator=g)).square().sum().backward()
            grads.append([p.grad.clone() for p in model.parameters()])
        mean = [sum(gs) / world for gs in zip(*grads)]
        for rank in range(world):
            for g_, m in zip(got[rank][step][:-1], mean):
                torch.testing.assert_close(torch.tensor(g_), m.flatten(), rtol=1e-5, atol=1e-6)
            assert got[rank][step][-1] == [0.0] * 4   # untouched parameter: zeros, no hang
        for rank in range(1, world):                  # every rank holds the same bits
            assert got[rank][step] == got[0][step]


def test_spectrogram_specification_is_self_consistent():
    """oracle/spectrogram_oracle.py (parity unpinned): the linear representation inverts to the
    audio it came from, the mel matrix has unit-height triangles, and its approximate inverse
    returns bin-centred energy."""
    import math
    from oracle import spectrogram_oracle as S
    cfg = S.SpecConfig(n_fft=256, hop_length=64, window_length=256)
    t = torch.arange(4000) / 16000.0
    x = (0.5 * torch.sin(2 * math.pi * 440 * t) + 0.3 * torch.sin(2 * math.pi * 1234 * t + 1))[None].double()
    sp = S.to_spectrogram(cfg, x, mel=False)
    assert sp.shape == (1, 2, 128, 63)
    y = S.to_audio(cfg, sp, mel=False)
    assert float((y[:, 300:3700] - x[:, 300:3700]).abs().max()) < 5e-3
    # the synthesis window satisfies the overlap-add identity  sum_k w(n + k hop) w_s(n + k hop) = 1
    w = torch.hann_window(256, periodic=True, dtype=torch.float64)
    ws = S.synthesis_window(cfg, torch.float64)
    acc = sum((w * ws)[k * 64:(k + 1) * 64] for k in range(4))
    assert torch.allclose(acc, torch.ones(64, dtype=torch.float64), atol=1e-12)
    M = S.mel_matrix(cfg)
    assert M.shape == (128, 128) and float(M.max()) <= 1.0 and float(M.min()) >= 0.0
    assert S.instantaneous_frequency(torch.tensor([[0.0, 3.0, -3.0]]))[0, 2] == pytest.approx((2 * math.pi - 6.0) / math.pi)


def test_make_time_indexes_matches_reference(golden_dir):
    """inpainting.make_time_indexes against outputs of the reference's function
    (flask_server.py:670-682; oracle/make_golden.py::time_indexes_fixtures)."""
    import inpainting
    z = np.load(golden_dir / "time_indexes.npz")
    assert len(z["cases"]) >= 20
    for start, codemap_duration, transformer_duration in z["cases"].tolist():
        got = inpainting.make_time_indexes(start, codemap_duration, transformer_duration)
        assert got == z[f"ti::{start}_{codemap_duration}_{transformer_duration}"].tolist()


def test_label_encoders_and_checkpoint_round_trip(tmp_path):
    """utils/datasets/label_encoders.py:8-26 and utils/training/checkpoint.py:6-31."""
    from sklearn.preprocessing import LabelEncoder
    from interactive_spectrogram_inpainting.utils.datasets.label_encoders import dump_label_encoders, load_label_encoders
    from interactive_spectrogram_inpainting.utils.training.checkpoint import Checkpoint
    enc = {"pitch": LabelEncoder().fit(list(range(24, 85))),
           "instrument_family_str": LabelEncoder().fit(["bass", "brass", "flute", "guitar", "keyboard"])}
    dump_label_encoders(enc, tmp_path)
    back = load_label_encoders(tmp_path / "label_encoders.json")
    assert set(back) == set(enc)
    assert back["pitch"].transform([60]).tolist() == enc["pitch"].transform([60]).tolist() == [36]
    assert back["instrument_family_str"].classes_.tolist() == enc["instrument_family_str"].classes_.tolist()
    lin = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(lin.parameters())
    ck = Checkpoint(lin, 4, 0.5, {"perplexity": 12.0}, opt)
    assert list(ck) == ["model", "epoch", "validation_loss", "validation_metrics", "optimizer", "scheduler", "scaler", "use_amp"]
    assert ck["scheduler"] is None and ck["use_amp"] is False and ck["epoch"] == 4
    torch.save(ck, tmp_path / "ck.pt")
    again = torch.load(tmp_path / "ck.pt", weights_only=False)
    assert torch.equal(again["model"]["weight"], lin.weight)


class _FakeLMDB:
    """Dictionary-backed stand-in for the slice of the `lmdb` API the code database uses (the package is not in
    this image): named sub-databases, write / read transactions, sorted-key cursors, stat()."""

    class _Txn:
        def __init__(self, env, db):
            self.env, self.db = env, db

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def _table(self, db=None):
            return self.env.tables[db if db is not None else self.db]

        def put(self, key, value):
            self._table()[bytes(key)] = bytes(value)

        def get(self, key):
            return self._table().get(bytes(key))

        def stat(self, db):
            return {'entries': len(self.env.tables[db])}

        def cursor(self):
            table = self._table()

            class _Cursor:
                def first(self):
                    return bool(table)

                def iternext(self, values=True):
                    return iter(sorted(table)) if not values else iter(sorted(table.items()))
            return _Cursor()

    class _Env:
        def __init__(self):
            self.tables = {None: {}}

        def open_db(self, name, **kw):
            self.tables.setdefault(name, {})
            return name

        def begin(self, db=None, write=False):
            return _FakeLMDB._Txn(self, db)

    def __init__(self):
        self.envs = {}

    def open(self, path, **kw):
        return self.envs.setdefault(path, _FakeLMDB._Env())


def test_code_database_round_trip(tmp_path, monkeypatch):
    """extract_code.lmdb_sink -> utils.datasets.lmdb_dataset.LMDBDataset: the layout of the reference's code
    database (extract_code.py:47-79, lmdb_dataset.py:15-89): db 'codes', key = note name, value = pickled CodeRow,
    `label_encoders` entry, label_encoders.json beside it."""
    import pickle
    import sys
    import types
    from sklearn.preprocessing import LabelEncoder
    fake = _FakeLMDB()
    mod = types.ModuleType("lmdb")
    mod.open = fake.open
    monkeypatch.setitem(sys.modules, "lmdb", mod)
    import extract_code
    from interactive_spectrogram_inpainting.utils.datasets.label_encoders import dump_label_encoders
    from interactive_spectrogram_inpainting.utils.datasets.lmdb_dataset import CodeRow, LMDBDataset
    enc = {"pitch": LabelEncoder().fit(list(range(24, 85))), "instrument_family_str": LabelEncoder().fit(["bass", "flute"])}
    sink = extract_code.lmdb_sink(tmp_path, enc)
    rng = np.random.default_rng(0)
    rows = {}
    for name in ("guitar_001-060-100", "bass_004-030-050", "flute_002-072-127"):
        row = extract_code.CodeRow(top=rng.integers(0, 512, (16, 64)), bottom=rng.integers(0, 512, (32, 128)),
                                   attributes={"pitch": torch.tensor(int(name[-7:-4]) - 24),
                                               "instrument_family_str": torch.tensor(0)}, filename=name)
        sink(name, row)
        rows[name] = row
    env = fake.envs[str(tmp_path)]
    assert set(env.tables[b'codes']) == {n.encode() for n in rows}
    assert set(pickle.loads(env.tables[None][b'label_encoders'])) == set(enc)
    assert isinstance(pickle.loads(env.tables[b'codes'][b'bass_004-030-050']), extract_code.CodeRow)
    dump_label_encoders(enc, tmp_path)
    ds = LMDBDataset(tmp_path, classes_for_conditioning=["pitch"])
    assert len(ds) == 3 and set(ds.label_encoders) == {"pitch"}
    names = sorted(rows)                                       # LMDB iterates keys in byte order
    for i, name in enumerate(names):
        top, bottom, attributes = ds[i]
        assert top.dtype == torch.int64 and torch.equal(top, torch.from_numpy(rows[name].top))
        assert torch.equal(bottom, torch.from_numpy(rows[name].bottom))
        assert list(attributes) == ["pitch"] and attributes["pitch"].shape == (1,)
        assert int(attributes["pitch"]) == int(rows[name].attributes["pitch"])
    assert CodeRow._fields == extract_code.CodeRow._fields == ('top', 'bottom', 'attributes', 'filename')


def test_service_helpers_host_side(tmp_path, monkeypatch):
    """The host halves of the round-6 routes, without a GPU: `/sample-from-dataset`'s lookup over the code database in the
    reference's on-disk layout (flask_server.py:314-372: constraints on decoded attributes, pitch class / octave derived
    from the pitch, cut / continued to the requested duration), the WAV reader behind `/analyze-audio` against the writer
    behind `/get-audio` (PCM16 and float32, stereo averaged), and the PNG writer of `/get-spectrogram-image`."""
    import struct
    import sys
    import types
    import zlib
    from sklearn.preprocessing import LabelEncoder
    fake = _FakeLMDB()
    mod = types.ModuleType("lmdb")
    mod.open = fake.open
    monkeypatch.setitem(sys.modules, "lmdb", mod)
    import extract_code
    import flask_server
    import inpainting
    from interactive_spectrogram_inpainting.utils.datasets.label_encoders import dump_label_encoders
    from interactive_spectrogram_inpainting.utils.datasets.lmdb_dataset import LMDBDataset
    enc = {"pitch": LabelEncoder().fit(list(range(24, 85))), "instrument_family_str": LabelEncoder().fit(["bass", "flute", "guitar"])}
    sink = extract_code.lmdb_sink(tmp_path, enc)
    rng = np.random.default_rng(1)
    notes = {"bass_004-030-050": 3, "flute_002-072-127": 5, "guitar_001-060-100": 4, "guitar_009-072-100": 6}
    for name, width in notes.items():
        family, pitch = name.split("_")[0], int(name[-7:-4])
        sink(name, extract_code.CodeRow(top=rng.integers(0, 512, (8, width)), bottom=rng.integers(0, 512, (16, 2 * width)),
                                        attributes={"pitch": torch.tensor(int(enc["pitch"].transform([pitch])[0])),
                                                    "instrument_family_str": torch.tensor(int(enc["instrument_family_str"].transform([family])[0]))},
                                        filename=name))
    dump_label_encoders(enc, tmp_path)
    ds = LMDBDataset(tmp_path, classes_for_conditioning=["pitch", "instrument_family_str"])
    g = torch.Generator().manual_seed(0)
    (top, bottom), attrs = inpainting.sample_from_database(ds, ds.label_encoders, 4, {"pitch": 72, "instrument_family_str": "guitar"}, g)
    assert attrs["pitch"] == 72 and attrs["instrument_family_str"] == "guitar" and attrs["pitch_class"] == 0 and attrs["octave"] == 6
    assert top.shape == (1, 8, 4) and bottom.shape == (1, 16, 8)               # 6 columns stored: cut
    (top, bottom), attrs = inpainting.sample_from_database(ds, ds.label_encoders, 5, {"octave": 2}, g)   # pitch 30 only: 3 columns
    assert attrs["pitch"] == 30 and top.shape == (1, 8, 5) and bool((top[..., 3:] == top[..., 2:3]).all())
    assert bottom.shape == (1, 16, 10) and bool((bottom[..., 6:] == bottom[..., 5:6]).all())
    seen = {inpainting.sample_from_database(ds, ds.label_encoders, 4, {"pitch_class": 0}, g)[1]["pitch"] for _ in range(40)}
    assert seen == {60, 72}                                                     # a random order over the matching items
    with pytest.raises(LookupError):
        inpainting.sample_from_database(ds, ds.label_encoders, 4, {"pitch": 61}, g)
    t2, b2 = inpainting.resize_codemaps_repeat_last(torch.arange(6).reshape(1, 2, 3), torch.arange(12).reshape(1, 2, 6), 2)
    assert t2.tolist() == [[[0, 1], [3, 4]]] and b2.shape == (1, 2, 4)
    # WAV: what the service writes it reads back (PCM16: to half a quantisation step); float32 stereo is averaged
    x = torch.sin(torch.arange(1000) * 0.03) * 0.7
    y, rate = flask_server._read_wav(flask_server._wav_bytes(x, 16000))
    assert rate == 16000 and y.shape == x.shape and float((y - x).abs().max()) <= 1.6 / 32768     # (written x 32767, read / 32768)
    stereo = torch.stack([x, -0.5 * x], 1).contiguous().numpy().astype("<f4").tobytes()
    wav = (b"RIFF" + struct.pack("<I", 36 + len(stereo)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 2, 22050, 22050 * 8, 8, 32)
           + b"data" + struct.pack("<I", len(stereo)) + stereo)
    y, rate = flask_server._read_wav(wav)
    assert rate == 22050 and torch.allclose(y, 0.25 * x, atol=1e-7)
    for bad in (b"", b"RIFFxxxxWAVE", wav[:20], wav.replace(struct.pack("<IHHIIHH", 16, 3, 2, 22050, 22050 * 8, 8, 32),
                                                            struct.pack("<IHHIIHH", 16, 1, 2, 22050, 22050 * 3, 3, 24))):
        with pytest.raises((ValueError, struct.error)):
            flask_server._read_wav(bad)
    # PNG: signature, header, one IDAT chunk holding filter-0 scanlines of the pixels, CRCs
    rgb = torch.randint(0, 256, (5, 7, 3), dtype=torch.uint8, generator=g)
    png = inpainting._png_bytes(rgb)
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">IIBBBBB", png[16:29]) == (7, 5, 8, 2, 0, 0, 0)
    pos, chunks = 8, []
    while pos < len(png):
        n, tag = struct.unpack(">I", png[pos:pos + 4])[0], png[pos + 4:pos + 8]
        assert struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(png[pos + 4:pos + 8 + n]) & 0xffffffff
        chunks.append((tag, png[pos + 8:pos + 8 + n]))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(5, 1 + 21)
    assert (raw[:, 0] == 0).all() and (raw[:, 1:].reshape(5, 7, 3) == rgb.numpy()).all()


def test_normalizer_statistics_and_spectral_basis_host_side():
    """Host-side constants of the front-end extras, checked without a GPU: DataNormalizer's measured statistics
    against the CPU specification, and the multi-scale loss's windowed DFT basis (window centred inside n_fft,
    hop = ceil((1 - overlap) * window), g = gcd(hop, n_fft) view) against torch.stft frames."""
    import math
    from GANsynth_pytorch.normalizer import DataNormalizer
    from interactive_spectrogram_inpainting.utils.losses.spectral import _Scale
    from oracle import spectrogram_oracle as S
    g = torch.Generator().manual_seed(1)
    batches = [torch.randn(2, 2, 8, 12, generator=g) * 3 - 2 for _ in range(3)]
    st = DataNormalizer._measure([(b, None) for b in batches], 0.8, 1.0)
    ref = S.normalizer_statistics(batches)
    assert all(abs(st[k] - ref[k]) <= 1e-12 * max(1.0, abs(ref[k])) for k in ref)
    y = S.normalize(torch.cat(batches), ref)
    assert abs(float(y[:, 0].max()) - 0.8) < 1e-5 and abs(float(y[:, 0].min()) + 0.8) < 1e-5
    assert abs(float(y[:, 1].max()) - 1.0) < 1e-5 and abs(float(y[:, 1].min()) + 1.0) < 1e-5
    for n_fft, win, overlap in ((64, 64, 0.75), (2048, 1200, 0.80), (256, 200, 0.75)):
        sc = _Scale(n_fft, win, overlap)
        assert sc.hop == math.ceil((1 - overlap) * win) and sc.hop % sc.g == 0 and n_fft % sc.g == 0
        N, F = n_fft, sc.F
        w = torch.zeros(N, dtype=torch.float64)
        left = (N - win) // 2
        w[left:left + win] = torch.hann_window(win, dtype=torch.float64)
        n = torch.arange(N, dtype=torch.float64)
        k = torch.arange(F, dtype=torch.float64)
        ang = 2 * math.pi * k[:, None] * n[None, :] / N
        basis = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], 0)          # what _Scale.build packs
        audio = torch.randn(1, N + 3 * sc.hop, generator=g, dtype=torch.float64)
        X = torch.stft(audio, n_fft=N, hop_length=sc.hop, win_length=win, window=torch.hann_window(win, dtype=torch.float64),
                       center=False, return_complex=True)[0]                        # [F, T]
        assert X.shape[1] == sc.frames(audio.shape[1]) == 4
        for t in range(4):
            frame = audio[0, t * sc.hop:t * sc.hop + N]
            mine = basis @ frame
            assert (mine[:F] - X[:, t].real).abs().max() < 1e-9 and (mine[F:] - X[:, t].imag).abs().max() < 1e-9


def _ema_worker(rank, world, port, q):
    import os
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
    sys.path.insert(0, str(root))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactive_spectrogram_inpainting.utils.distributed import (
        DistributedEvalSampler, DistributedTrainSampler, assert_same_step_count)
    from interactive_spectrogram_inpainting.vqvae._train import exchange_ema_statistics
    from oracle import vqvae_oracle as O
    D, K, N = 8, 16, 40
    g = torch.Generator().manual_seed(5)
    embed = torch.randn(D, K, generator=g)
    z = torch.randn(N, D, generator=g)               # the GLOBAL batch; this rank owns every world-th vector
    mine = z[rank::world]
    ind = O.quantize(mine, embed)[2]
    counts = torch.bincount(ind, minlength=K).float()
    embed_sum = mine.t() @ torch.nn.functional.one_hot(ind, K).float()
    counts, embed_sum = exchange_ema_statistics(counts, embed_sum)
    # bottleneck.py:79-92 applied to the exchanged statistics
    cs = 0.99 * torch.zeros(K) + 0.01 * counts
    ea = 0.99 * embed + 0.01 * embed_sum
    n = cs.sum()
    new_embed = ea / ((cs + 1e-5) / (n + K * 1e-5) * n).unsqueeze(0)
    # samplers: 63 samples, batch 8, 2 ranks
    data = list(range(63))
    steps = {}
    for name, cls in (("eval", DistributedEvalSampler), ("train", DistributedTrainSampler)):
        s = cls(data, shuffle=True, seed=3)
        loader = torch.utils.data.DataLoader(data, batch_size=8, sampler=s, drop_last=True)
        steps[name] = (len(loader), sorted(int(i) for b in loader for i in b))
    uneven_raises = False
    try:
        assert_same_step_count(steps["eval"][0])
    except RuntimeError:
        uneven_raises = True
    same = assert_same_step_count(steps["train"][0])
    # (plain lists: a tensor in a queue travels as a file descriptor served by THIS process -- a rank that has exited by the
    # time the parent unpickles its tuple left the parent with FileNotFoundError)
    q.put((rank, new_embed.tolist(), cs.tolist(), ea.tolist(), steps, uneven_raises, same))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ema_statistics_exchange_and_even_training_shards_gloo(world):
    """SURVEY 8e (config 3): (1) the EMA-statistics all-reduce -- the one place where the DP design deliberately
    differs from the reference's DDP buffer broadcast -- makes `world` ranks x B/world vectors update the codebook exactly
    like one process with B vectors (checked against the oracle's single-process ema_update); (2) training shards are
    even: with 63 samples / batch 8 the eval sampler gives two ranks 4 and 3 steps (a hang: every step holds collectives),
    which `assert_same_step_count` rejects on every rank (at 8 ranks the shards are 8, 8, ..., 7 samples: one step on
    seven ranks, none on the last); the training sampler gives every rank the same count.  World sizes 2 and 8."""
    import torch.multiprocessing as mp
    from oracle import vqvae_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ema_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    D, K, N = 8, 16, 40
    g = torch.Generator().manual_seed(5)
    embed = torch.randn(D, K, generator=g)
    z = torch.randn(N, D, generator=g)
    ind = O.quantize(z, embed)[2]
    ref_embed, ref_cs, ref_ea = O.ema_update(z, ind, embed, torch.zeros(K), embed.clone())
    out.sort(key=lambda t: t[0])
    per_rank = 63 // world // 8 if world == 8 else 3              # drop_last on the even shards: 7 samples -> 0 steps of 8
    seen = []
    for rank, new_embed, cs, ea, steps, uneven_raises, same in out:
        new_embed, cs, ea = torch.tensor(new_embed), torch.tensor(cs), torch.tensor(ea)
        torch.testing.assert_close(cs, ref_cs, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(ea, ref_ea, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new_embed, ref_embed, rtol=1e-5, atol=1e-6)
        if world == 2:
            assert steps["eval"][0] == (4 if rank == 0 else 3)
        else:
            assert steps["eval"][0] == (1 if rank < 7 else 0)
        assert steps["train"][0] == per_rank and same == per_rank
        assert uneven_raises, "ranks with different step counts must be rejected before the first collective"
        seen.append(steps["train"][1])
    assert len({len(x) for x in seen}) == 1
    flat = [i for x in seen for i in x]
    assert len(flat) == len(set(flat)), "training shards overlap"
    for r in range(1, world):
        torch.testing.assert_close(out[0][1], out[r][1], rtol=0, atol=0)      # identical codebooks on every rank


def test_weight_version_counts_optimizer_steps():
    """Packed-weight caches key on `_hip.version_of`: the tensor's version AND a global optimizer-step count (torch's
    fused optimizers update parameters without bumping `tensor._version`).  Host logic only."""
    import torch
    from interactive_spectrogram_inpainting import _hip
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    opt = torch.optim.SGD([p], lr=0.1)
    before = _hip.version_of(p)
    steps = _hip.optimizer_steps()
    opt.step()
    after = _hip.version_of(p)
    assert after != before and _hip.optimizer_steps() == steps + 1
    with torch.no_grad():
        p.add_(1.0)                        # an in-place update outside any optimizer: the version counter moves
    assert _hip.version_of(p) != after and _hip.optimizer_steps() == steps + 1


def test_weight_range_monitor_and_optimizer_helper():
    """`WeightRange` (split-f16 operand range of a weight without a device read-back per step) and `make_adam`."""
    import torch
    from interactive_spectrogram_inpainting.priors._ops import WeightRange
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    w = torch.nn.Parameter(torch.full((4, 32), 0.5))
    mon = WeightRange()
    calls = []
    real_abs = torch.Tensor.abs

    def spy(self, *a, **k):
        calls.append(1)
        return real_abs(self, *a, **k)
    torch.Tensor.abs = spy
    try:
        assert mon.update(w, False) is True and len(calls) == 1
        opt = make_adam([w], lr=1e-3)
        assert isinstance(opt, torch.optim.Adam)
        for _ in range(5):                                  # training: no further checks for ~256 versions
            w.grad = torch.ones_like(w)
            opt.step()
            assert mon.update(w, False) is True
        assert len(calls) == 1
        with torch.no_grad():
            w[0, 0] = 40.0                                   # inside the range, beyond HALF of it
        assert mon.update(w, True) is True                   # inference weights: exact limit, checked now
        assert WeightRange().update(w, False) is False       # a weight under training: half the limit
        with torch.no_grad():
            w[0, 0] = 70.0
        assert mon.update(w, True) is False
    finally:
        torch.Tensor.abs = real_abs


def _segmented_worker(rank, world, port, q):
    import os
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactive_spectrogram_inpainting.utils.distributed import GradBucketReducer
    from interactive_spectrogram_inpainting.utils.training.graphed_step import (
        GraphedTrainingStep, OpListBackend, host_boundary, launch, recording)
    from interactive_spectrogram_inpainting.vqvae._train import PendingEma
    torch.manual_seed(0)
    sizes = [6, 10, 4, 12, 8]
    params = [torch.nn.Parameter(torch.randn(n)) for n in sizes]
    red = GradBucketReducer(params, bucket_mb=1e-4)       # ~26 floats per bucket: 2+ buckets
    x = torch.zeros(3)
    loss = torch.zeros(())
    lr = 0.1
    trace = []                                            # host calls of the replays, in order

    # a second kind of boundary inside the "forward": the asynchronous EMA-statistics exchange of the VQ-VAE step
    class _Q:
        decay, eps = 0.99, 1e-5

    class _Ema(PendingEma):
        updates = []

        @staticmethod
        def _update(qz, packed, D, K):
            launch(lambda: _Ema.updates.append(packed.clone()))
    stats_c, stats_s = torch.zeros(4), torch.zeros(2, 4)

    def step_fn(xb):
        red.zero()
        pend = _Ema()
        launch(lambda: (stats_c.copy_(xb.sum() + torch.arange(4.0)), stats_s.copy_(xb[:2, None] * torch.ones(2, 4))))
        packed_holder = []
        # (PendingEma.submit concatenates eagerly; here the concatenation is device work of the recorded step)
        packed = torch.zeros(4 + 8)
        launch(lambda: packed.copy_(torch.cat([stats_c, stats_s.reshape(-1)])))
        pend.items.append((_Q(), packed, 2, 4))

        def go():
            trace.append("ema-allreduce")
            pend.handles.append(dist.all_reduce(packed, async_op=True))
        host_boundary(go)
        for i in reversed(range(len(params))):            # the "backward": last parameters first
            p = params[i]
            launch(lambda p=p, i=i: p.grad.copy_(p.detach() * xb.mean() + xb.sum() * (i + 1)))
            red._on_grad(p)
            if i == 2:
                pend.flush()                              # wait + codebook write somewhere inside the step
        red.finish()
        launch(lambda: [p.data.sub_(lr * p.grad) for p in params])
        launch(lambda: loss.copy_(sum(p.detach().sum() for p in params)))
        return loss

    batches = [torch.tensor([1.0, 2.0, 3.0]) * (1 + rank) + k for k in range(4)]
    g = GraphedTrainingStep(step_fn, (x,), warmup=1, backend=OpListBackend())    # warm-up = one eager step on zeros
    assert not recording()
    losses = [float(g(b)) for b in batches]
    # (plain lists: a tensor travels through the queue as a shared-memory handle the parent must fetch while this
    # process is still alive)
    q.put((rank, [p.detach().tolist() for p in params], losses, g.n_segments, len(red.buckets),
           [u.tolist() for u in _Ema.updates]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_segmented_graph_replay_two_process_gloo(world):
    """data-parallel steps under graph replay.  The recording is cut at every host call into the
    collective library (`host_boundary`); a replay runs segment, host call, segment, ...  Checked with the op-list
    backend (segments = lists of closures instead of HIP graphs) and gloo, world sizes 2 and 8, through the real
    GradBucketReducer and PendingEma: after one eager warm-up step and four replays on different batches both ranks hold
    the parameters of a single process that averages the two ranks' gradients, the EMA statistics every rank applied are
    the sum over ranks, and the number of segments is the number of boundaries + 1."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_segmented_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, params, losses, nseg, nb, ema = q.get(timeout=240)
        got[rank] = ([torch.tensor(p_) for p_ in params], losses, nseg, nb, [torch.tensor(u_) for u_ in ema])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: mean of the two ranks' gradients, warm-up step on zeros first
    torch.manual_seed(0)
    sizes = [6, 10, 4, 12, 8]
    ref = [torch.randn(n) for n in sizes]
    xs = [[torch.zeros(3)] * world] + [[torch.tensor([1.0, 2.0, 3.0]) * (1 + r) + k for r in range(world)] for k in range(4)]
    ref_losses, ref_ema = [], []
    for xr in xs:
        grads = [sum(p * x.mean() + x.sum() * (i + 1) for x in xr) / world for i, p in enumerate(ref)]
        ref = [p - 0.1 * g_ for p, g_ in zip(ref, grads)]
        ref_losses.append(float(sum(p.sum() for p in ref)))
        ref_ema.append(sum(torch.cat([x.sum() + torch.arange(4.0), (x[:2, None] * torch.ones(2, 4)).reshape(-1)]) for x in xr))
    for rank in range(world):
        params, losses, nseg, nb, ema = got[rank]
        assert nb >= 2
        assert nseg == 1 + 1 + nb + 1 + 1, (nseg, nb)      # EMA exchange, nb buckets, EMA wait, gradient wait
        for a, b in zip(params, ref):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        for a, b in zip(losses, ref_losses[1:]):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
        assert len(ema) == 5
        for a, b in zip(ema, ref_ema):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    for r in range(1, world):
        for a, b in zip(got[0][0], got[r][0]):
            assert torch.equal(a, b)
