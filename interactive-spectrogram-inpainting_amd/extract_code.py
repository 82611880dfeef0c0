#!/usr/bin/env python3
"""Code extraction: the VQ-VAE's `forward+quantize` pass over a dataset, sharded
over GPUs, producing one `CodeRow(top, bottom, attributes, filename)` per sample.

Counterpart of `extract` in the reference's `extract_code.py:42-82` (not of its CLI /
GANSynth dataloaders).  The compute is `VQVAE.encode` on MI355X -- the reference runs
the full `forward` and throws the reconstruction away (`:67`); the codes are
identical.  Rows are handed to a `sink(key: str, row: CodeRow)`; `lmdb_sink` writes
the reference's on-disk format (named DB `codes`, key = note name utf-8, value =
pickle(CodeRow), `label_encoders` entry) when the `lmdb` package is available.
Sharding: `DistributedEvalSampler(shuffle=False)` (no sample added or dropped, no
data-path collective).
"""
from __future__ import annotations

import pathlib
import pickle
import sys
from collections import namedtuple
from typing import Callable, Iterable, Mapping, Optional

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

import torch  # noqa: E402

from interactive_spectrogram_inpainting.utils.distributed import is_master_process  # noqa: E402

# the pickled rows name this class by module path: the reference's own (extract_code.py:26-27), so databases written
# by either code base unpickle in the other
from interactive_spectrogram_inpainting.utils.datasets.lmdb_dataset import CodeRow  # noqa: E402


@torch.no_grad()
def extract(loader: Iterable, model, device, sink: Callable[[str, CodeRow], None],
            label_encoders: Mapping[str, object] = {}) -> int:
    """loader yields (sample_batch [B,C,H,W], *categorical_attribute_batches, attributes_batch)
    with attributes_batch['note_str'] the sample names (extract_code.py:62-66)."""
    attribute_names = list(label_encoders.keys())
    model.eval()
    n = 0
    for sample_batch, *categorical_attributes_batch, attributes_batch in loader:
        sample_batch = sample_batch.to(device, non_blocking=True)
        sample_names = attributes_batch['note_str']
        _, _, _, id_t, id_b, _, _ = model.encode(sample_batch)
        id_t = id_t.cpu().numpy()          # D2H of int64 [B,Ht,Wt] / [B,Hb,Wb] (extract_code.py:68-69)
        id_b = id_b.cpu().numpy()
        for top, bottom, *attributes, name in zip(id_t, id_b, *categorical_attributes_batch, sample_names):
            sink(name, CodeRow(top=top, bottom=bottom, attributes=dict(zip(attribute_names, attributes)),
                               filename=name))
            n += 1
    return n


def lmdb_sink(path, label_encoders: Mapping[str, object] = {}, map_size: int = 100 * 1024 ** 3):
    """Sink writing the reference's LMDB layout (extract_code.py:47-79,
    utils/datasets/lmdb_dataset.py:15-89).  Needs the `lmdb` package."""
    try:
        import lmdb
    except ImportError as e:  # pragma: no cover - not installed in the build image
        raise RuntimeError("the `lmdb` package is required to write the reference's code database") from e
    env = lmdb.open(str(path), map_size=map_size, max_dbs=2)
    codes_db = env.open_db('codes'.encode('utf-8'), dupsort=False)
    if is_master_process():
        with env.begin(write=True) as txn:
            txn.put('label_encoders'.encode('utf-8'), pickle.dumps(label_encoders))

    def sink(name: str, row: CodeRow) -> None:
        with env.begin(db=codes_db, write=True) as txn:   # one transaction per sample, like the reference
            txn.put(name.encode('utf-8'), pickle.dumps(row))

    return sink
