"""Reader of the code database `extract_code.extract` writes (reference utils/datasets/lmdb_dataset.py:15-89):
an LMDB environment with the named database `codes` (key = note name, utf-8; value = pickle of
`CodeRow(top, bottom, attributes, filename)`, codes as numpy int64 maps) and `label_encoders.json` beside it.
Items are `(top [F_t, T_t] int64, bottom [F_b, T_b] int64, {class name: tensor [1]})`, the samples the prior's
training loop consumes (train_autoregressive_model.py).  The `lmdb` package is imported on first use (this
build image does not ship it; tests run the logic against a minimal in-memory stand-in)."""
from __future__ import annotations

import pathlib
import pickle
from collections import OrderedDict, namedtuple
from typing import Mapping, Sequence, Union

import torch
from torch.utils.data import Dataset

from .label_encoders import load_label_encoders

CodeRow = namedtuple('CodeRow', ['top', 'bottom', 'attributes', 'filename'])


class LMDBDataset(Dataset):
    def __init__(self, path: Union[str, pathlib.Path], classes_for_conditioning: Sequence[str] = (),
                 dataset_db_name: str = 'codes'):
        try:
            import lmdb
        except ImportError as e:
            raise RuntimeError("the `lmdb` package is required to read the reference's code database") from e
        self.env = lmdb.open(str(path), max_readers=32, lock=False, readahead=False, meminit=False, max_dbs=2,
                             map_size=100 * 1024 ** 3)
        if not self.env:
            raise IOError('Cannot open lmdb dataset', path)
        self.dataset_db = self.env.open_db(dataset_db_name.encode('utf-8'))
        with self.env.begin(db=self.dataset_db) as txn:      # index -> key, in the database's key order
            cursor = txn.cursor()
            cursor.first()
            self._keys = [key for key in cursor.iternext(values=False)]
        self.classes_for_conditioning = list(classes_for_conditioning or [])
        self.label_encoders: Mapping[str, object] = {}
        if self.classes_for_conditioning:
            encoders = load_label_encoders(pathlib.Path(path) / 'label_encoders.json')
            self.label_encoders = {name: enc for name, enc in encoders.items() if name in self.classes_for_conditioning}

    def __len__(self) -> int:
        with self.env.begin() as txn:
            return txn.stat(self.dataset_db)['entries']

    def __getitem__(self, index: int):
        with self.env.begin(db=self.dataset_db, write=False) as txn:
            row = pickle.loads(txn.get(self._keys[index]))
        attributes = OrderedDict((name, torch.as_tensor(row.attributes[name]).view(1))
                                 for name in self.classes_for_conditioning)
        return torch.from_numpy(row.top), torch.from_numpy(row.bottom), attributes
