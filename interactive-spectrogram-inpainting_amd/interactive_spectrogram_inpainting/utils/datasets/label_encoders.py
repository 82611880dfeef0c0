"""Persistence of the class-conditioning label encoders (reference utils/datasets/label_encoders.py:8-26):
`label_encoders.json` maps every categorical field (pitch, instrument_family_str, ...) to the sorted list of
its classes; the encoders are rebuilt by fitting sklearn LabelEncoders on those lists."""
from __future__ import annotations

import json
import pathlib
from typing import Dict, Mapping

import numpy as np
from sklearn.preprocessing import LabelEncoder

FILENAME = "label_encoders.json"


def dump_label_encoders(label_encoders: Mapping[str, LabelEncoder], savedir_path: pathlib.Path) -> None:
    classes = {field: np.asarray(encoder.classes_).tolist() for field, encoder in label_encoders.items()}
    with open(pathlib.Path(savedir_path) / FILENAME, "w") as f:
        json.dump(classes, f)


def load_label_encoders(path: pathlib.Path) -> Dict[str, LabelEncoder]:
    with open(path, "r") as f:
        classes = json.load(f)
    return {field: LabelEncoder().fit(values) for field, values in classes.items()}
