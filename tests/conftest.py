"""pytest configuration: registers the `gpu` marker and puts the repo root
(for `oracle`) and the product package directory on sys.path."""
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG_DIR = ROOT / "interactive-spectrogram-inpainting_amd"
for p in (str(ROOT), str(PKG_DIR), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
