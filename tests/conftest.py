import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def band_slice(rec, b):
    """Per-band record dict -> the same dict restricted to band b (kept 1-long)."""
    keys = ("eps", "kappa", "calib", "weights", "means", "covars", "rho", "phi", "ups", "ups_inv", "R")
    return {k: np.asarray(rec[k])[b:b + 1] for k in keys}


def unpack_ragged(flat, offs, shapes):
    return [flat[offs[i]:offs[i + 1]].reshape(shapes[i]) for i in range(len(shapes))]
