import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def csr_to_list(off, idx):
    return [torch.from_numpy(idx[off[k]:off[k + 1]].astype(np.int64)) for k in range(len(off) - 1)]


def rel_rowwise(a, b):
    """max over rows of |a - b| / |b| (the tolerance of BASELINE.md: relative to the per-target norm)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b, axis=-1)
    den = np.where(den == 0, 1.0, den)
    return float((np.linalg.norm(a - b, axis=-1) / den).max())


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
