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


# `chosen` = the interaction of the patch picked at each greedy step.  The reference accumulates E in fp32 in visit
# order and sums the patch in fp32; the drivers sum fp64 W entries.  CHOSEN_RTOL = 10 x the worst deviation measured
# over every golden trace (profiles/r03_chosen_deviation.txt: 5.4e-6 on G6's point-scrambled fandisk cloud, where the
# late interactions are cancellation residues; <= 6.3e-7 on every 100 000-point trace), relative to |chosen|.
# Rounds 1-2 used 2e-4 with no measurement behind it.
CHOSEN_RTOL = 6e-5
_chosen_log = []


def check_chosen(got, want, label):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    dev = np.abs(got - want) / np.abs(want)
    _chosen_log.append((label, float(dev.max()), int(dev.argmax()), len(want)))
    assert dev.max() <= CHOSEN_RTOL, (label, float(dev.max()))


def pytest_sessionfinish(session, exitstatus):
    if _chosen_log:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "chosen_deviation.txt"), "w") as f:
                f.write("# max |chosen - golden| / |golden| per golden trace (label, worst, step, steps)\n")
                for row in _chosen_log:
                    f.write("%-40s %.3e  step %d of %d\n" % row)
        except OSError:
            pass
