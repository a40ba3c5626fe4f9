#!/usr/bin/env python3
"""The representatives driver's final field (config 3: 93 411 representatives -> 6589 rest points of boxunion) alone: sources in
point order (what the driver passes) against sources in patch order (spatially coherent chunks: the far-field chain can fire), under
the product library and under build variants (DNP_LIB) that lower the far chain's pair threshold.
    python tools/gpu_rest_field_ab.py   (tools/gpu_rest_field_ab.sh runs it per variant -> profiles/r05_rest_field_ab.txt)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from dipole_normal_prop_amd._staging import _pairs_into  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "G15_boxunion_config3.npz"))
work = torch.from_numpy(g["pc"]).to(dev)
rep_idx = torch.from_numpy(g["rep_idx"].astype(np.int64)).to(dev)
rest = torch.from_numpy(g["rest_idx"].astype(np.int64)).to(dev)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


ref = fu.field_grad(work[rep_idx.sort().values].double(), work[rest][:, :3].double())
scale = ref.norm(dim=1).max().item()
for name, order in (("point order", rep_idx.sort().values), ("patch order", rep_idx)):
    src = work[order].contiguous()
    E2 = torch.empty((rest.shape[0], 3), dtype=work.dtype, device=dev)
    t = timed(lambda: _pairs_into("field", src, None, work, rest, 1e-5, 15000, E2))
    err = ((E2.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-3 * scale)).max().item()
    print(f"{name}: {t:7.1f} us per call (pair launch + second pass), worst row error vs the fp64 kernels {err:.2e}", flush=True)
