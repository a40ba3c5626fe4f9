#!/usr/bin/env python3
"""fp32 field_grad against the fp64 kernel on clouds that go through the scalar-unit kernel (>= 5e8 pairs): a
random-dipole cloud (cancellation residues: the hardest case for a relative bound) and the boxunion cloud.
Run once per library build (DNP_LIB=... selects one) to compare accumulation settings."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
def random_dipoles(n):
    x = torch.randn(n, 3, generator=g); x = x / x.norm(dim=-1, keepdim=True) * 0.5
    d = torch.randn(n, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
    return torch.cat([x, d], 1)
cases = {"random-dipole shell 25000": random_dipoles(25000), "random-dipole shell 40000": random_dipoles(40000),
         "boxunion 100000": torch.from_numpy(load_golden("G15_boxunion_config3")["pc"])}
print("library:", os.environ.get("DNP_LIB", "product"))
for name, pc in cases.items():
    pc = pc.to(dev)
    E32 = fu.field_grad(pc, pc).double()
    E64 = fu.field_grad(pc.double(), pc.double())
    rel = ((E32 - E64).norm(dim=1) / E64.norm(dim=1)).cpu().numpy()
    print(f"{name:28s} median {np.median(rel):.2e}  p99 {np.quantile(rel, 0.99):.2e}  max {rel.max():.2e}  rows > 5e-6: {(rel > 5e-6).sum()}  > 1e-5: {(rel > 1e-5).sum()}", flush=True)
