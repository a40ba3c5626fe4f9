#!/usr/bin/env python3
"""Probe behind a fuzz finding (tools/gpu_fuzz.py seed 2025, case 2320: "potential 3621x1679 off by 1.59e-05"): on the fuzz's
clustered clouds with close pairs, how far is the fp32 potential from fp64 - for the HIP path and for the reference-class fp32
torch path (oracle/dipole_oracle.py, the reference's own operation order) - relative to max |phi| and after the 16 u
sum-of-|term| allowance the field checks use?    python tools/gpu_potential_probe.py   (on the GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from oracle import c_oracle, dipole_oracle as O  # noqa: E402
import tools.gpu_fuzz as gf  # noqa: E402

dev = torch.device("cuda:0")
worst = []
for seed in range(60):
    gf.rng = np.random.default_rng(1000 + seed)
    S, T = int(gf.rng.integers(1, 5000)), int(gf.rng.integers(1, 3000))
    src, tgt = gf.cloud(S), gf.cloud(T)
    tgt[:, :3] += 0.37
    ref = c_oracle.potential_f64(src.numpy(), tgt.numpy())
    hip = fu.potential(src.to(dev), tgt.to(dev)).cpu().numpy().astype(np.float64)
    cpu32 = O.potential(src, tgt).numpy().astype(np.float64)
    rr = src.numpy()[None, :, :3].astype(np.float64) - tgt.numpy()[:, None, :3].astype(np.float64)
    dd = np.linalg.norm(rr, axis=-1)
    allow = 16 * 6e-8 * np.where(dd > 0, 1.0 / np.maximum(dd, 1e-300) ** 2, 0).sum(axis=1)
    scale = max(np.abs(ref).max(), 1e-30)
    e_hip, e_cpu = np.abs(hip - ref), np.abs(cpu32 - ref)
    worst.append((e_hip.max() / scale, e_cpu.max() / scale, np.maximum(e_hip - allow, 0).max() / scale, float(dd[dd > 0].min()), S, T))
worst.sort(reverse=True)
print("# max |phi - phi64| / max |phi64|: HIP fp32, reference-class torch fp32 (CPU), HIP beyond the term allowance; closest pair; S x T")
for w in worst[:12]:
    print(f"{w[0]:.2e}  {w[1]:.2e}  {w[2]:.2e}   closest pair {w[3]:.2e}   {w[4]} x {w[5]}")
print(f"# of {len(worst)} cases: HIP above 1e-5: {sum(w[0] > 1e-5 for w in worst)}, torch fp32 above 1e-5: {sum(w[1] > 1e-5 for w in worst)}, "
      f"HIP above 1e-5 beyond the allowance: {sum(w[2] > 1e-5 for w in worst)}")
