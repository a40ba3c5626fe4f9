#!/usr/bin/env python3
"""Time field_grad N x N for the library selected by $DNP_LIB (developer tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pc = sphere(n).to("cuda:0")
for _ in range(3):
    E = fu.field_grad(pc, pc)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        E = fu.field_grad(pc, pc)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 5)
print(f"{os.path.basename(os.environ.get('DNP_LIB','libdnp.so')):28s} N={n}: {best*1e3:.3f} ms  {n*n/best/1e9:.1f} Gpairs/s  checksum={float(E.norm(dim=-1).sum()):.6e}", flush=True)
