#!/usr/bin/env python3
"""Scale check beyond the headline size: strongest_field_propagation (diffuse) on a 1 000 000-point sphere split into
2048 Fibonacci patches with whole patches sign-scrambled - 10^12 pair evaluations, a 24.6 GB slab set.  Prints the
time and checks that the propagation leaves one consistent orientation."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu, util
from tools.gpu_check import sphere

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
pc = sphere(N).to(dev)
k = torch.arange(P, dtype=torch.float64) + 0.5
phi, theta = torch.acos(1 - 2 * k / P), np.pi * (1 + 5 ** 0.5) * k
c = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], 1).float().to(dev)
lab = torch.cat([(pc[i:i + 65536, 3:6] @ c.T).argmax(dim=1) for i in range(0, N, 65536)])
order = torch.argsort(lab, stable=True)
sizes = torch.bincount(lab, minlength=P).cpu().numpy()
patches = util.PatchList(order, sizes, disjoint=True)
scr = (torch.rand(P, generator=torch.Generator().manual_seed(0)) < 0.5).to(dev)
truth = pc[:, 3:].clone()
pc[scr[lab], 3:] *= -1
torch.cuda.synchronize()
print(f"N={N} P={P} patch sizes {sizes.min()}..{sizes.max()}, {int(scr.sum())} patches scrambled", flush=True)
for rep in range(2):
    pts = pc.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fu.strongest_field_propagation(pts, [(i, patches[i]) for i in range(P)], patches, diffuse=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    agree = ((pts[:, 3:] * truth).sum(-1) > 0).float().mean().item()
    pairs = float((sizes * (N - sizes)).sum())
    print(f"run {rep}: {dt * 1e3:.1f} ms  ({pairs / dt / 1e12:.3f} Tpairs/s), fraction aligned with the outward normals {agree:.6f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    assert agree in (0.0, 1.0) or min(agree, 1 - agree) < 1e-4
