#!/usr/bin/env python3
"""End-to-end timing of the patch drivers on the 100k sphere / fandisk (developer tool)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import patch_drivers as pd  # noqa: E402
from tools.workloads import sphere_cloud, fibonacci_patches
dev = torch.device("cuda:0")
pc = sphere_cloud()
patches = [p.to(dev) for p in fibonacci_patches(pc)]
gen = torch.Generator().manual_seed(0)
flip = torch.rand(256, generator=gen) < 0.5
work0 = pc.clone().to(dev)
for k, p in enumerate(patches):
    if flip[k]: work0[p, 3:] *= -1
for mode, diffuse in (("batched", True), ("batched", False), ("sequential", True)):
    pd.PATCH_MODE = mode
    w = work0.clone(); fu.strongest_field_propagation(w, list(enumerate(patches)), patches, diffuse=diffuse); torch.cuda.synchronize()
    w = work0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    fu.strongest_field_propagation(w, list(enumerate(patches)), patches, diffuse=diffuse); torch.cuda.synchronize()
    print(f"100k sphere, 256 patches, {mode}, diffuse={diffuse}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
pd.PATCH_MODE = "batched"
pr = cProfile.Profile(); w = work0.clone(); pr.enable()
fu.strongest_field_propagation(w, list(enumerate(patches)), patches, diffuse=True); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
