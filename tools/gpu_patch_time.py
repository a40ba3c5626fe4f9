#!/usr/bin/env python3
"""Time the per-patch slab launch (bench workload) for the library selected by $DNP_LIB (developer tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu, util
from tools.workloads import sphere_cloud, fibonacci_patches
dev = torch.device("cuda:0")
pc = sphere_cloud(); patches = fibonacci_patches(pc)
off, idx, _ = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
pp = torch.repeat_interleave(torch.arange(256, device=dev), off[1:] - off[:-1])
for _ in range(3): fu._patch_slabs(pts, off, None, pp, 0, 256, 1e-5)
ts = []
for _ in range(15):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dE = fu._patch_slabs(pts, off, None, pp, 0, 256, 1e-5); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"{os.path.basename(os.environ.get('DNP_LIB','libdnp.so')):26s} slabs: median {np.median(ts):.3f} ms min {min(ts):.3f}  checksum {float(dE.double().abs().sum()):.6e}", flush=True)
