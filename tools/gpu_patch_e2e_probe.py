#!/usr/bin/env python3
"""BASELINE config 4's driver call (100k sphere, 256 patches, diffuse) x 5 - run under rocprofv3 --kernel-trace to see
the timeline of strongest_field_propagation (tools/gpu_reps_probe.py is the same for the representatives driver)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.workloads import fibonacci_patches, sphere_cloud  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
dev = torch.device("cuda:0")
pc = sphere_cloud()
patches = fibonacci_patches(pc)
allp = util.PatchList(torch.cat(patches).to(dev), [len(p) for p in patches], disjoint=True)
filt = [(i, allp[i]) for i in range(len(patches))]
cloud = pc.to(dev)
for _ in range(5):
    pts = cloud.clone()
    fu.strongest_field_propagation(pts, filt, allp, diffuse=True)
    torch.cuda.synchronize()
