#!/usr/bin/env python3
"""BASELINE config 4 end to end (fu.strongest_field_propagation, diffuse, on the patch-sorted bench cloud) 12 times - the process
rocprofv3 --kernel-trace --stats is wrapped around to see what the driver's launches cost besides the pair kernel:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c4 -- python3 $REPO/tools/gpu_config4_kernels.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")
pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
ranges = util.PatchList(torch.arange(pts.shape[0], device=dev), sizes, disjoint=True)
for _ in range(12):
    fu.strongest_field_propagation(pts.clone(), list(enumerate(ranges)), ranges, diffuse=True)
torch.cuda.synchronize()
