#!/usr/bin/env python3
"""Per-point propagation on ok.xyz (10 000 points) by the number of workgroups of the multi-workgroup form."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import point_driver as ptd  # noqa: E402
g = load_golden("G8_point_propagation")
cloud = torch.from_numpy(g["pc_full"]).cuda()
ref = None
for dtype in (torch.float32, torch.float64):
    for groups in (0, 20, 10, 7, 5, 4, 3):
        ptd.POINT_GREEDY_FORM, ptd.POINT_GREEDY_GROUPS = 2, groups
        ts = []
        for _ in range(5):
            pts = cloud.to(dtype).clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            fu.strongest_field_propagation_points(pts, diffuse=True, starting_point=0); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        order = fu.last_trace("points")["order"]
        order = order.cpu().numpy() if hasattr(order, "cpu") else order
        if dtype == torch.float32 and ref is None: ref = order
        print(f"{str(dtype):14s} groups={groups:3d}  median {np.median(ts) * 1e3:7.2f} ms  min {min(ts) * 1e3:7.2f}  same order as default: {np.array_equal(order, ref) if dtype == torch.float32 else '-'}", flush=True)
