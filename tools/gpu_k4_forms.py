#!/usr/bin/env python3
"""Per-point kernel: single-workgroup form against one-workgroup-per-CU form by cloud size (fp32 and fp64)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import point_driver as ptd  # noqa: E402
dev = torch.device("cuda:0")
ok = torch.from_numpy(load_golden("G8_point_propagation")["pc_full"])
for dtype in (torch.float32, torch.float64):
    for n in (256, 512, 768, 1024, 1280, 1536, 1792, 2048, 4096):
        row = []
        for form in (1, 2):
            ptd.POINT_GREEDY_FORM = form
            pc = ok[:n].to(dtype)
            fu.strongest_field_propagation_points(pc.clone().to(dev), diffuse=True); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                a = pc.clone().to(dev); torch.cuda.synchronize(); t0 = time.perf_counter()
                fu.strongest_field_propagation_points(a, diffuse=True); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            row.append(min(ts) / n * 1e6)
        print(f"{str(dtype):14s} N={n:5d}: single workgroup {row[0]:6.2f} us/step | per-CU workgroups {row[1]:6.2f} us/step", flush=True)
