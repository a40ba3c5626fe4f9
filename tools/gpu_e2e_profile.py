#!/usr/bin/env python3
"""cProfile of the orient_pointcloud / orient_large counterparts on the 100k sphere file (second run of each, synchronised):
where the ~50 ms of a whole file-to-file call go.    python tools/gpu_e2e_profile.py   (on the GPU box)"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import options, orient_large, orient_pointcloud, util  # noqa: E402
from tools.workloads import sphere_cloud  # noqa: E402

pc = sphere_cloud(100_000, 1234)
flip = torch.rand(pc.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
pc[flip, 3:] *= -1
d = Path(tempfile.mkdtemp())
util.export_pc(pc.transpose(0, 1), d / "s.xyz")
for mod, name in ((orient_pointcloud, "orient_pointcloud"), (orient_large, "orient_large")):
    o = options.get_parser().parse_args(["--pc", str(d / "s.xyz"), "--export_dir", str(d / name), "--number_parts", "41",
                                         "--minimum_points_per_patch", "100", "--diffuse", "--iters", "1"])
    o.export_dir.mkdir(exist_ok=True, parents=True)
    mod.run(o)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        mod.run(o)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"== {name}: {min(ts) * 1e3:.1f} ms min / {sorted(ts)[2] * 1e3:.1f} ms median of 5 file-to-file calls")
    pr = cProfile.Profile()
    pr.enable()
    mod.run(o)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
