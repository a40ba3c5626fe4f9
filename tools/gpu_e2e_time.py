#!/usr/bin/env python3
"""Stage timing of the orient_large / orient_pointcloud counterparts on the 100k sphere (developer tool)."""
import os, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dipole_normal_prop_amd import options, orient_large, orient_pointcloud, util
from test_oracle_golden import sphere100k
pc = sphere100k()
flip = torch.rand(pc.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
pc[flip, 3:] *= -1
d = Path(tempfile.mkdtemp())
t0 = time.perf_counter(); util.export_pc(pc.transpose(0, 1), d / "s.xyz"); print(f"export_pc 100k: {time.perf_counter()-t0:.2f}s")
for mod, name in ((orient_large, "orient_large"), (orient_pointcloud, "orient_pointcloud")):
    o = options.get_parser().parse_args(["--pc", str(d / "s.xyz"), "--export_dir", str(d / name), "--number_parts", "41",
                                         "--minimum_points_per_patch", "100", "--diffuse", "--iters", "1"])
    o.export_dir.mkdir(exist_ok=True, parents=True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = mod.run(o); torch.cuda.synchronize()
        print(f"== {name} run {rep}: total {time.perf_counter()-t0:.2f}s, outward {float(((out[:, 3:]*out[:, :3]).sum(-1) > 0).float().mean()):.4f}", flush=True)
