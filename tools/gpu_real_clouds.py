#!/usr/bin/env python3
"""Robustness run on clouds without normals (scratch_data/*.xyz, not part of the repository): the orient_large flow
with the flags of the reference's demos/lion.sh (PCA normals, 41^3 voxels, min 100 points, <= 500 representatives
per patch) - time per stage and the local consistency of the result (share of 8-nearest-neighbour pairs whose
normals agree in sign), before and after the propagation."""
import glob, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import options, pipeline, util  # noqa: E402


def consistency(cloud, sample=20000, k=8):
    g = torch.Generator().manual_seed(0)
    pick = torch.randperm(cloud.shape[0], generator=g)[:sample].to(cloud.device)
    d2 = torch.cdist(cloud[pick, :3], cloud[:, :3])
    nn = d2.topk(k + 1, dim=1, largest=False).indices[:, 1:]
    dots = (cloud[pick, None, 3:] * cloud[nn, 3:]).sum(-1)
    return float((dots > 0).float().mean())


for path in sorted(glob.glob(os.path.join(ROOT, "scratch_data", "*.xyz"))):
    out = os.path.join(ROOT, "gpurun_out", "real_" + os.path.basename(path)[:-4])
    argv = ["--pc", path, "--export_dir", out, "--estimate_normals", "--n", "50", "--number_parts", "41",
            "--minimum_points_per_patch", "100", "--diffuse"]
    opts = options.get_parser().parse_args(argv)
    raw = util.estimate_normals(util.load_xyz(path, append_normals=False).cuda(), 50)
    before = consistency(raw)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cloud = pipeline.orient_representatives(opts)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    after = consistency(cloud)
    nrm = cloud[:, 3:].norm(dim=1)
    print(f"{os.path.basename(path):12s} N={cloud.shape[0]}  whole flow {dt:.3f} s (second run)  neighbour sign agreement "
          f"{before:.4f} -> {after:.4f}  |n| in [{float(nrm.min()):.6f}, {float(nrm.max()):.6f}]  finite: {bool(torch.isfinite(cloud).all())}", flush=True)
