#!/usr/bin/env python3
"""One leg of the product per process, for rocprofv3 (tools/profile_legs.sh): 3 synchronised calls (lazy initialisation, clock ramp),
then `reps` calls back to back, and a JSON sidecar gpurun_out/prof_<round>_legs/<leg>.json that says what one call is - the
anchor kernel that starts a call, and for every product kernel of the leg the work units (pairs / bytes) of ONE launch and the
roofline that bounds it - so that tools/make_summary.py can turn the kernel traces into profiles/SUMMARY_<round>.md without
guessing.  Legs = the BASELINE configs and the entry points bench.py times (other_configs), fp32 and fp64.

    python tools/gpu_leg.py <leg> <sidecar.json>"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import N_PATCHES, N_POINTS, headline_workload  # noqa: E402

leg, sidecar = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
gdir = os.path.join(ROOT, "tests", "golden")
FLOP = 33
meta = {"leg": leg, "kernels": {}}


def K(substr, units, unit, bound, per_unit, note=""):
    """kernel whose name contains `substr`: `units` per launch; bound "valu32"/"valu64": per_unit = flop per unit; "hbm": bytes per unit"""
    meta["kernels"][substr] = {"units": float(units), "unit": unit, "bound": bound, "per_unit": per_unit, "note": note}


def sphere_sorted():
    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr(patches, dev)
    pts = pc.to(dev)[idx].contiguous()
    ranges = util.PatchList(torch.arange(pts.shape[0], device=dev), sizes, disjoint=True)
    return pts, ranges, sizes


reps = 12
if leg in ("config4_driver", "config4_driver_f64"):
    pts, ranges, sizes = sphere_sorted()
    f64 = leg.endswith("f64")
    if f64:
        pts = pts.double()
    call = lambda: fu.strongest_field_propagation(pts.clone(), list(enumerate(ranges)), ranges, diffuse=True)
    meta.update(anchor="patch_pca_kernel", what=f"strongest_field_propagation(diffuse) on the 100 000-point sphere, 256 patches, {'float64' if f64 else 'float32'}")
    pairs = float(sizes.sum()) * N_POINTS
    K("pair_kernel_scalar", pairs, "pairs", "valu64" if f64 else "valu32", FLOP, "all 256 per-patch fields + interaction partials in one launch")
    K("patch_greedy_kernel", N_PATCHES - 1, "steps", "latency", 0, "255 dependent steps of one wavefront")
    K("combine_signed_kernel", float(N_PATCHES) * N_POINTS * 3 * (8 if f64 else 4), "bytes", "hbm", 1, "reads all slabs once")
    reps = 6 if f64 else 12
elif leg == "config3_reps":
    g = np.load(os.path.join(gdir, "G15_boxunion_config3.npz"))
    cloud = torch.from_numpy(g["pc"]).clone()
    cloud[~torch.from_numpy(g["prefilter_sign"]), 3:] *= -1
    cloud = cloud.to(dev)
    i64 = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev)
    rl = util.RepLists(util.PatchList(i64(g["rep_idx"]), np.diff(g["rep_off"]), disjoint=True),
                       util.PatchList(i64(g["rest_idx"]), np.diff(g["rest_off"]), disjoint=True))
    call = lambda: fu.strongest_field_propagation_reps(cloud.clone(), rl, diffuse=True)
    nrep, nrest = int(g["rep_off"][-1]), int(g["rest_off"][-1])
    meta.update(anchor="patch_pca_kernel", what=f"strongest_field_propagation_reps(diffuse) on boxunion: 369 patches, {nrep} representatives, {nrest} rest points")
    K("pair_kernel_scalar<float, float", float(nrep) ** 2, "pairs", "valu32", FLOP, "369 per-patch fields on the representatives")
    K("pair_kernel_scalar<float, double", float(nrep) * nrest, "pairs", "valu32", FLOP, "field of all representatives at the rest points (sources in point order: exact chain)")
    K("patch_greedy_kernel", 368, "steps", "latency", 0)
    K("combine_signed_kernel", 369.0 * nrep * 12, "bytes", "hbm", 1)
elif leg in ("config2_fandisk", "config2_fandisk_f64"):
    fd = torch.from_numpy(np.load(os.path.join(gdir, "G5_fandisk_allpairs.npz"))["pc"]).to(dev)
    f64 = leg.endswith("f64")
    fd = fd.double() if f64 else fd
    call = lambda: fu.field_grad(fd, fd)
    n = fd.shape[0]
    meta.update(anchor="pair_kernel", what=f"field_grad, fandisk all-pairs ({n}^2), {'float64' if f64 else 'float32'}")
    K("pair_kernel", float(n) ** 2, "pairs", "valu64" if f64 else "valu32", FLOP, "LDS-staged kernel, KT = 1")
    K("reduce_kernel", 64.0 * n * 24, "bytes", "hbm", 1, "second pass over the fp64 chunk sums")
    reps = 50
elif leg in ("allpairs_100k", "allpairs_100k_f64", "config5_reference_field", "config5_reference_field_f64", "potential_lattice",
             "potential_lattice_f64"):
    pts, _, _ = sphere_sorted()
    f64 = leg.endswith("f64")
    pts = pts.double() if f64 else pts
    kind = "valu64" if f64 else "valu32"
    if leg.startswith("allpairs"):
        call = lambda: fu.field_grad(pts, pts)
        meta.update(anchor="pair_kernel_scalar", what="field_grad 100 000^2 on the patch-sorted sphere")
        K("pair_kernel_scalar", float(N_POINTS) ** 2, "pairs", kind, FLOP)
        K("reduce_kernel", 0, "bytes", "hbm", 1, "second pass over the fp64 chunk sums")
    elif leg.startswith("config5"):
        gg = torch.Generator().manual_seed(3)
        tgt = (pts[:, :3].cpu().float() + 1e-3 * torch.randn(N_POINTS, 3, generator=gg)).to(dev).to(pts.dtype)
        call = lambda: fu.reference_field(pts, tgt)
        meta.update(anchor="pair_kernel_scalar", what="reference_field 100 000 -> 100 000 (3-column targets)")
        K("pair_kernel_scalar", float(N_POINTS) ** 2, "pairs", kind, FLOP)
        K("reduce_rows_kernel", 0, "bytes", "hbm", 1, "second pass + normalisation tail")
    else:
        grid = util.gen_grid().to(dev).to(pts.dtype)
        call = lambda: fu.potential(pts, grid)
        meta.update(anchor="pair_kernel", what="potential, 100 000 sources x 1000-point lattice")
        K("pair_kernel", float(N_POINTS) * grid.shape[0], "pairs", kind, 13, "13 flop per pair (3 sub, 5 r.r, 5 p.r + rsq, cube, fma)")
        reps = 50
    if not leg.startswith("potential"):
        reps = 4 if f64 else 8
elif leg in ("config1_points", "config1_points_f64"):
    ok = torch.from_numpy(np.load(os.path.join(gdir, "G8_point_propagation.npz"))["pc_full"]).to(dev)
    ok = ok.double() if leg.endswith("f64") else ok
    call = lambda: fu.strongest_field_propagation_points(ok.clone(), diffuse=True)
    meta.update(anchor="point_greedy", what=f"strongest_field_propagation_points on ok.xyz ({ok.shape[0]} points)")
    K("point_greedy", ok.shape[0] - 1, "steps", "latency", 0, "one persistent launch, one workgroup per CU")
    reps = 6
elif leg in ("xie_order", "xie_order_f64"):
    gen = torch.Generator().manual_seed(5)
    n = 10000
    x = torch.randn(n, 6, generator=gen)
    pc = torch.cat([0.4 * x[:, :3] / x[:, :3].norm(dim=1, keepdim=True), torch.nn.functional.normalize(x[:, 3:], dim=1)], 1).to(dev)
    f64 = leg.endswith("f64")
    pc = pc.double() if f64 else pc
    orders = np.stack([np.random.default_rng(s).permutation(n) for s in range(5)])
    call = lambda: fu.xie_propagation_points_in_order(pc, 0.1, orders, diffuse=True)
    esz = 8 if f64 else 4
    meta.update(anchor="xie_pairs_kernel", what=f"xie_propagation_points_in_order(diffuse), N = {n}, 5 orders, {'float64' if f64 else 'float32'}")
    K("xie_pairs_kernel", float(n) * n * esz, "bytes", "hbm", 1, "N x N matrix written once")
    K("xie_block_rows_kernel", 256.0 * n * esz * 5, "bytes", "hbm", 1, "row sums of one 256-step block: 256 matrix rows per order, 5 orders")
    K("xie_block_solve_kernel", 256, "steps", "latency", 0, "the 256 dependent steps of a block: one wavefront per order")
    K("xie_rowdots_kernel", float(n) * n * esz, "bytes", "hbm", 1, "the diffuse pass: one pass over the matrix for all orders")
    reps = 8
elif leg == "xie_knn":
    gen = torch.Generator().manual_seed(5)
    n = 10000
    x = torch.randn(n, 6, generator=gen)
    pc = torch.cat([0.4 * x[:, :3] / x[:, :3].norm(dim=1, keepdim=True), torch.nn.functional.normalize(x[:, 3:], dim=1)], 1).to(dev)
    call = lambda: fu.xie_intersaction(pc, pc, 0.1, 20, 3)
    meta.update(anchor="xie_knn_kernel", what=f"xie_intersaction with knn_mask = 20, N = {n} (selection + masked matrix)")
    K("xie_knn_kernel", n, "sources", "latency", 0, "k-th nearest target of every source: one wavefront per source over all N targets")
    K("xie_pairs_kernel", float(n) * n * 4, "bytes", "hbm", 1, "masked N x N matrix written once")
    reps = 10
elif leg == "prep_partition":
    g = np.load(os.path.join(gdir, "G15_boxunion_config3.npz"))
    cloud = torch.from_numpy(g["pc"]).to(dev)
    call = lambda: util.divide_pc(cloud[:, :3], 41, min_patch=100)
    meta.update(anchor="", what="util.divide_pc: voxel partition + merge of 100 000 points (41^3 voxels, min 100)")
    reps = 8
else:
    sys.exit(f"unknown leg {leg}")

for _ in range(3):
    call()
    torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
meta["wall_ms_per_call_under_profiler"] = (time.perf_counter() - t0) / reps * 1e3
meta["reps"], meta["warm"] = reps, 3
fu.flush_warnings()
with open(sidecar, "w") as f:
    json.dump(meta, f, indent=1)
print(f"{leg}: {meta['wall_ms_per_call_under_profiler']:.3f} ms per call", flush=True)
