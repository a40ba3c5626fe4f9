#!/usr/bin/env python3
"""One rank's share of an N-rank run on ONE GPU, for several partitions (round-4 verdict, item 3b): the pair-kernel launch
rank 0 of N = 1, 2, 4, 8 makes (field_utils._balanced_blocks + _pick_source_split, exactly the drivers' choices) against
its ideal - the full launch's time times the rank's fraction of the pairs.  Partitions: the bench's 256 Fibonacci patches
(343..439 points), the REFERENCE's grid partition of the same sphere (util.divide_pc n_part = 24: 243 patches of 100..677
points - round 4's split tail never fired there: patches of <= 128 or > 512 points) and G15's boxunion representatives (369
patches of 100..500).  Every rank of the N is timed (the slowest one decides a step); columns: ms and share of ideal with the drivers' launch plan
(longest patch first, a split tail when even the shortest patches are long) and with the plain patch-order launch.
    python tools/gpu_rank_share.py  -> profiles/r05_rank_share_partitions.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.08:
        fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def run(name, cloud, patches):
    off, idx, sizes = util.patch_csr(patches, dev)
    swork = cloud.to(dev)[idx].contiguous()
    N, P = swork.shape[0], len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(swork, off, None), fu._TileTables(swork, sizes)
    print(f"## {name}: {N} points, {P} patches of {sizes.min()}..{sizes.max()} points (median {int(np.median(sizes))}), "
          f"interaction partials with {tiles.slots} group slots per tile")
    full = None
    for world in (1, 2, 4, 8):
        b = fu._balanced_blocks(sizes, world)
        shares, plain_shares = [], []
        for r in range(world):
            lo, hi = int(b[r]), int(b[r + 1])
            pairs = float(sizes[lo:hi].sum()) * N
            order, split = fu._launch_plan(sizes[lo:hi], N, dev)          # the drivers' plan: longest patch first (+ tail)
            wp = torch.empty((hi - lo, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=dev) if tiles.fused else None
            ref = fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, 1)
            got = fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, split, order)
            same = bool(torch.equal(ref, got))
            del ref, got
            t_plan, _ = timed(lambda: fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, split, order))
            t_plain, _ = timed(lambda: fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, 1))
            if world == 1:
                full = (min(t_plan, t_plain), pairs)
            ideal = full[0] * pairs / full[1]
            shares.append(ideal / t_plan)
            plain_shares.append(ideal / t_plain)
            if r == 0 or not same:
                print(f"N={world}: rank {r} = patches [{lo},{hi}) {pairs:.3e} pairs, longest first: {order is not None}, source_split {split}: "
                      f"{t_plan:7.4f} ms (share of ideal {ideal / t_plan:5.3f}) | patch order, no tail {t_plain:7.4f} ms ({ideal / t_plain:5.3f}) | "
                      f"slabs bit-identical: {same}", flush=True)
        print(f"N={world}: share of ideal over the {world} ranks: min {min(shares):5.3f}, max {max(shares):5.3f} "
              f"(patch order, no tail: min {min(plain_shares):5.3f}, max {max(plain_shares):5.3f})", flush=True)


pc, patches, _ = headline_workload()
run("bench workload (256 Fibonacci patches)", pc, patches)
grid = util.divide_pc(pc[:, :3].to(dev), 24, min_patch=100)
run("the same sphere, the reference's grid partition (util.divide_pc n_part 24, min 100)", pc, [p.cpu() for p in grid])
g = np.load(os.path.join(ROOT, "tests", "golden", "G15_boxunion_config3.npz"))
cloud = torch.from_numpy(g["pc"])
reps = [torch.from_numpy(g["rep_idx"][g["rep_off"][k]:g["rep_off"][k + 1]].astype(np.int64)) for k in range(len(g["rep_off"]) - 1)]
sub = cloud[torch.cat(reps)]
starts = np.concatenate([[0], np.cumsum([len(r) for r in reps])])
run("G15 boxunion representatives (369 patches, config 3)", sub, [torch.arange(starts[k], starts[k + 1]) for k in range(len(reps))])
