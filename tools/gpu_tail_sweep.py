#!/usr/bin/env python3
"""Sweep of the split-tail size (patch_drivers.TAIL_SOURCES) for one rank's share of eight on the three partitions of
tools/gpu_rank_share.py: the N = 8 launch of rank 0 against the full launch's time / 8.
    python tools/gpu_tail_sweep.py -> profiles/r05_tail_sweep.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, patch_drivers as pd, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=40):
    for _ in range(10):
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
    return float(np.median(ts))


def run(name, cloud, patches):
    off, idx, sizes = util.patch_csr(patches, dev)
    swork = cloud.to(dev)[idx].contiguous()
    N, P = swork.shape[0], len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(swork, off, None), fu._TileTables(swork, sizes)
    wp_full = torch.empty((P, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=dev) if tiles.fused else None
    t_full = timed(lambda: fu._patch_slabs(swork, off, None, point_patch, 0, P, 1e-5, boxes, tiles.boxes, wp_full, 1), 10)
    pairs_full = float(sizes.sum()) * N
    b = fu._balanced_blocks(sizes, 8)
    print(f"## {name}: {P} patches of {sizes.min()}..{sizes.max()} points; full launch {t_full:.4f} ms", flush=True)
    for r in range(8):
        lo, hi = int(b[r]), int(b[r + 1])
        ideal = t_full * float(sizes[lo:hi].sum()) * N / pairs_full
        wp = torch.empty((hi - lo, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=dev) if tiles.fused else None
        line = [f"rank {r} (patches [{lo},{hi}))"]
        for ts in (0, 600, 800, 1000, 1300, 1600, 2000, 2600):
            pd.TAIL_SOURCES = ts if ts else 1000
            split = fu._pick_source_split(sizes[lo:hi], N) if ts else 1
            t = timed(lambda: fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, split))
            line.append(f"{ts or 'plain'}:{split}:{ideal / t:.3f}")
        pd.TAIL_SOURCES = 1000
        print("  " + "  ".join(line), flush=True)
        line = [f"    forced k (last patches {sizes[max(lo, hi - 8):hi].tolist()}):"]
        for k in range(0, 9):
            split = -k if k else 1
            t = timed(lambda: fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, wp, split))
            line.append(f"{k}:{ideal / t:.3f}")
        print("  ".join(line), flush=True)


def largest_first_within_blocks(patches):
    """the same 8 rank blocks, every block's patches in descending size: what an LPT launch order would dispatch"""
    sizes = np.array([len(p) for p in patches])
    b = fu._balanced_blocks(sizes, 8)
    out = []
    for r in range(8):
        blk = list(range(int(b[r]), int(b[r + 1])))
        out += [patches[i] for i in sorted(blk, key=lambda i: -sizes[i])]
    return out


pc, patches, _ = headline_workload()
run("bench workload (256 Fibonacci patches)", pc, patches)
grid = [p.cpu() for p in util.divide_pc(pc[:, :3].to(dev), 24, min_patch=100)]
run("the same sphere, the reference's grid partition (util.divide_pc n_part 24, min 100)", pc, grid)
run("the grid partition, every rank block largest patch first", pc, largest_first_within_blocks(grid))
g = np.load(os.path.join(ROOT, "tests", "golden", "G15_boxunion_config3.npz"))
cloud = torch.from_numpy(g["pc"])
reps = [torch.from_numpy(g["rep_idx"][g["rep_off"][k]:g["rep_off"][k + 1]].astype(np.int64)) for k in range(len(g["rep_off"]) - 1)]
sub = cloud[torch.cat(reps)]
starts = np.concatenate([[0], np.cumsum([len(r) for r in reps])])
box = [torch.arange(starts[k], starts[k + 1]) for k in range(len(reps))]
run("G15 boxunion representatives (369 patches, config 3)", sub, box)
run("boxunion representatives, every rank block largest patch first", sub, largest_first_within_blocks(box))
