#!/usr/bin/env python3
"""How much of a pair-kernel launch is ramp-up and drain?  The bench launch (100k sphere, 256 patches, cloud sorted by
patch) re-cut into finer work items WITHOUT changing the pairs: every patch is split into `split` consecutive parts of
its point list (so every part keeps the whole patch's bounding box and the far / near mix stays what it is), and the
launch evaluates P * split slabs of the same N targets.  Twice the items of half the length: what changes is the item
time T (and with it the drain at the end of the launch, ~T/2 by the model in DESIGN.md section 4) against the
per-item prologue / epilogue.  Also sweeps the number of patches in one launch (K = 32 ... 256 of the 256) to separate
the per-launch constant from the per-patch slope.

    python tools/gpu_item_size.py            (on the GPU box)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib, util  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
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


def main():
    dev = torch.device("cuda:0")
    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr(patches, dev)
    pts = pc.to(dev)[idx].contiguous()
    N, P = pts.shape[0], len(sizes)
    print("# per-launch constant: K of the 256 patches in one launch (ms median / min)")
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes = fu._patch_boxes(pts, off, None)
    rows = []
    for K in (16, 32, 64, 128, 192, 256):
        med, mn = timed(lambda: fu._patch_slabs(pts, off, None, point_patch, 0, K, 1e-5, boxes))
        pairs = float(sizes[:K].sum()) * N
        rows.append((K, med, mn, pairs))
        print(f"K={K:4d}  {med:.4f}  {mn:.4f}   {pairs / mn / 1e9:.2f} Tpairs/s")
    ks = np.array([r[0] for r in rows], dtype=float)
    mins = np.array([r[2] for r in rows])
    slope, const = np.polyfit(ks[1:], mins[1:], 1)
    print(f"# fit over K >= 32: {slope * 1e3:.2f} us per patch + {const * 1e3:.1f} us per launch")

    print("# source split inside the workgroup (dnp_patch_fields_tiled_f32 source_split): K patches x split, ms median / min")
    tiles = fu._TileTables(pts, sizes)
    w_part = torch.empty((P, tiles.n_tiles, 2), dtype=torch.float64, device=dev)
    ref = fu._patch_slabs(pts, off, None, point_patch, 0, 32, 1e-5, boxes, tiles.boxes, w_part[:32], 1)
    wref = w_part[:32].clone()
    for ss in (-32,):          # every patch split (round 4: through the exchange buffer; round 3's LDS form was source_split = 4)
        got = fu._patch_slabs(pts, off, None, point_patch, 0, 32, 1e-5, boxes, tiles.boxes, w_part[:32], ss)
        print(f"#   split {ss}: slabs bit-identical to split 1: {bool(torch.equal(got, ref))}, partials: {bool(torch.equal(w_part[:32], wref))}")
    for K in (16, 32, 64, 128, 256):
        line = f"K={K:4d}"
        for ss in (1, -K):
            med, mn = timed(lambda: fu._patch_slabs(pts, off, None, point_patch, 0, K, 1e-5, boxes, tiles.boxes, w_part[:K], ss),
                            reps=30 if K < 256 else 16)
            line += f"   {'plain' if ss == 1 else 'all split'} {med:.4f} / {mn:.4f}"
        print(line, flush=True)

    print("# item size: every patch cut into `split` parts (same pairs, split x the items, 1/split the item length)")
    off_np = off.cpu().numpy()
    for split in (1, 2, 3, 4):
        cuts = [off_np[0]]
        for k in range(P):
            lo, hi = off_np[k], off_np[k + 1]
            for i in range(1, split + 1):
                cuts.append(lo + (hi - lo) * i // split)
        off2 = torch.from_numpy(np.array(cuts, dtype=np.int64)).to(dev)
        P2 = P * split
        # the far test must see the PATCH's box for every part (so that the far / near mix is unchanged)
        boxes2 = boxes.repeat_interleave(split, dim=0).contiguous()
        # own-patch exclusion: rows of the whole patch must stay excluded -> give every part the same group as the
        # kernel compares tgt_group[row] == chunk id: use a group table per part (only the part itself is excluded here,
        # the other parts of the patch are evaluated: a few more near pairs, < 0.4 % of the launch)
        pp2 = torch.repeat_interleave(torch.arange(P2, device=dev), off2[1:] - off2[:-1])
        K2 = P2
        med, mn = timed(lambda: fu._patch_slabs(pts, off2, None, pp2, 0, K2, 1e-5, boxes2), reps=12)
        print(f"split={split}  items x{split}  {med:.4f}  {mn:.4f} ms")


if __name__ == "__main__":
    main()
