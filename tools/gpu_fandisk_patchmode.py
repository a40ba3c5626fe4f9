#!/usr/bin/env python3
"""Round 4, one more door for fandisk all-pairs (BASELINE config 2, 82 us against a 51 us arithmetic floor): the DRIVERS' tabled
kernel (two-wavefront workgroups, no LDS, no barrier, box tables) on the unsorted fandisk cloud cut into 23 ... 128 chunks of
consecutive rows, pair kernel only (fp32 slabs; the second pass would come on top), against the product call (LDS kernel +
reduce).  Synchronised single calls through the Python wrappers.  Result (one MI355X): product 101.7 us; tabled kernel alone
111.6 / 119.5 / 88.3 / 87.1 / 79.2 / 82.0 us at 23 / 32 / 46 / 64 / 93 / 128 chunks - no better once the second pass is added.
    python tools/gpu_fandisk_patchmode.py      (GPU box)"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden
from dipole_normal_prop_amd import field_utils as fu
dev = torch.device("cuda:0")
pc = torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]).to(dev)
N = pc.shape[0]
def timed(fn, reps=40):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return np.median(ts), np.min(ts)
print("field_grad (product, LDS kernel + reduce):", timed(lambda: fu.field_grad(pc, pc)))
for nch in (23, 32, 46, 64, 93, 128):
    sizes = np.full(nch, N // nch, dtype=np.int64); sizes[: N - sizes.sum()] += 1
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)])).to(dev)
    pp = torch.repeat_interleave(torch.arange(nch, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(pc, off, None), fu._TileTables(pc, sizes)
    print(f"patch-mode tabled kernel, {nch} chunks of ~{N // nch} sources (pair kernel only, fp32 slabs):",
          timed(lambda: fu._patch_slabs(pc, off, None, pp, 0, nch, 1e-5, boxes, tiles.boxes, None, 1)))
