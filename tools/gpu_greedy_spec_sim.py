#!/usr/bin/env python3
"""Would speculative prefetch pay in the patch greedy loop (dnp_patch_greedy: 255 dependent steps x 0.95 us, each waiting for the
winner's W row from L2)?  Computes W of the headline workload with the product kernels, replays the greedy loop in numpy and
counts how often the winner of step s + 1 was among the top-K runners-up of step s (whose rows a speculative load issued at
step s would already hold).
    python tools/gpu_greedy_spec_sim.py      (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")
pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
P = len(sizes)
pp = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
boxes, tiles = fu._patch_boxes(pts, off, None), fu._TileTables(pts, sizes)
_, W = fu._slabs_and_rows(pts, off, pp, 0, P, 1e-5, boxes, tiles, sizes)
W = W.cpu().numpy()
start = int(np.load(os.path.join(ROOT, "tests", "golden", "G19_headline_sphere_patch_propagation.npz"))["order"][0])
sigma = np.ones(P)
visited = np.zeros(P, bool)
visited[start] = True
inter = W[start].copy()
hits = {k: 0 for k in (1, 2, 3, 4, 8)}
prev_top = None
steps = 0
for _ in range(P - 1):
    mag = np.where(visited, -np.inf, np.abs(inter))
    rank = np.argsort(-mag, kind="stable")
    j = int(rank[0])
    if prev_top is not None:
        steps += 1
        for k in hits:
            hits[k] += j in prev_top[:k]
    prev_top = list(rank[1:9])            # the runners-up of THIS step: candidates for the next one
    if inter[j] < 0:
        sigma[j] = -1.0
    visited[j] = True
    inter += sigma[j] * W[j]
print(f"headline workload, {P} patches: winner of step s+1 among the top-K runners-up of step s:")
for k, h in hits.items():
    print(f"  K = {k}: {h} of {steps} steps ({100.0 * h / steps:.1f} %)")
