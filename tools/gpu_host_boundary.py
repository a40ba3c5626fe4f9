#!/usr/bin/env python3
"""What does it cost to hand the drivers HOST tensors (the reference's functions take either)?  The config-4 patch driver and a
100 000^2 field_grad with the cloud on the host against the same calls on device tensors, and a cProfile of one host-tensor
driver call (synchronised).    python tools/gpu_host_boundary.py   (on the GPU box)"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")
pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
N = pts.shape[0]
ranges = util.PatchList(torch.arange(N, device=dev), sizes, disjoint=True)
host = pts.cpu()
pinned = host.clone().pin_memory()


def timed(fn, reps=8):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def driver(t):
    fu.strongest_field_propagation(t, list(enumerate(ranges)), ranges, diffuse=True)


print(f"patch driver (config 4), device tensor:          {timed(lambda: driver(pts.clone())):8.3f} ms")
print(f"patch driver, host tensor (pageable, in place):  {timed(lambda: driver(host)):8.3f} ms")
print(f"patch driver, host tensor (pageable, cloned per call - the clone is a torch CPU copy): {timed(lambda: driver(host.clone())):8.3f} ms")
print(f"patch driver, host tensor (pinned, in place):    {timed(lambda: driver(pinned)):8.3f} ms")
print(f"field_grad 100k^2, device tensors:               {timed(lambda: fu.field_grad(pts, pts)):8.3f} ms")
print(f"field_grad 100k^2, host tensors (pageable):      {timed(lambda: fu.field_grad(host, host)):8.3f} ms")
print(f"  the copies alone: H2D pageable {timed(lambda: host.to(dev)):.3f} ms, pinned {timed(lambda: pinned.to(dev, non_blocking=True)):.3f} ms, "
      f"D2H of [N,3] {timed(lambda: pts[:, 3:].contiguous().cpu()):.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    driver(host)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
