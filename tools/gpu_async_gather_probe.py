#!/usr/bin/env python3
"""Probe of parallel.gather_rows_async on this RCCL build with a ONE-rank nccl group (all a one-GPU box allows): the call
pattern bench.py uses at N > 1 - kernel -> asynchronous all-gather -> next kernel, two gathers in flight, wait() of the
oldest, drain before a barrier - runs, returns the gathered rows, and does not make the launch stream wait (the time of
a loop of pair launches with and without the gathers).    python tools/gpu_async_gather_probe.py   (on the GPU box)"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import parallel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
bounds = np.array([0, 32])
a = torch.randn(2048, 2048, device=dev)


def loop(with_gather, steps=40):
    pending, outs = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        b = a @ a                                            # stands for the pair kernel
        W = (b[:32, :256].double() + i).contiguous()
        if with_gather:
            out, work = parallel.gather_rows_async(W, bounds, force=True)
            assert work is not None
            pending.append((out, work, W))
            while len(pending) > 2:
                o, w, src = pending.pop(0)
                w.wait()
                outs.append((o, src))
    while pending:
        o, w, src = pending.pop(0)
        w.wait()
        outs.append((o, src))
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    for o, src in outs:
        assert torch.equal(o, src)
    return dt


loop(True, 5)
print(f"loop of 40 steps: {loop(False):.4f} ms per step without the gathers, {loop(True):.4f} ms with the asynchronous gathers "
      f"(one-rank nccl group; results equal the local rows)")
dist.destroy_process_group()
