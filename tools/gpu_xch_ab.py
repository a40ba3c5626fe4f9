#!/usr/bin/env python3
"""Round 4: the split tail through the exchange buffer (pair_kernel.h XCH: no LDS, no barrier, last arriver adds the run
terms in run order) against the plain launch, on the bench workload's first K patches (K = a rank's share of 256 / K ranks
... the whole launch).  (While round 3's LDS forms of the split still existed this tool also timed them: the lds-3 / lds-all
columns of profiles/r04_xch_ab.txt.)  Same process, interleaved, medians of 30 launches;
every variant's slabs and interaction partials are compared bit for bit with the plain launch's.
    python tools/gpu_xch_ab.py            (on the GPU box)"""
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

dev = torch.device("cuda:0")
lib = _lib.require_device()
pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
N, P = pts.shape[0], len(sizes)
pp = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
boxes, tiles = fu._patch_boxes(pts, off, None), fu._TileTables(pts, sizes)
dE = torch.empty((P, N, 3), dtype=torch.float32, device=dev)
wp = torch.empty((P, tiles.n_tiles, 2), dtype=torch.float64, device=dev)
xch = torch.zeros(int(lib.dnp_patch_exchange_bytes(N, P)), dtype=torch.uint8, device=dev)
main = torch.cuda.current_stream()


def launch(K, ss, exchange):
    rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(pp), _lib.ptr(boxes),
                                        _lib.ptr(tiles.boxes), 0, K, 1e-5, _lib.ptr(dE), _lib.ptr(wp), 2, ss,
                                        _lib.ptr(xch) if exchange else None, xch.numel() if exchange else 0,
                                        ctypes.c_void_p(main.cuda_stream))
    assert rc == 0, lib.dnp_last_error()


def timed(fn, reps=30):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        fn()
        b.record(main)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


ks = [int(x) for x in os.environ.get("XCH_KS", "1,2,3,4,6,8,12,16").split(",")]
print("# ms per launch (median of 30): plain = source_split 1; x-k = the last k patches split through the exchange buffer")
for K in [int(x) for x in os.environ.get("XCH_K", "4,8,16,32,48,64,128,256").split(",")]:
    launch(K, 1, False)
    torch.cuda.synchronize()
    ref, wref = dE[:K].clone(), wp[:K].clone()
    line = f"K={K:3d}: plain {timed(lambda: launch(K, 1, False)):.4f}"
    for k in ks + [K]:
        if k > K or (k == K and K in ks):
            continue
        dE[:K].zero_()
        wp[:K].zero_()
        launch(K, -k, True)
        launch(K, -k, True)           # a second launch on the same buffer: the counters re-armed themselves
        torch.cuda.synchronize()
        same = bool(torch.equal(dE[:K], ref)) and bool(torch.equal(wp[:K], wref))
        line += f"  x-{k} {timed(lambda: launch(K, -k, True)):.4f}{'' if same else ' RESULTS DIFFER'}"
    line += f"  plain again {timed(lambda: launch(K, 1, False)):.4f}"
    print(line, flush=True)
