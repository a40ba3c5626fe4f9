#!/usr/bin/env python3
"""Experiment: fill the tail of a patch-mode launch with short items (two forms: second stream; same stream without a barrier).  The last k of K patches are evaluated by a second
launch with source_split = 4 (items a third as long; results bit-identical) on a LOW-priority stream while the first K - k
run on the normal stream: if the hardware honours the priorities the short items flow in when the long ones run out.
Total time (fork -> join) against the single launch, K = 32 and 256, several k."""
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
main = torch.cuda.current_stream()
lo_prio, hi_prio = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
side = torch.cuda.Stream(priority=0)
fast = torch.cuda.Stream(priority=-1)


def launch(p0, p1, ss, stream):
    rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(pp), _lib.ptr(boxes),
                                        _lib.ptr(tiles.boxes), p0, p1, 1e-5, ctypes.c_void_p(dE[p0].data_ptr()),
                                        ctypes.c_void_p(wp[p0].data_ptr()), ss, ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0


def timed(fn, reps=40):
    for _ in range(60):
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
    return float(np.median(ts)), float(np.min(ts))


for K in ((32, 256) if os.environ.get("TAIL_FILL_STREAMS") else ()):      # the two-stream form: TAIL_FILL_STREAMS=1
    print(f"K={K}: single launch {timed(lambda: launch(0, K, 1, main))}")
    for k in (2, 4, 8):
        for label, hi_s, lo_s in (("main=normal, tail=low", main, side), ("main=high, tail=normal", fast, side)):
            ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()

            def both():
                ev_fork.record(main)
                hi_s.wait_event(ev_fork)
                lo_s.wait_event(ev_fork)
                launch(0, K - k, 1, hi_s)
                launch(K - k, K, 4, lo_s)
                ev_join.record(lo_s)
                main.wait_event(ev_join)
                if hi_s is not main:
                    e2 = torch.cuda.Event()
                    e2.record(hi_s)
                    main.wait_event(e2)
            print(f"   last {k} patches split x4 on a second stream ({label}): {timed(both)}")

# (A second form - the tail launch dispatched BEHIND the main launch in the same stream without a barrier, hipExtAnyOrderLaunch -
# was measured with an experiment build of this round and removed again: gfx9 keeps the barrier; profiles/r03_tail_fill.txt.)
# Third form, the one that shipped: ONE launch whose last k patches are split (source_split = -k, pair_kernel.h TAIL).
print("# one launch, the last k patches split (source_split = -k); interleaved with the plain launch and the all-split launch")
for K in (4, 8, 12, 16, 24, 32, 40, 48, 64, 128, 256):
    launch(0, K, 1, main)
    torch.cuda.synchronize()
    ref, wref = dE[:K].clone(), wp[:K].clone()
    line = f"K={K}: split 1 {timed(lambda: launch(0, K, 1, main), reps=30)[0]:.4f}  split 4 {timed(lambda: launch(0, K, 4, main), reps=30)[0]:.4f}"
    for k in (1, 2, 3, 4, 6, 8):
        if k >= K:
            continue
        dE[:K].zero_()
        wp[:K].zero_()
        launch(0, K, -k, main)
        torch.cuda.synchronize()
        same = bool(torch.equal(dE[:K], ref)) and bool(torch.equal(wp[:K], wref))
        line += f"  -{k}: {timed(lambda: launch(0, K, -k, main), reps=30)[0]:.4f}{'' if same else ' RESULTS DIFFER'}"
    line += f"  split 1 again {timed(lambda: launch(0, K, 1, main), reps=30)[0]:.4f}"
    print(line, flush=True)
