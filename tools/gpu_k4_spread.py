#!/usr/bin/env python3
"""Why does the per-point propagation of ok.xyz (BASELINE config 1: 10 000 steps of a latency-bound persistent kernel, 20
workgroups on 20 CUs) take 28 ms in one bench run and 50 ms in another?  (Round-3 verdict, weak #2.)

Part 1 - the product library: the call timed 14 times in three regimes of what ran on the device just before it:
    cold    after 1.5 s of an idle device
    warm    directly after 60 ms of pair-kernel launches (the chip at its sustained clock)
    chained back to back, each call the previous one's predecessor
Part 2 - a -DDNP_K4_STATS build (tools/bin/libdnp_k4stats.so): workgroup 0 stamps every step with the 100 MHz wall clock AND
the shader clock counter (their ratio IS the clock the kernel ran at) and its spin count; every workgroup leaves the XCD
and CU it ran on.  Printed per call: total ms, effective shader clock, step-time percentiles in ns, spins of the slowest polling lane, the
XCDs of the 20 workgroups, and the step split into workgroup 0's own work (field update + local argmax, up to the publish) and the
rest (all-gather of the 20 granules, the winner's row, the wave argmax).
    python tools/gpu_k4_spread.py            (on the GPU box; output -> profiles/r04_k4_spread.txt)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import _lib  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

dev = torch.device("cuda:0")
ok = torch.from_numpy(load_golden("G8_point_propagation")["pc_full"])
N = ok.shape[0]


def heat(ms=60.0):
    """pair-kernel launches for `ms` of device time: the load that takes the chip to its sustained clock"""
    pc, patches, _ = headline_workload()
    if not hasattr(heat, "state"):
        from dipole_normal_prop_amd import util
        off, idx, sizes = util.patch_csr(patches, dev)
        pts = pc.to(dev)[idx].contiguous()
        pp = torch.repeat_interleave(torch.arange(len(sizes), device=dev), off[1:] - off[:-1])
        heat.state = (pts, off, pp, fu._patch_boxes(pts, off, None), fu._TileTables(pts, sizes))
    pts, off, pp, boxes, tiles = heat.state
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        fu._patch_slabs(pts, off, None, pp, 0, 64, 1e-5, boxes, tiles.boxes, None, 1)
        torch.cuda.synchronize()


def one_call(lib, dtype=torch.float32, stats=None):
    work = ok.to(dtype).to(dev).clone()
    order = torch.empty(N, dtype=torch.int64, device=dev)
    ws = torch.empty(int(lib.dnp_point_greedy_workspace_bytes(N, work.element_size())), dtype=torch.uint8, device=dev)
    fn = lib.dnp_point_greedy_f64 if dtype == torch.float64 else lib.dnp_point_greedy_f32
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    rc = fn(_lib.ptr(work), N, 6, 0, 1e-6, 1, _lib.ptr(order), None, 0, 0, _lib.ptr(ws), ws.numel(), _lib.current_stream())
    b.record()
    torch.cuda.synchronize()
    assert rc == 0
    return a.elapsed_time(b)


def main():
    lib = _lib.require_device()
    print("# part 1: the product library, ok.xyz fp32 (10 000 steps), ms per call by what the device did just before")
    for regime in ("cold", "warm", "chained"):
        ts = []
        for r in range(14):
            if regime == "cold":
                time.sleep(1.5)
            elif regime == "warm":
                heat(60.0)
            ts.append(one_call(lib))
        ts = np.array(ts)
        print(f"{regime:8s} median {np.median(ts):6.2f}  min {ts.min():6.2f}  max {ts.max():6.2f}   all: {' '.join(f'{t:.1f}' for t in ts)}", flush=True)
    for regime in ("warm", "chained"):
        ts = np.array([(heat(60.0) if regime == "warm" else None, one_call(lib, torch.float64))[1] for _ in range(10)])
        print(f"fp64 {regime:8s} median {np.median(ts):6.2f}  min {ts.min():6.2f}  max {ts.max():6.2f}", flush=True)

    path = os.path.join(ROOT, "tools", "bin", "libdnp_k4stats.so")
    if not os.path.exists(path):
        print("# part 2 skipped: tools/bin/libdnp_k4stats.so not built")
        return
    slib = ctypes.CDLL(path)
    for name in ("dnp_point_greedy_f32", "dnp_point_greedy_f64", "dnp_point_greedy_workspace_bytes"):
        res, args = _lib.SIGNATURES[name]
        getattr(slib, name).restype, getattr(slib, name).argtypes = res, args
    slib.dnp_debug_set_k4_stats.argtypes = [ctypes.c_void_p]
    stats = torch.zeros(N * 4 + 256 * 2, dtype=torch.int64, device=dev)
    slib.dnp_debug_set_k4_stats(_lib.ptr(stats))
    print("# part 2: -DDNP_K4_STATS build, per call: ms | effective shader clock GHz (shader-clock ticks / wall time, whole call and by "
          "quarter of the steps) | step ns p10 p50 p90 p99 max | slowest lane's spins per step, mean | XCD of each workgroup | split of a step")
    for regime in ("cold", "warm", "warm", "chained", "chained", "chained"):
        if regime == "cold":
            time.sleep(1.5)
        elif regime == "warm":
            heat(60.0)
        stats.zero_()
        ms = one_call(slib)
        st = stats.cpu().numpy().astype(np.uint64)
        per = st[: (N - 1) * 4].reshape(N - 1, 4)
        wall, core, spins = per[:, 0].astype(np.float64), per[:, 1].astype(np.float64), per[:, 2].astype(np.float64)
        pub = per[:, 3].astype(np.float64)
        step_ns = np.diff(wall) * 10.0
        own_ns = (pub[1:] - wall[:-1]) * 10.0        # end of the previous step -> this step's granule published (field update + local argmax)
        wait_ns = (wall[1:] - pub[1:]) * 10.0        # published -> all granules polled, winner's row fetched, wave argmax done
        ghz = (core[-1] - core[0]) / ((wall[-1] - wall[0]) * 10.0)
        q = len(wall) // 4
        ghz_q = [(core[(i + 1) * q - 1] - core[i * q]) / ((wall[(i + 1) * q - 1] - wall[i * q]) * 10.0) for i in range(4)]
        grp = st[N * 4: N * 4 + 40].reshape(20, 2)
        print(f"{regime:8s} {ms:6.2f} ms | {ghz:.2f} GHz ({' '.join(f'{g:.2f}' for g in ghz_q)}) | "
              f"{np.percentile(step_ns, 10):.0f} {np.percentile(step_ns, 50):.0f} {np.percentile(step_ns, 90):.0f} "
              f"{np.percentile(step_ns, 99):.0f} {step_ns.max():.0f} | {spins.mean():.1f} | {''.join(str(int(x)) for x in grp[:, 1])}"
              f" | own work p50 {np.percentile(own_ns, 50):.0f} ns, gather + row + argmax p50 {np.percentile(wait_ns, 50):.0f} ns", flush=True)


if __name__ == "__main__":
    main()
