#!/usr/bin/env python3
"""Timing of the per-point greedy propagation forms (developer tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import point_driver as ptd  # noqa: E402
from conftest import load_golden
from tools.gpu_check import sphere

dev = torch.device("cuda:0")
def run(pc, label):
    """median of three timed propagations after two untimed ones (round 2 timed a single call after one warm-up and
    recorded 45 ms where the steady figure is 28-29 ms)"""
    ts = []
    for i in range(5):
        a = pc.clone().to(dev); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fu.strongest_field_propagation_points(a, diffuse=True); torch.cuda.synchronize()
        if i >= 2:
            ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    n = pc.shape[0]
    print(f"{label}: N={n} {dt*1e3:.1f} ms -> {dt/n*1e6:.2f} us/step, {n*n/dt/1e9:.2f} Gpairs/s", flush=True)

ok = torch.from_numpy(load_golden("G8_point_propagation")["pc_full"])
ptd.POINT_GREEDY_FORM = 1
run(ok, "ok.xyz fp32 single-workgroup")
ptd.POINT_GREEDY_FORM = 2
run(ok, "ok.xyz fp32 multi-workgroup ")
run(ok.double(), "ok.xyz fp64 multi-workgroup ")
ptd.POINT_GREEDY_FORM = 1
run(ok.double()[:4096], "ok.xyz[:4096] fp64 single-workgroup")
ptd.POINT_GREEDY_FORM = 0
run(ok[:2000], "ok.xyz[:2000] fp32 auto")
run(sphere(30000), "sphere fp32 multi-workgroup")
run(sphere(100000), "sphere fp32 multi-workgroup")
run(sphere(100000).double(), "sphere fp64 multi-workgroup")
b = ok[:2000].clone().to(dev)
t0 = time.perf_counter(); fu._points_stepwise(b, True, 0); torch.cuda.synchronize()
print(f"step-wise fallback: N=2000 {(time.perf_counter()-t0)*1e3:.1f} ms -> {(time.perf_counter()-t0)/2000*1e6:.1f} us/step")
