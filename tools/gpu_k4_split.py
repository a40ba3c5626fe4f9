#!/usr/bin/env python3
"""Where a step of the multi-workgroup per-point kernel goes, fp32 against fp64 (ok.xyz, 10 000 points; -DDNP_K4_STATS build:
workgroup 0 stamps the publish of its candidate and the end of every step with the 100 MHz wall clock): own work (field update +
workgroup argmax, up to the publish) and the rest (all-gather of the candidates, the winner's row, the wave argmax, hand-over).
    python tools/gpu_k4_split.py [extra build flags]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import _lib, build  # noqa: E402

dev = torch.device("cuda:0")
ok = torch.from_numpy(load_golden("G8_point_propagation")["pc_full"])
N = ok.shape[0]
path = os.path.join(ROOT, "tools", "bin", "libdnp_k4split.so")
build.build(extra_flags=["-DDNP_K4_STATS"] + sys.argv[1:], out=path, verbose=False)
slib = ctypes.CDLL(path)
for name in ("dnp_point_greedy_f32", "dnp_point_greedy_f64", "dnp_point_greedy_workspace_bytes"):
    res, args = _lib.SIGNATURES[name]
    getattr(slib, name).restype, getattr(slib, name).argtypes = res, args
slib.dnp_debug_set_k4_stats.argtypes = [ctypes.c_void_p]
stats = torch.zeros(N * 4 + 256 * 2, dtype=torch.int64, device=dev)
slib.dnp_debug_set_k4_stats(_lib.ptr(stats))
for dtype in (torch.float32, torch.float64, torch.float32, torch.float64):
    work = ok.to(dtype).to(dev).clone()
    order = torch.empty(N, dtype=torch.int64, device=dev)
    ws = torch.empty(int(slib.dnp_point_greedy_workspace_bytes(N, work.element_size())), dtype=torch.uint8, device=dev)
    fn = slib.dnp_point_greedy_f64 if dtype == torch.float64 else slib.dnp_point_greedy_f32
    stats.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(_lib.ptr(work), N, 6, 0, 1e-6, 1, _lib.ptr(order), None, 2, 0, _lib.ptr(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    b.record()
    torch.cuda.synchronize()
    assert rc == 0
    st = stats.cpu().numpy().astype(np.uint64)
    per = st[: (N - 1) * 4].reshape(N - 1, 4)
    wall, spins, pub = per[:, 0].astype(np.float64), per[:, 2].astype(np.float64), per[:, 3].astype(np.float64)
    step_ns = np.diff(wall) * 10.0
    own_ns = (pub[1:] - wall[:-1]) * 10.0
    wait_ns = (wall[1:] - pub[1:]) * 10.0
    print(f"{str(dtype):14s} {a.elapsed_time(b):6.2f} ms | step ns p50 {np.percentile(step_ns, 50):.0f} p90 {np.percentile(step_ns, 90):.0f} | "
          f"own work (field update + workgroup argmax, to the publish) p50 {np.percentile(own_ns, 50):.0f} ns | gather + row + wave argmax + hand-over p50 "
          f"{np.percentile(wait_ns, 50):.0f} ns | slowest lane's spins per step {spins.mean():.1f}", flush=True)
