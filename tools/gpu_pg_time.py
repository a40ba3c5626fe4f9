#!/usr/bin/env python3
"""dnp_patch_greedy: the one-wavefront kernel against the one-workgroup kernel by patch count (HIP events).
    PG_VARIANTS="block=-DDNP_PG_BLOCK_FROM=0" python tools/gpu_pg_time.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib, build  # noqa: E402

def bind(path):
    lib = ctypes.CDLL(path)
    res, args = _lib.SIGNATURES["dnp_patch_greedy"]
    lib.dnp_patch_greedy.restype, lib.dnp_patch_greedy.argtypes = res, args
    return lib

build.build(verbose=False)
libs = {"product": bind(build.LIB)}
for item in [v for v in os.environ.get("PG_VARIANTS", "").split(";") if v]:
    name, flags = item.split("=", 1)
    path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
    if not os.path.exists(path):
        build.build(extra_flags=flags.split(), out=path, verbose=False)
    libs[name] = bind(path)
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(1)
for P in (72, 128, 256, 369, 512, 700, 1024, 1500, 2048, 4096, 8192, 16384):
    W0 = torch.randn(P, P, generator=g, dtype=torch.float64).to(dev)
    W = torch.empty_like(W0)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    res = {}
    for name, lib in libs.items():
        order = torch.empty(P, dtype=torch.int64, device=dev); sigma = torch.empty(P, dtype=torch.float64, device=dev)
        chosen = torch.empty(P, dtype=torch.float64, device=dev)
        ts = []
        for _ in range(4):
            W.copy_(W0)            # W freshly written by other CUs, as in the drivers (not warm in the greedy CU's caches)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.dnp_patch_greedy(_lib.ptr(W), P, _lib.ptr(start), _lib.ptr(order), _lib.ptr(sigma), _lib.ptr(chosen), stream)
            b.record(); torch.cuda.synchronize()
            assert rc == 0
            ts.append(a.elapsed_time(b))
        res[name] = (min(ts), order.clone())
    line = f"P={P:6d}  " + "  ".join(f"{k} {v[0] * 1e3:9.1f} us ({v[0] * 1e3 / P:5.2f} us/step)" for k, v in res.items())
    same = all(torch.equal(v[1], res["product"][1]) for v in res.values())
    print(line, " same order:", same, flush=True)
