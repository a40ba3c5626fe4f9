#!/usr/bin/env python3
"""A/B of build variants of the blocked xie ordered propagation (DNP_LIB-free: binds the variant libraries directly).
    XIE_VARIANTS="name=flags;..." python tools/gpu_xie_block_ab.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib, build  # noqa: E402

def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("dnp_xie_order_blocked_f32", "dnp_xie_order_blocked_f64", "dnp_xie_order_workspace_bytes"):
        res, args = _lib.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib

build.build(verbose=False)
libs = {"product": bind(build.LIB)}
for item in [v for v in os.environ.get("XIE_VARIANTS", "").split(";") if v]:
    name, flags = item.split("=", 1)
    path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
    build.build(extra_flags=flags.split(), out=path, verbose=False)
    libs[name] = bind(path)
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 10000
for dt, f64 in ((torch.float32, False), (torch.float64, True)):
    M = (torch.rand(n, n, generator=torch.Generator().manual_seed(1), dtype=torch.float32) - 0.5).to(dt).to(dev)
    for R in (1, 5):
        order = torch.from_numpy(np.stack([np.random.default_rng(s).permutation(n) for s in range(R)]).astype(np.int64)).to(dev)
        ref = None
        for name, lib in libs.items():
            w, it = torch.empty((R, n), dtype=dt, device=dev), torch.empty((R, n), dtype=dt, device=dev)
            nb = lib.dnp_xie_order_workspace_bytes(n, R, 8 if f64 else 4)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            fn = lib.dnp_xie_order_blocked_f64 if f64 else lib.dnp_xie_order_blocked_f32
            ts = []
            for rep in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); rc = fn(_lib.ptr(M), n, _lib.ptr(order), R, _lib.ptr(w), _lib.ptr(it), _lib.ptr(ws), nb, stream); b.record()
                torch.cuda.synchronize(); assert rc == 0
                if rep >= 3: ts.append(a.elapsed_time(b))
            same = None if ref is None else bool(torch.equal(w, ref))
            ref = w.clone() if ref is None else ref
            print(f"{str(dt):14s} R={R} {name:10s} median {np.median(ts):7.3f} ms  min {min(ts):7.3f}  same signs as product: {same}", flush=True)
