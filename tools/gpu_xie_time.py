#!/usr/bin/env python3
"""The xie pair kernels against their roofline (HBM writes): xie_intersaction (T x S matrix, 4 B per pair) and xie_field
(T x S x 3, 12 B per pair) at N = 4 000 / 10 000 / 16 000 points, through the C ABI (dnp_xie_pairs_f32) with the output
buffer reused, and through the host mirror (which allocates the result); the ordered propagation per step.
    python tools/gpu_xie_time.py   (on the GPU box)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.require_device()
IEEE = os.path.join(ROOT, "tools", "bin", "libdnp_xie_ieee.so")      # -DDNP_XIE_IEEE_DIV=1: the compiler's own division per quotient
if not os.path.exists(IEEE):
    from dipole_normal_prop_amd import build
    build.build(extra_flags=["-DDNP_XIE_IEEE_DIV=1"], out=IEEE, verbose=False)
ieee = ctypes.CDLL(IEEE)
ieee.dnp_xie_pairs_f32.restype, ieee.dnp_xie_pairs_f32.argtypes = _lib.SIGNATURES["dnp_xie_pairs_f32"]
ieee.dnp_xie_pairs_f64.restype, ieee.dnp_xie_pairs_f64.argtypes = _lib.SIGNATURES["dnp_xie_pairs_f64"]
PLAIN = os.path.join(ROOT, "tools", "bin", "libdnp_xie_plain.so")
if not os.path.exists(PLAIN):
    from dipole_normal_prop_amd import build
    build.build(extra_flags=["-DDNP_XIE_ORDER_PLAIN=1"], out=PLAIN, verbose=False)
plain = ctypes.CDLL(PLAIN)
plain.dnp_xie_order_f32.restype, plain.dnp_xie_order_f32.argtypes = _lib.SIGNATURES["dnp_xie_order_f32"]
DEPTH2 = os.path.join(ROOT, "tools", "bin", "libdnp_xie_depth2.so")      # -DDNP_XIE_DEPTH=2: one row ahead, the form of rounds 3-4
if not os.path.exists(DEPTH2):
    from dipole_normal_prop_amd import build
    build.build(extra_flags=["-DDNP_XIE_DEPTH=2"], out=DEPTH2, verbose=False)
depth2 = ctypes.CDLL(DEPTH2)
depth2.dnp_xie_order_f32.restype, depth2.dnp_xie_order_f32.argtypes = _lib.SIGNATURES["dnp_xie_order_f32"]


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


g = torch.Generator().manual_seed(5)
for n in (4000, 10000, 16000):
    x = torch.randn(n, 3, generator=g)
    pc = torch.cat([x / x.norm(dim=1, keepdim=True), torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)], 1).to(dev)
    stream = _lib.current_stream()
    for vec, label in ((0, "matrix"), (1, "field ")):
        out = torch.empty((n, n, 3) if vec else (n, n), dtype=torch.float32, device=dev)
        med, mn = timed(lambda: lib.dnp_xie_pairs_f32(_lib.ptr(pc), n, 6, _lib.ptr(pc), n, 6, 3.0, vec, _lib.ptr(out), stream))
        gb = out.numel() * 4 / 1e9
        ref = torch.empty_like(out)
        assert ieee.dnp_xie_pairs_f32(_lib.ptr(pc), n, 6, _lib.ptr(pc), n, 6, 3.0, vec, _lib.ptr(ref), stream) == 0
        med_i, mn_i = timed(lambda: ieee.dnp_xie_pairs_f32(_lib.ptr(pc), n, 6, _lib.ptr(pc), n, 6, 3.0, vec, _lib.ptr(ref), stream))
        same = bool(torch.equal(out, ref))
        print(f"N={n:6d} xie {label} with the compiler's division per quotient: {mn_i * 1e3:8.1f} us min; results bit-identical: {same}")
        print(f"N={n:6d} xie {label} C ABI: {med * 1e3:8.1f} us median / {mn * 1e3:8.1f} min   {gb / mn * 1e3:7.1f} GB/s written "
              f"({gb * 1e3:.0f} MB), {n * n / mn / 1e6:.1f} Gpairs/s")
    if n <= 10000:      # fp64: the shared refined reciprocal (round 5) against the compiler's division, clustered cloud included
        for cname, c64 in (("sphere", pc.double()), ("clustered", torch.cat([pc[:n // 2].double() * 1e-3, pc[n // 2:].double()]).contiguous())):
            for vec, label in ((0, "matrix"), (1, "field ")):
                if vec and n > 4000:
                    continue
                out = torch.empty((n, n, 3) if vec else (n, n), dtype=torch.float64, device=dev)
                ref = torch.empty_like(out)
                med, mn = timed(lambda: lib.dnp_xie_pairs_f64(_lib.ptr(c64), n, 6, _lib.ptr(c64), n, 6, 3.0, vec, _lib.ptr(out), stream), reps=10)
                med_i, mn_i = timed(lambda: ieee.dnp_xie_pairs_f64(_lib.ptr(c64), n, 6, _lib.ptr(c64), n, 6, 3.0, vec, _lib.ptr(ref), stream), reps=10)
                print(f"N={n:6d} fp64 xie {label} ({cname}): {mn * 1e3:8.1f} us min | compiler's division per quotient {mn_i * 1e3:8.1f} us; "
                      f"results bit-identical: {bool(torch.equal(out, ref))}", flush=True)
                del out, ref
    med, mn = timed(lambda: fu.xie_intersaction(pc, pc, 0.1, -1, 3), reps=10)
    print(f"N={n:6d} fu.xie_intersaction (allocates the matrix): {med * 1e3:8.1f} us median / {mn * 1e3:8.1f} min")
    if n <= 10000:
        orders = np.stack([np.random.default_rng(i).permutation(n) for i in range(3)])
        med, mn = timed(lambda: fu.xie_propagation_points_in_order(pc, 0.1, orders, diffuse=False, knn_mask=-1, C=3), reps=5)
        print(f"N={n:6d} ordered propagation, 3 orders: {med:8.2f} ms median = {med * 1e3 / n:.2f} us per step (matrix included)")
        M = fu.xie_intersaction(pc, pc, 0.1, -1, 3).contiguous()
        ot = torch.from_numpy(orders.astype(np.int64)).to(dev)
        wts, itr = torch.empty((3, n), device=dev), torch.empty((3, n), device=dev)
        res = {}
        for name, L in (("product (weights in registers, three rows in flight, one barrier)", lib),
                        ("two rows in flight (-DDNP_XIE_DEPTH=2, rounds 3-4)", depth2), ("plain (-DDNP_XIE_ORDER_PLAIN=1)", plain)):
            L.dnp_xie_order_f32(_lib.ptr(M), n, _lib.ptr(ot), 3, _lib.ptr(wts), _lib.ptr(itr), stream)
            torch.cuda.synchronize()
            res[name] = (wts.clone(), itr.clone())
            med, mn = timed(lambda: L.dnp_xie_order_f32(_lib.ptr(M), n, _lib.ptr(ot), 3, _lib.ptr(wts), _lib.ptr(itr), stream), reps=5)
            print(f"N={n:6d}   dnp_xie_order_f32 {name}: {mn:8.3f} ms = {mn * 1e3 / n:.2f} us per step")
        (w1, i1), (w2, i2), (w3, i3) = res.values()
        print(f"N={n:6d}   weights and interactions bit-identical between the three forms: "
              f"{bool(torch.equal(w1, w2) and torch.equal(i1, i2) and torch.equal(w1, w3) and torch.equal(i1, i3))}")

# ---- round 5: the blocked form of the ordered propagation (dnp_xie_order_blocked_*) against the row-per-step kernels
for n in (1000, 4000, 10000, 16000):
    gen = torch.Generator().manual_seed(n + 1)
    for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        M = (torch.rand(n, n, generator=gen, dtype=torch.float32) - 0.5).to(dt).to(dev)
        for R in (1, 5):
            order_t = torch.from_numpy(np.stack([np.random.default_rng(s).permutation(n) for s in range(R)]).astype(np.int64)).to(dev)
            w1, i1 = torch.empty((R, n), dtype=dt, device=dev), torch.empty((R, n), dtype=dt, device=dev)
            w2, i2 = torch.empty_like(w1), torch.empty_like(i1)
            f64 = dt == torch.float64
            seq = lib.dnp_xie_order_f64 if f64 else lib.dnp_xie_order_f32
            blk = lib.dnp_xie_order_blocked_f64 if f64 else lib.dnp_xie_order_blocked_f32
            nbytes = lib.dnp_xie_order_workspace_bytes(n, R, 8 if f64 else 4)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ms, _ = timed(lambda: seq(_lib.ptr(M), n, _lib.ptr(order_t), R, _lib.ptr(w1), _lib.ptr(i1), _lib.current_stream()), reps=5)
            mb, _ = timed(lambda: blk(_lib.ptr(M), n, _lib.ptr(order_t), R, _lib.ptr(w2), _lib.ptr(i2), _lib.ptr(ws), nbytes, _lib.current_stream()), reps=5)
            print(f"N={n:6d} {name} {R} order(s): row-per-step {ms:8.3f} ms ({ms * 1e3 / n:5.2f} us per step) | blocked {mb:8.3f} ms "
                  f"({mb * 1e3 / n:5.3f} us per step, {ms / mb:4.1f}x) | same signs: {bool(torch.equal(w1, w2))}", flush=True)
        del M

# ---- round 5: the diffuse pass (dnp_xie_rowdots_*, one pass over M for all orders) against the torch matmul it replaced, the
# kNN selection (dnp_xie_knn_*) and the float64 forms
for n in (4000, 10000):
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, 6, generator=gen)
    pc = torch.cat([0.4 * x[:, :3] / x[:, :3].norm(dim=1, keepdim=True), torch.nn.functional.normalize(x[:, 3:], dim=1)], 1).to(dev)
    for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        pcd = pc.to(dt)
        M = fu.xie_intersaction(pcd, pcd, 0.1, -1, 3).contiguous()
        w = torch.where(torch.rand(5, n, device=dev) < 0.5, -1.0, 1.0).to(dt)
        out = torch.empty_like(w)
        fn = lib.dnp_xie_rowdots_f64 if dt == torch.float64 else lib.dnp_xie_rowdots_f32
        med, mn = timed(lambda: fn(_lib.ptr(M), n, _lib.ptr(w), 5, _lib.ptr(out), _lib.current_stream()))
        medt, mnt = timed(lambda: torch.matmul(w, M.transpose(0, 1)))
        gb = n * n * M.element_size() / 1e9
        print(f"N={n:6d} {name} diffuse pass, 5 orders: dnp_xie_rowdots {mn * 1e3:8.1f} us min ({gb / mn * 1e3:6.0f} GB/s of the matrix) | "
              f"torch w @ M.T (rocBLAS) {mnt * 1e3:8.1f} us")
    # the kNN forms: dnp_xie_knn (k-th nearest target per source, fp64 distances) + the masked pair matrix, against the torch
    # top-k mask of rounds 1-4 (4.8 ms at N = 10 000, k = 20 - plus a [T,S,3] tensor times an fp64 [T,S] mask behind it)
    kd = torch.empty(n, dtype=torch.float64, device=dev)
    ki = torch.empty(n, dtype=torch.int64, device=dev)
    for k in (5, 20, 50):
        med, mn = timed(lambda: lib.dnp_xie_knn_f32(_lib.ptr(pc), n, 6, _lib.ptr(pc), n, 6, k, _lib.ptr(kd), _lib.ptr(ki), _lib.current_stream()), reps=10)
        print(f"N={n:6d} dnp_xie_knn_f32 k = {k:2d}: {mn * 1e3:8.1f} us min / {med * 1e3:8.1f} median")

    def topk_mask():
        d2 = ((pc[:, None, :3].double() - pc[None, :, :3].double()) ** 2).sum(dim=-1) if n <= 4000 else None
        if d2 is None:
            return None
        return d2.topk(20, dim=1, largest=False).indices
    if n <= 4000:
        med, mn = timed(topk_mask, reps=5)
        print(f"N={n:6d} torch fp64 distances + topk(20) (the selection of rounds 1-4): {mn * 1e3:8.1f} us min")
    med, mn = timed(lambda: fu.xie_intersaction(pc, pc, 0.1, 20, 3), reps=10)
    med0, mn0 = timed(lambda: fu.xie_intersaction(pc, pc, 0.1, -1, 3), reps=10)
    print(f"N={n:6d} xie_intersaction with knn_mask = 20: {mn * 1e3:8.1f} us min (selection + masked matrix) | without a mask {mn0 * 1e3:8.1f} us")
    med, mn = timed(lambda: fu.xie_propagation_points_in_order(pc.double(), 0.1, [np.arange(n)], diffuse=True), reps=3)
    print(f"N={n:6d} f64 ordered propagation, 1 order, diffuse: {med:8.2f} ms = {med * 1e3 / n:.2f} us per step (matrix included)")
