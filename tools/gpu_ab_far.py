#!/usr/bin/env python3
"""A/B of the pair-kernel variants on the bench workload (100k sphere, 256 patches, cloud sorted by patch):
    far     the product library: scalar-unit kernel + far-field (one-transcendental) chain
    exact   the same kernel built with -DDNP_FAR=0 (exact chain everywhere)
    lds     the LDS-staged kernel of round 1 (-DDNP_FORCE_LDS=1)
plus any variant listed in AB_VARIANTS="name=flags;...", interleaved launches in ONE process on ONE device (cdna
guide rule 24), and the accuracy of each against the fp64 oracle on a row sample.

    python tools/gpu_ab_far.py            (on the GPU box; builds the variants under tools/bin/ first if missing)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.workloads import fibonacci_patches, sphere_cloud  # noqa: E402
from dipole_normal_prop_amd import _lib, build, util  # noqa: E402
from oracle import c_oracle  # noqa: E402

NOFAR = os.path.join(ROOT, "tools", "bin", "libdnp_nofar.so")


def bind(path):
    lib = ctypes.CDLL(path)
    for fn in ("dnp_patch_fields_f32", "dnp_patch_fields_boxed_f32", "dnp_patch_boxes_f32", "dnp_patch_fields_tiled_f32",
               "dnp_tile_boxes_f32", "dnp_interactions_from_tiles", "dnp_interactions_f32"):
        if hasattr(lib, fn):
            res, args = _lib.SIGNATURES[fn]
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    return lib


def main():
    build.build(verbose=False)
    libs = {"far": bind(build.LIB)}
    extra = "exact=-DDNP_FAR=0;lds=-DDNP_FORCE_LDS=1;" + os.environ.get("AB_VARIANTS", "")
    for item in [v for v in extra.split(";") if v]:
        name, flags = item.split("=", 1)
        path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
        if not os.path.exists(path):
            build.build(extra_flags=flags.split(), out=path, verbose=False)
        libs[name] = bind(path)
    for item in [v for v in os.environ.get("AB_LIBS", "").split(";") if v]:       # ready-made libraries: name=path
        name, path = item.split("=", 1)
        libs[name] = bind(path if os.path.isabs(path) else os.path.join(ROOT, path))
    dev = torch.device("cuda:0")
    pc = sphere_cloud()
    patches = fibonacci_patches(pc)
    if os.environ.get("AB_MORTON"):           # experiment: points of a patch in Morton order (compact waves of targets)
        def morton(x):
            q = ((x - x.min(0).values) / (x.max(0).values - x.min(0).values + 1e-12) * 1023).long().clamp(0, 1023)
            def spread(v):
                v = (v | (v << 16)) & 0x030000FF
                v = (v | (v << 8)) & 0x0300F00F
                v = (v | (v << 4)) & 0x030C30C3
                v = (v | (v << 2)) & 0x09249249
                return v
            return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        patches = [p[torch.argsort(morton(pc[p, :3]))] for p in patches]
    off, idx, sizes = util.patch_csr(patches, dev)
    pts = pc.to(dev)[idx].contiguous()
    N, P = pts.shape[0], len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    libs["nobox"] = libs["far"]       # run-time variants of the product library: no patch-box table,
    libs["tiled"] = libs["far"]       # + target-tile box table,
    libs["tiledw"] = libs["far"]      # + interaction partials in the epilogue
    n_tiles = (N + 127) // 128
    tile_boxes = torch.empty((n_tiles, 6), dtype=torch.float32, device=dev)
    assert libs["far"].dnp_tile_boxes_f32(_lib.ptr(pts), N, 6, 128, _lib.ptr(tile_boxes),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    w_part = torch.empty((P, n_tiles, 2), dtype=torch.float64, device=dev)
    boxes = torch.empty((P, 6), dtype=torch.float32, device=dev)
    assert libs["far"].dnp_patch_boxes_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(boxes),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    dE = {k: torch.empty((P, N, 3), dtype=torch.float32, device=dev) for k in libs}
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(name):
        if name == "nobox":          # the product library without the box table: every workgroup finds its patch's box
            rc = libs[name].dnp_patch_fields_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch), 0, P,
                                                 1e-5, _lib.ptr(dE[name]), stream)
        elif name in ("tiled", "tiledw") or name.startswith("tw"):     # AB_LIBS names tw*: a library through the tabled entry + partials
            rc = libs[name].dnp_patch_fields_tiled_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch),
                                                       _lib.ptr(boxes), _lib.ptr(tile_boxes), 0, P, 1e-5,
                                                       _lib.ptr(dE[name]), _lib.ptr(w_part) if name != "tiled" else None,
                                                       2, 1, None, 0, stream)
        else:
            rc = libs[name].dnp_patch_fields_boxed_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch),
                                                       _lib.ptr(boxes), 0, P, 1e-5, _lib.ptr(dE[name]), stream)
        assert rc == 0

    for name in libs:
        for _ in range(3):
            launch(name)
    torch.cuda.synchronize()
    times = {k: [] for k in libs}
    for rnd in range(24):
        for name in (list(libs) if rnd % 2 == 0 else list(libs)[::-1]):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            launch(name)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b))
    for name, ts in times.items():
        ts = np.array(ts)
        print(f"{name:6s} median {np.median(ts):.3f} ms  min {ts.min():.3f}  max {ts.max():.3f}  "
              f"({1e10 / np.median(ts) / 1e9:.3f} Tpairs/s, roofline frac {33e10 / (np.median(ts) * 1e-3) / 157.3e12:.3f})")
    print(f"far / exact = {np.median(times['far']) / np.median(times['exact']):.4f}")
    for name in ("nobox", "tiled", "tiledw"):
        print(f"{name} slabs bit-identical to far: {bool(torch.equal(dE[name], dE['far']))}")
    # the fused interaction matrix against the K3 pass over the slabs
    W3 = torch.empty((P, P), dtype=torch.float64, device=dev)
    Wt = torch.empty((P, P), dtype=torch.float64, device=dev)
    assert libs["far"].dnp_interactions_f32(_lib.ptr(dE["far"]), P, N, _lib.ptr(pts), 6, _lib.ptr(off), None, P, _lib.ptr(W3), stream) == 0
    assert libs["far"].dnp_interactions_from_tiles(_lib.ptr(w_part), 2, P, N, _lib.ptr(point_patch), _lib.ptr(off), P, _lib.ptr(Wt), stream) == 0
    torch.cuda.synchronize()
    print(f"W from tiles vs K3: max |diff| / max |W| = {float((Wt - W3).abs().max() / W3.abs().max()):.2e}")
    for label, fn in (("K3 dnp_interactions_f32", lambda: libs["far"].dnp_interactions_f32(_lib.ptr(dE["far"]), P, N, _lib.ptr(pts), 6, _lib.ptr(off), None, P, _lib.ptr(W3), stream)),
                      ("dnp_interactions_from_tiles", lambda: libs["far"].dnp_interactions_from_tiles(_lib.ptr(w_part), 2, P, N, _lib.ptr(point_patch), _lib.ptr(off), P, _lib.ptr(Wt), stream))):
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print(f"{label}: median {np.median(ts):.1f} us  min {min(ts):.1f}")

    # accuracy on a row sample: summed field of all patches (= all-pairs field minus the own-patch part) vs fp64
    rows = np.arange(0, N, 997)
    cpu = pts.cpu().numpy()
    pp = point_patch.cpu().numpy()
    for name in libs:
        tot = dE[name][:, rows].double().sum(dim=0).cpu().numpy()
        ref = np.zeros_like(tot)
        for i, r in enumerate(rows):
            others = cpu[pp != pp[r]]
            ref[i] = c_oracle.field_grad_f64(others, cpu[r:r + 1])[0]
        err = np.linalg.norm(tot - ref, axis=1) / np.linalg.norm(ref, axis=1)
        print(f"{name:6s} summed slabs vs fp64 oracle on {len(rows)} rows: median {np.median(err):.2e}  max {err.max():.2e}")
    # per-slab comparison far vs exact: one far patch and the nearest ones
    d = (dE["far"] - dE["exact"]).double().norm(dim=-1)
    n = dE["exact"].double().norm(dim=-1).clamp(min=1e-300)
    rel = (d / n)
    print(f"far vs exact per slab row: max rel {float(rel.max()):.2e}, mean {float(rel.mean()):.2e}; "
          f"fraction of slab rows bitwise equal {float((d == 0).double().mean()):.3f}")


if __name__ == "__main__":
    main()
