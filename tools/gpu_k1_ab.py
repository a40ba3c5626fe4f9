#!/usr/bin/env python3
"""A/B of planning / kernel variants of the generic field_grad entry point (dnp_field_grad_f32: pair kernel + reduce)
on fandisk all-pairs (BASELINE config 2) and the 100k sphere, HIP events, interleaved in one process.
    K1_VARIANTS="name=flags;name=flags" python tools/gpu_k1_ab.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import _lib, build  # noqa: E402
from tools.gpu_check import sphere  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("dnp_field_grad_f32", "dnp_field_grad_workspace_bytes"):
        res, args = _lib.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


def main():
    build.build(verbose=False)
    libs = {"product": bind(build.LIB)}
    for item in [v for v in os.environ.get("K1_VARIANTS", "").split(";") if v]:
        name, flags = item.split("=", 1)
        path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
        if not os.path.exists(path):
            build.build(extra_flags=flags.split(), out=path, verbose=False)
        libs[name] = bind(path)
    for item in [v for v in os.environ.get("K1_LIBS", "").split(";") if v]:      # ready-made libraries: name=path
        name, path = item.split("=", 1)
        libs[name] = bind(path if os.path.isabs(path) else os.path.join(ROOT, path))
    dev = torch.device("cuda:0")
    clouds = {"fandisk": torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]).to(dev), "sphere3k": sphere(3000).to(dev),
              "sphere8k": sphere(8000).to(dev), "sphere16k": sphere(16000).to(dev), "sphere22k": sphere(22000).to(dev),
              "sphere30k": sphere(30000).to(dev), "sphere100k": sphere(100000).to(dev)}
    from tools.workloads import headline_workload
    from dipole_normal_prop_amd import util
    hpc, hpatches, _ = headline_workload()
    clouds["sorted100k"] = hpc.to(dev)[util.patch_csr(hpatches, dev)[1]].contiguous()   # the bench cloud, sorted by patch
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # rectangular shapes "S x T,S x T": the first S rows of the UNSORTED 100k sphere as sources, T gathered target rows
    for shape in [v for v in os.environ.get("K1_SHAPES", "").split(",") if v]:
        S, T = (int(x) for x in shape.split("x"))
        src = sphere(100000).to(dev)[:S].contiguous()
        tgt_idx = torch.randperm(100000, generator=torch.Generator().manual_seed(1))[:T].to(dev)
        full = sphere(100000).to(dev)
        out = torch.empty(T, 3, device=dev)
        ws = {k: torch.empty(lib.dnp_field_grad_workspace_bytes(S, T, 15000), dtype=torch.uint8, device=dev) for k, lib in libs.items()}
        times = {k: [] for k in libs}
        for rnd in range(24):
            for name in (list(libs) if rnd % 2 == 0 else list(libs)[::-1]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = libs[name].dnp_field_grad_f32(_lib.ptr(src), S, 6, None, _lib.ptr(full), T, 6, _lib.ptr(tgt_idx), 1e-5, 15000, _lib.ptr(out),
                                                   3, 0, 0, None, None, _lib.ptr(ws[name]), ws[name].numel(), stream)
                b.record()
                torch.cuda.synchronize()
                assert rc == 0
                if rnd >= 4:
                    times[name].append(a.elapsed_time(b))
        for name, ts in times.items():
            ts = np.array(ts) * 1e3
            print(f"{shape:14s} {name:12s} median {np.median(ts):9.1f} us  min {ts.min():9.1f}  ({S * T / np.median(ts) / 1e3:7.1f} Gpairs/s)", flush=True)
    only = [c for c in os.environ.get("K1_CLOUDS", "").split(",") if c]
    for cname, pc in clouds.items():
        if only and cname not in only:
            continue
        n = pc.shape[0]
        out = torch.empty(n, 3, device=dev)
        ws = {k: torch.empty(lib.dnp_field_grad_workspace_bytes(n, n, 15000), dtype=torch.uint8, device=dev) for k, lib in libs.items()}

        def launch(name):
            rc = libs[name].dnp_field_grad_f32(_lib.ptr(pc), n, 6, None, _lib.ptr(pc), n, 6, None, 1e-5, 15000, _lib.ptr(out), 3, 0, 0,
                                               None, None, _lib.ptr(ws[name]), ws[name].numel(), stream)
            assert rc == 0
        for name in libs:
            for _ in range(3):
                launch(name)
        torch.cuda.synchronize()
        times = {k: [] for k in libs}
        reps = 30 if n < 50000 else 8
        for rnd in range(reps):
            for name in (list(libs) if rnd % 2 == 0 else list(libs)[::-1]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); launch(name); b.record()
                torch.cuda.synchronize()
                times[name].append(a.elapsed_time(b))
        # the same calls back to back (what bench.py's legs time): 30 launches between two events, three rounds, best round
        b2b = {}
        for name in libs:
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30):
                    launch(name)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 30)
            b2b[name] = best * 1e3
        for name, ts in times.items():
            ts = np.array(ts) * 1e3
            print(f"{cname:11s} {name:12s} median {np.median(ts):9.1f} us  min {ts.min():9.1f}  ({n * n / np.median(ts) / 1e3:7.1f} Gpairs/s)  "
                  f"back to back {b2b[name]:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
