#!/usr/bin/env python3
"""A/B of per-point greedy kernel variants (K4) on ok.xyz (10 000 points, BASELINE config 1), one workgroup per CU:
    K4_VARIANTS="name=flags;..." python tools/gpu_k4_ab.py
Every variant must reproduce the reference's 10 000-step visit order (G8)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import _lib, build  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("dnp_point_greedy_f32", "dnp_point_greedy_f64", "dnp_point_greedy_workspace_bytes"):
        res, args = _lib.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


def main():
    build.build(verbose=False)
    libs = {"product": bind(build.LIB)}
    for item in [v for v in os.environ.get("K4_VARIANTS", "").split(";") if v]:
        name, flags = item.split("=", 1)
        path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
        if not os.path.exists(path):
            build.build(extra_flags=flags.split(), out=path, verbose=False)
        libs[name] = bind(path)
    dev = torch.device("cuda:0")
    g = load_golden("G8_point_propagation")
    cloud = torch.from_numpy(g["pc_full"]).to(dev)
    if os.environ.get("K4_N"):                   # a larger random sphere instead of ok.xyz (no golden order to compare with)
        n = int(os.environ["K4_N"])
        x = torch.randn(n, 6, generator=torch.Generator().manual_seed(3))
        cloud = torch.cat([0.5 * x[:, :3] / x[:, :3].norm(dim=1, keepdim=True), torch.nn.functional.normalize(x[:, 3:], dim=1)], 1).to(dev)
    N = cloud.shape[0]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, fn_name, esz in ((torch.float32, "dnp_point_greedy_f32", 4), (torch.float64, "dnp_point_greedy_f64", 8)):
        times = {k: [] for k in libs}
        ok = {}
        for rnd in range(7 if N <= 20000 else 3):
            for name in (list(libs) if rnd % 2 == 0 else list(libs)[::-1]):
                lib = libs[name]
                work = cloud.to(dtype).clone()
                order = torch.empty(N, dtype=torch.int64, device=dev)
                nbytes = lib.dnp_point_greedy_workspace_bytes(N, esz)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = getattr(lib, fn_name)(_lib.ptr(work), N, 6, 0, 1e-6, 1, _lib.ptr(order), None, 2, 0, _lib.ptr(ws), nbytes, stream)
                b.record()
                torch.cuda.synchronize()
                assert rc == 0
                times[name].append(a.elapsed_time(b))
                ok[name] = bool(np.array_equal(order.cpu().numpy(), g["order_full_d"])) if dtype == torch.float32 and not os.environ.get("K4_N") else None
        for name, ts in times.items():
            ts = np.array(ts[1:])
            print(f"{fn_name} {name:10s} median {np.median(ts):7.2f} ms  min {ts.min():7.2f}  ({np.median(ts) / N * 1e3:.2f} us/step)"
                  f"  reference order: {ok[name]}", flush=True)


if __name__ == "__main__":
    main()
