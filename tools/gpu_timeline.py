#!/usr/bin/env python3
"""Where does a pair-kernel launch spend its time?  A -DDNP_STAMP build (tools/bin/libdnp_stamp.so: every workgroup
leaves its start / end time, 100 MHz wall clock) runs
    patch mode  K = 32 and K = 256 patches of the bench workload (100k sphere, patch-sorted),
    fandisk all-pairs through dnp_field_grad_f32 (BASELINE config 2),
and prints for each launch: when the first / last workgroup starts, how the number of workgroups in flight evolves
(time at >= 90 % / 50 % / 10 % of the peak), the distribution of workgroup lifetimes, and how much of the launch is
ramp-up and drain.

    python tools/gpu_timeline.py          (on the GPU box; the stamp library is built here if missing)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import _lib, build, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

STAMP = os.path.join(ROOT, "tools", "bin", os.environ.get("STAMP_LIB", "libdnp_stamp.so"))


def bind():
    if not os.path.exists(STAMP):
        build.build(extra_flags=["-DDNP_STAMP=1"], out=STAMP, verbose=False)
    lib = ctypes.CDLL(STAMP)
    for fn in ("dnp_patch_fields_tiled_f32", "dnp_tile_boxes_f32", "dnp_patch_boxes_f32", "dnp_field_grad_f32",
               "dnp_field_grad_workspace_bytes"):
        res, args = _lib.SIGNATURES[fn]
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    for fn in ("dnp_debug_set_stamps_patch", "dnp_debug_set_stamps_field"):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = ctypes.c_int, [ctypes.c_void_p]
    return lib


def report(label, stamps, n_wg, slots):
    st = stamps[:n_wg].cpu().numpy().astype(np.int64)
    t0, t1 = st[:, 0], st[:, 1]
    base = t0.min()
    t0, t1 = (t0 - base) * 1e-2, (t1 - base) * 1e-2           # 100 MHz ticks -> us
    total = t1.max()
    life = t1 - t0
    # workgroups in flight over time
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    inflight = np.cumsum(ev[:, 1])
    peak = inflight.max()
    times = ev[:, 0]

    def first_at(frac):
        return float(times[np.argmax(inflight >= frac * peak)])

    def last_at(frac):
        idx = np.nonzero(inflight >= frac * peak)[0][-1]
        return float(times[min(idx + 1, len(times) - 1)])
    busy = float(np.trapezoid(inflight, times)) / (peak * total)
    print(f"## {label}: {n_wg} workgroups, {slots} resident slots ({n_wg / slots:.2f} sets)")
    print(f"   launch {total:.1f} us from the first workgroup's start; last workgroup starts at {t0.max():.1f} us")
    print(f"   in flight: peak {int(peak)}; >= 90 % of the peak from {first_at(0.9):.1f} to {last_at(0.9):.1f} us, "
          f">= 50 % until {last_at(0.5):.1f} us, >= 10 % until {last_at(0.1):.1f} us")
    print(f"   workgroup lifetime: median {np.median(life):.1f} us, 10 % {np.percentile(life, 10):.1f}, 90 % "
          f"{np.percentile(life, 90):.1f}, max {life.max():.1f}; first-set lifetime {np.median(life[t0 < 5.0]):.1f}, "
          f"last-finishing 5 % lifetime {np.median(life[t1 > np.percentile(t1, 95)]):.1f}")
    print(f"   slot-time used / (peak x launch) = {busy:.3f}; ideal launch at full peak occupancy "
          f"{busy * total:.1f} us -> ramp + drain cost {total - busy * total:.1f} us")


def main():
    lib = bind()
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stamps = torch.zeros((1 << 17, 2), dtype=torch.int64, device=dev)
    assert lib.dnp_debug_set_stamps_patch(ctypes.c_void_p(stamps.data_ptr())) == 0
    assert lib.dnp_debug_set_stamps_field(ctypes.c_void_p(stamps.data_ptr())) == 0

    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr(patches, dev)
    pts = pc.to(dev)[idx].contiguous()
    N, P = pts.shape[0], len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes = torch.empty((P, 6), dtype=torch.float32, device=dev)
    assert lib.dnp_patch_boxes_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(boxes), stream) == 0
    n_tiles = (N + 127) // 128
    tile_boxes = torch.empty((n_tiles, 6), dtype=torch.float32, device=dev)
    assert lib.dnp_tile_boxes_f32(_lib.ptr(pts), N, 6, 128, _lib.ptr(tile_boxes), stream) == 0
    dE = torch.empty((P, N, 3), dtype=torch.float32, device=dev)
    w_part = torch.empty((P, n_tiles, 2), dtype=torch.float64, device=dev)
    for K, ss in ((32, 1), (32, 4), (256, 1)):
        for rep in range(4):                                    # the last repetition is reported (warm clocks)
            stamps.zero_()
            rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch),
                                                _lib.ptr(boxes), _lib.ptr(tile_boxes), 0, K, 1e-5, _lib.ptr(dE),
                                                _lib.ptr(w_part), ss, stream)
            assert rc == 0
            torch.cuda.synchronize()
        tiles_x = -(-N // (512 // ss))
        report(f"patch mode, K = {K} patches, source_split {ss}", stamps, tiles_x * K, 2048)

    fd = torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]).to(dev)
    n = fd.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.dnp_field_grad_workspace_bytes(n, n, 15000), dtype=torch.uint8, device=dev)
    for rep in range(6):
        stamps.zero_()
        rc = lib.dnp_field_grad_f32(_lib.ptr(fd), n, 6, None, _lib.ptr(fd), n, 6, None, 1e-5, 15000, _lib.ptr(out), 3, 0, 0,
                                    None, None, _lib.ptr(ws), ws.numel(), stream)
        assert rc == 0
        torch.cuda.synchronize()
    n_wg = int((stamps[:, 1] != 0).sum())
    report("fandisk all-pairs (LDS kernel, KT = 1)", stamps, n_wg, 2048)


if __name__ == "__main__":
    main()
