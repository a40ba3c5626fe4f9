#!/usr/bin/env python3
"""Where does a pair-kernel launch spend its time?  A -DDNP_STAMP build (tools/bin/libdnp_stamp.so) lets every
wavefront leave its end time (100 MHz wall clock) and the SIMD it ran on (HW_ID, XCC_ID) through the scalar unit; the
LDS kernel also leaves its start time.  (The scalar kernel cannot: any side effect in its prologue costs it 11-17 VGPRs,
see pair_kernel.h - with end stamps alone the stamp build has the product's register allocation, 60/61 VGPRs.)

    patch mode  K = 32 (one rank's share of 8) and K = 256 patches of the bench workload, source_split 1 and 4:
                the DRAIN - from the moment the queue is empty (the (W - 8192)-th wavefront end: until then every end is
                followed by a start) the occupancy of SIMD x at time t is the number of its wavefronts ending after t -
                as SIMD-time spent at occupancy 0, 1, 2, ... in whole-chip microseconds; the rate of wavefront ends over
                the launch (steady rate -> what the launch would take without head and drain)
    fandisk all-pairs through dnp_field_grad_f32 (BASELINE config 2, LDS kernel): start / end per wavefront

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
SLOTS_PER_SIMD, SIMDS = 8, 1024


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


def simd_key(word):
    hw, xcc = word & 0xffffffff, (word >> 32) & 0xf
    # gfx9 HW_ID: wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
    return (xcc << 12) | (((hw >> 8) & 0xff) << 2) | ((hw >> 4) & 3)


def report_drain(label, stamps, launch_us):
    st = stamps.cpu().numpy()
    st = st[st[:, 1] != 0]
    W = len(st)
    end = (st[:, 1] - st[:, 1].min()).astype(np.float64) * 1e-2          # us, from the first wavefront end
    key = simd_key(st[:, 2].astype(np.int64))
    simds = np.unique(key)
    last = end.max()
    order = np.sort(end)
    slots = SLOTS_PER_SIMD * SIMDS
    print(f"## {label}: {W} wavefronts on {len(simds)} SIMDs ({W / slots:.2f} resident sets); launch {launch_us:.1f} us by HIP "
          f"events; first wavefront end {launch_us - last:.1f} us after the launch began, last = the launch's end")
    # rate of wavefront ends, 20 us bins
    bins = np.arange(0.0, last + 20.0, 20.0)
    hist, _ = np.histogram(end, bins)
    mid = hist[len(hist) // 4: 3 * len(hist) // 4]
    if len(mid) >= 3:
        rate = mid.mean() / 20.0
        print(f"   wavefront ends per us: middle half of the end window {rate:.1f} -> {W} wavefronts at that rate {W / rate:.1f} us "
              f"(launch - that = {launch_us - W / rate:.1f} us of head + drain)")
    if W <= slots:
        print("   (a single resident set: no queue)")
        t_qe = 0.0
    else:
        t_qe = order[W - slots - 1]
    print(f"   queue empty {last - t_qe:.1f} us before the end of the launch (the {W - slots}-th end)")
    # occupancy of every SIMD after the queue is empty
    occ_time = np.zeros(SLOTS_PER_SIMD + 6)
    for k in simds:
        e = np.sort(end[(key == k) & (end > t_qe)])
        n = len(e)
        t_prev = t_qe
        for i, t in enumerate(e):                      # between t_prev and t the SIMD holds n - i wavefronts
            occ_time[min(n - i, len(occ_time) - 1)] += t - t_prev
            t_prev = t
        occ_time[0] += last - t_prev
    window = (last - t_qe) * len(simds)
    chip = occ_time / len(simds)
    print("   SIMD-time in that window by occupancy, in whole-chip us: " +
          "  ".join(f"{i}: {chip[i]:.1f}" for i in range(len(chip)) if chip[i] >= 0.05) +
          f"   (window {window / len(simds):.1f})")
    for n_sat in (2, 3, 4):
        lost = sum(chip[i] * (1.0 - min(1.0, i / n_sat)) for i in range(len(chip)))
        print(f"   issue slots lost in the drain if {n_sat} wavefronts saturate a SIMD: {lost:.1f} us")
    per_simd_last = np.array([end[key == k].max() for k in simds])
    print(f"   a SIMD's last wavefront ends {np.median(last - per_simd_last):.1f} us (median) / {np.percentile(last - per_simd_last, 90):.1f} us "
          f"(90 %) before the launch does; wavefronts per SIMD min {np.bincount(np.searchsorted(simds, key)).min()} max "
          f"{np.bincount(np.searchsorted(simds, key)).max()}")


def report_lds(label, stamps, slots):
    st = stamps.cpu().numpy()
    st = st[st[:, 1] != 0]
    t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) * 1e-2, (t1 - base) * 1e-2
    total = t1.max()
    life = t1 - t0
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    inflight = np.cumsum(ev[:, 1])
    peak = inflight.max()
    times = ev[:, 0]

    def last_at(frac):
        idx = np.nonzero(inflight >= frac * peak)[0][-1]
        return float(times[min(idx + 1, len(times) - 1)])
    busy = float(np.trapezoid(inflight, times)) / (peak * total)
    print(f"## {label}: {len(st)} wavefronts, {slots} resident slots ({len(st) / slots:.2f} sets)")
    print(f"   launch {total:.1f} us from the first wavefront's start; last wavefront starts at {t0.max():.1f} us")
    print(f"   in flight: peak {int(peak)}; >= 90 % of the peak until {last_at(0.9):.1f} us, >= 50 % until {last_at(0.5):.1f} us, "
          f">= 10 % until {last_at(0.1):.1f} us")
    print(f"   wavefront lifetime: median {np.median(life):.1f} us, 10 % {np.percentile(life, 10):.1f}, 90 % "
          f"{np.percentile(life, 90):.1f}, max {life.max():.1f}")
    print(f"   slot-time used / (peak x launch) = {busy:.3f}; ideal launch at full peak occupancy {busy * total:.1f} us -> "
          f"ramp + drain cost {total - busy * total:.1f} us")


def main():
    lib = bind()
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stamps = torch.zeros((1 << 19, 4), dtype=torch.int64, device=dev)
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
    for K, ss in ((32, 1), (64, 1), (256, 1)):
        wgs = -(-N // (256 if ss == 1 else 128)) * K
        assert wgs * 4 <= stamps.shape[0]
        ms = []
        for rep in range(24 if K < 256 else 8):                 # the last repetition is reported (warm clocks)
            stamps.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(pts), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch),
                                                _lib.ptr(boxes), _lib.ptr(tile_boxes), 0, K, 1e-5, _lib.ptr(dE),
                                                _lib.ptr(w_part), 2, ss, None, 0, stream)
            b.record()
            assert rc == 0
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        print(f"# K = {K}, source_split {ss}: launch ms by events, last 8 repetitions: " + " ".join(f"{m:.4f}" for m in ms[-8:]))
        report_drain(f"patch mode, K = {K} patches, source_split {ss}", stamps, ms[-1] * 1e3)

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
    report_lds("fandisk all-pairs (LDS kernel, KT = 1)", stamps, 8192)


if __name__ == "__main__":
    main()
