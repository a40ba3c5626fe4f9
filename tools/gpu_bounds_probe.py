#!/usr/bin/env python3
"""Runs under a -DDNP_BOUNDS build of the library (DNP_LIB=tools/bin/libdnp_bounds.so; tests/test_gpu_bounds.py starts it as a
subprocess): (1) the shape that faulted in round 3 - a cloud whose tile count leaves target-less wavefronts in the last
workgroup (N = 300 + 84: 3 tiles of 128 rows, two-wavefront workgroups -> the 4th wavefront has no tile), in the plain form
and with the split tail (four-wavefront workgroups), box tables given; (2) tools/gpu_fuzz.py for `seconds`; then prints
the counters of the check build as one JSON line per run: (sum of the eight out-of-bounds counters, each counter by table + the
number of (slab, tile) items that broke w_part's two-group precondition - the fuzz passes w_part for ANY cut and ignores it
where the cut does not allow it, so that one is expected to be nonzero there).
    DNP_LIB=tools/bin/libdnp_bounds.so python tools/gpu_bounds_probe.py [fuzz seconds = 12] [seed]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402

NAMES = ["chunk_off", "chunk_box", "tile_box", "tgt_group", "w_part", "partial", "exchange", "source_range", "two_group_precondition"]


def counters(lib, reset):
    lib.dnp_debug_bounds_errors.restype = ctypes.c_longlong
    lib.dnp_debug_bounds_errors.argtypes = [ctypes.c_void_p, ctypes.c_int]
    h = (ctypes.c_uint * 16)()
    total = lib.dnp_debug_bounds_errors(h, int(reset))
    return int(total), {n: int(v) for n, v in zip(NAMES, h)}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20251005
    lib = _lib.require_device()
    if not hasattr(lib, "dnp_debug_bounds_errors"):
        raise SystemExit("this library is not a -DDNP_BOUNDS build")
    dev = torch.device("cuda:0")
    out = {}
    # (1) the deterministic shape: 384 rows = 3 tiles; patches of 129 / 150 / 105 rows (the last one a single run)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(384, 3, generator=g)
    pc = torch.cat([x / x.norm(dim=1, keepdim=True) * 0.4, x / x.norm(dim=1, keepdim=True)], 1).to(dev)
    sizes = np.array([129, 150, 105], dtype=np.int64)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)])).to(dev)
    pp = torch.repeat_interleave(torch.arange(3, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(pc, off, None), fu._TileTables(pc, sizes)
    counters(lib, True)
    plain = fu._patch_slabs(pc, off, None, pp, 0, 3, 1e-5, boxes, tiles.boxes, None, 1)
    wp = torch.zeros((3, tiles.n_tiles, 2), dtype=torch.float64, device=dev)
    fu._patch_slabs(pc, off, None, pp, 0, 3, 1e-5, boxes, tiles.boxes, wp, 1)
    out["round3_shape_plain"] = counters(lib, True)
    tail = fu._patch_slabs(pc, off, None, pp, 0, 3, 1e-5, boxes, tiles.boxes, wp, -2)
    out["round3_shape_split_tail"] = counters(lib, True)
    out["round3_shape_results_equal"] = bool(torch.equal(plain, tail))
    # (2) the fuzz slice under the check build
    if seconds > 0:
        from tools import gpu_fuzz
        cases, fails = gpu_fuzz.run(budget=seconds, seed=seed)
        out["fuzz"] = {"cases": cases, "failures": fails[:5], "bounds": counters(lib, True)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
