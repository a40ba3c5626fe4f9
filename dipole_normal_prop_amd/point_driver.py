"""The per-point greedy driver of the reference's field_utils on the device library: strongest_field_propagation_points
(field_utils.py:353-388) - one persistent launch of dnp_point_greedy_* (single workgroup or one workgroup per CU, fp32 / fp64) and
the step-wise fallback beyond the kernels' capacity or after a time-out.  Split out of field_utils.py in round 5; the public name
and the POINT_GREEDY_* knobs are re-exported there - tests that patch a knob do it on this module."""
import ctypes

import numpy as np
import torch

from . import _lib
from . import util
from ._staging import _compute_device, _ld, _on_device, _pairs_into, _set_trace, _stage, _work_dtype  # noqa: F401
from .patch_drivers import _store_normals


def strongest_field_propagation_points(pts: torch.Tensor, diffuse=False, starting_point=0, verbose=False):
    """Per-point greedy orientation (field_utils.py:353-388): N-1 sequential steps of
    `E += field of the chosen point (eps=1e-6)`, `argmax |E.n|` over unvisited points, flip.
    Runs as one persistent kernel (K4) in the cloud's own precision: float64 clouds (the reference's socket
    path, util.py:71-77) are propagated in fp64, everything else in fp32.  Normals are updated in place when
    pts already lives on the device (the reference's own in-place contract, orient_simple.py:24 relies on it,
    holds for CPU tensors here as well); returns pts.  Trace: last_trace("points")."""
    lib = _lib.require_device()
    with torch.no_grad():
        dev = pts.device if pts.is_cuda else _compute_device()
        wd = torch.float64 if pts.dtype == torch.float64 else torch.float32
        work = pts.detach().to(device=dev, dtype=wd).contiguous()
        if work.data_ptr() == pts.data_ptr():
            work = work.clone()
        N = work.shape[0]
        order = torch.empty(N, dtype=torch.int64, device=dev)
        done = False
        if N < lib.dnp_point_greedy_max_points() and N <= POINT_GREEDY_MAX_PER_GROUP[wd] * _cu_count(dev):
            # one persistent launch: a single workgroup for small clouds, one workgroup per CU beyond ~1800 points
            nbytes = lib.dnp_point_greedy_workspace_bytes(N, work.element_size())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            fn = lib.dnp_point_greedy_f64 if wd == torch.float64 else lib.dnp_point_greedy_f32
            with _on_device(dev):
                rc = fn(_lib.ptr(work), N, work.stride(0), int(starting_point), 1e-6, int(bool(diffuse)),
                        _lib.ptr(order), None, int(POINT_GREEDY_FORM), int(POINT_GREEDY_GROUPS), _lib.ptr(ws), nbytes,
                        _lib.current_stream())
            _lib.check(rc)
            done = True
            if int(ws[:4].view(torch.int32).item()) != 0:
                # a workgroup of the multi-workgroup form gave up waiting for its peers (GPU shared with another
                # process): the kernel left pts untouched; redo the propagation step by step
                print("warning: persistent per-point kernel timed out, falling back to step-wise launches")
                done = False
        if not done:
            order = _points_stepwise(work, diffuse, int(starting_point))
        _store_normals(pts, work[:, 3:])
        _set_trace("points", order=order)
        return pts


# form / workgroup cap handed to dnp_point_greedy_* (0 = let the library choose); tests pin them to cover both forms
POINT_GREEDY_FORM = 0
POINT_GREEDY_GROUPS = 0
POINT_GREEDY_MAX_PER_GROUP = {torch.float32: 256 * 20, torch.float64: 256 * 8}      # threads per workgroup x points per thread (csrc/dnp_greedy.hip: multi_threads, GreedyCap)
_cu_cache = {}


def _cu_count(dev) -> int:
    n = _cu_cache.get(dev.index)
    if n is None:
        n = _cu_cache[dev.index] = min(256, torch.cuda.get_device_properties(dev).multi_processor_count)
    return n


def _points_stepwise(work, diffuse, start):
    """Fallback for clouds beyond the persistent kernel's capacity: the loop of
    field_utils.py:361-380 with one single-source field launch per step."""
    dev = work.device
    N = work.shape[0]
    E = torch.zeros((N, 3), dtype=work.dtype, device=dev)
    visited = torch.zeros(N, dtype=torch.bool, device=dev)
    order = torch.empty(N, dtype=torch.int64, device=dev)
    cur = start
    for step in range(N):
        visited[cur] = True
        order[step] = cur
        _pairs_into("field", work[cur:cur + 1], None, work, None, 1e-6, 0, E, accumulate=True)
        if step + 1 == N:
            break
        inter = (E * work[:, 3:]).sum(dim=-1)
        mag = torch.where(visited, torch.full_like(inter, -1.0), inter.abs())
        cur = int(mag.argmax().item())
        if float(inter[cur]) < 0:
            work[cur, 3:] *= -1
    if diffuse:
        s = ((E * work[:, 3:]).sum(dim=-1) > 0).to(work.dtype) * 2 - 1
        work[:, 3:] = work[:, 3:] * s[:, None]
    return order
