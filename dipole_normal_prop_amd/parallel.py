"""Multi-GPU sharding of the per-patch field evaluations (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The path shards by PATCH: field_grad is linear in the dipoles and a flip negates a whole patch, so
the field dE_k of every patch on every other point can be evaluated before the greedy order is
known (include/dnp.h, dnp_patch_fields_f32).  Every rank holds the full cloud (2.4 MB at 100 k
points), evaluates a contiguous, size-balanced block of patches and its rows of the P x P
interaction matrix W; ONE all-gather of the W rows (P*P*8 bytes = 512 KB at P = 256) gives every
rank what the sequential greedy loop needs (it runs redundantly on every rank as the one-wavefront kernel
dnp_patch_greedy: same W, deterministic).  For the diffuse per-point pass the sign vector sigma[P] is then
known on every rank, each rank combines its own slabs in fp64, and one all-reduce (N*24 bytes) sums the
partial fields; the start patch is rank 0's (8-byte broadcast).  No other collective is on the path.

Reference lines: the loop being distributed is field_utils.py:308-335.
"""
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _staging


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves host memory only: device tensors are staged through the CPU (rehearsals of the N>1
    path on a single-GPU box); with nccl (RCCL) the collective runs on the device buffers over xGMI."""
    return t.is_cuda and dist.get_backend() == "gloo"


def gather_rows(W_local: torch.Tensor, bounds: np.ndarray, group=None) -> torch.Tensor:
    """All-gather the per-rank row blocks of W ([rows_r, P] fp64, rows_r = bounds[r+1]-bounds[r])
    into the full [P, P] matrix on every rank.  Blocks are padded to the largest block so that a
    single fixed-size all-gather (one RCCL call) moves everything."""
    rank, size = world()
    if size == 1:
        return W_local
    P = W_local.shape[1]
    rows = np.diff(bounds)
    if int(rows.min()) == int(rows.max()) and not _host_staged(W_local):
        # equal blocks (e.g. 256 patches over 8 ranks): gather straight into the final matrix
        out = torch.empty((int(rows.sum()), P), dtype=W_local.dtype, device=W_local.device)
        dist.all_gather_into_tensor(out, W_local.contiguous(), group=group)
        return out
    pad = int(rows.max())
    buf = torch.zeros((pad, P), dtype=W_local.dtype, device=W_local.device)
    buf[: W_local.shape[0]] = W_local
    if _host_staged(buf):
        parts = [torch.empty((pad, P), dtype=W_local.dtype) for _ in range(size)]
        dist.all_gather(parts, buf.cpu(), group=group)
        out = torch.stack(parts).to(W_local.device)
    else:
        out = torch.empty((size, pad, P), dtype=W_local.dtype, device=W_local.device)
        dist.all_gather_into_tensor(out.view(size * pad, P), buf, group=group)
    return torch.cat([out[r, : int(rows[r])] for r in range(size)], dim=0)


def gather_rows_async(W_local: torch.Tensor, bounds: np.ndarray, group=None, force: bool = False):
    """gather_rows issued WITHOUT making the caller's stream wait: returns (W, work).  W is complete once work.wait() has
    been called (a stream-side wait, the host does not block) - until then the collective runs on RCCL's own stream, behind
    the kernels that produced W_local, and whatever the caller launches next overlaps it (a caller that evaluates batch
    after batch: the all-gather of one batch under the pair kernel of the next).  Falls back to the synchronous form
    (work = None) where the backend stages through the host (gloo), for unequal blocks, and for a single rank
    (`force`: run the one-rank collective anyway - the probe of this code path on a one-GPU box)."""
    rank, size = world()
    rows = np.diff(bounds)
    if ((size == 1 and not force) or not dist.is_initialized() or _host_staged(W_local) or not W_local.is_cuda
            or int(rows.min()) != int(rows.max())):
        return gather_rows(W_local, bounds, group), None
    out = torch.empty((int(rows.sum()), W_local.shape[1]), dtype=W_local.dtype, device=W_local.device)
    work = dist.all_gather_into_tensor(out, W_local.contiguous(), group=group, async_op=True)
    return out, work


def reduce_field(E_partial: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank partial fields (each rank combined its own slabs, in fp64) on every rank.  The partial
    sums are fp64 sums of +-1-signed fp32 slabs, so the total is independent of the split over ranks up to
    fp64 reassociation (~1e-16 relative): after the single rounding to fp32 the field - and every sign taken
    from it - is the one a single GPU computes, except when a component lies within that distance of a
    rounding boundary (probability ~1e-9 per component)."""
    _, size = world()
    if size > 1:
        if _host_staged(E_partial):
            host = E_partial.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            E_partial.copy_(host)
        else:
            dist.all_reduce(E_partial, op=dist.ReduceOp.SUM, group=group)
    return E_partial


def agree_on_start(start_t: torch.Tensor, group=None) -> torch.Tensor:
    """Every rank uses rank 0's start patch.  The device PCA is deterministic, so the ranks compute the same
    index anyway; the broadcast (8 bytes) makes the agreement unconditional."""
    _, size = world()
    if size > 1:
        if _host_staged(start_t):
            host = start_t.cpu()
            dist.broadcast(host, src=0, group=group)
            start_t.copy_(host)
        else:
            dist.broadcast(start_t, src=0, group=group)
    return start_t


def _finish_sharded(fu, pts, work, w, st, patches, diffuse, listed, start_t):
    """Tail of a sharded propagation once the greedy loop has run: the all-reduce of the partial fields (diffuse) and the
    stores into the caller's tensor."""
    if diffuse and st.Es is not None:
        st.Es = reduce_field(st.Es)                 # patch-sorted rows: the same permutation on every rank
    if not diffuse or listed is not None:
        fu._finish_batched(pts, st, diffuse, listed, w)
    else:
        flip = torch.where(st.point_patch >= 0, st.sigma[st.point_patch.clamp(min=0)], 1.0).to(work.dtype)
        work[:, 3:] = work[:, 3:] * flip[:, None]
        fu._diffuse_sign_pass(work, st.field().to(work.dtype), [patch for _, patch in patches])
        fu._finish_patch_driver(pts, work, w)
    fu._set_trace("sharded", order=st.order, sigma=st.sigma, chosen=st.chosen, start=start_t)


def sharded_patch_propagation(pts: torch.Tensor, patches, all_patches, diffuse=False, weights=None,
                              start_patch: Optional[int] = None):
    """strongest_field_propagation (field_utils.py:286-348) with the per-patch fields sharded over
    the ranks of the default process group.  Every rank must call it with the same arguments; every
    rank ends with the same oriented normals in `pts` (in place).  Trace: field_utils.last_trace("sharded").

    The collectives are issued in the launch stream's order: the greedy loop needs ALL of W, so for one cloud there is
    nothing to overlap the all-gather with (sharded_patch_propagation_many does, across clouds)."""
    from . import patch_drivers as fu          # the drivers' own module: where their helpers are looked up (and patched by tests)

    rank, size = world()
    with torch.no_grad():
        if len(all_patches) == 0:
            return
        work, w = fu._prepare_work(pts, weights)
        start_t = agree_on_start(fu._start_tensor(work, all_patches, start_patch))
        listed = fu._listed_patches(patches, all_patches, work.device) if diffuse else None
        st = fu._batched_patch_propagation(work, all_patches, start_t, diffuse, shard=(rank, size, gather_rows))
        _finish_sharded(fu, pts, work, w, st, patches, diffuse, listed, start_t)


def sharded_patch_propagation_many(jobs, diffuse=False, force_async=False):
    """sharded_patch_propagation for SEVERAL clouds, pipelined: `jobs` is a sequence of (pts, patches, all_patches) or
    (pts, patches, all_patches, weights, start_patch) tuples, every rank passes the same sequence, every `pts` is oriented in
    place.  Returns the list of traces (order / sigma / chosen / start as numpy arrays), one per job IN THE JOBS' ORDER - None
    for a job without patches (nothing to propagate), so the list lines up with the caller's jobs.

    One cloud cannot hide its all-gather - the greedy loop needs all of W - but a queue of clouds can: the W rows of cloud
    i are gathered on the collective library's own stream (gather_rows_async) while this rank's pair kernel of cloud i + 1
    runs, and cloud i's greedy loop, combine and tail follow behind it.  The start patches of all jobs are agreed on with
    ONE broadcast up front (a per-job broadcast would queue behind the asynchronous gather of the next job and pull it onto
    the critical path); the diffuse form's all-reduce of the partial fields stays in stream order.  Results are those of
    sharded_patch_propagation job by job (bit for bit); where the backend stages through the host (gloo) or blocks are
    unequal, the gather falls back to the in-order form and only the launch order differs.  force_async: take the
    asynchronous path with a ONE-rank group too (the rehearsal of that path on a one-GPU box; tests)."""
    from . import patch_drivers as fu          # the drivers' own module: where their helpers are looked up (and patched by tests)

    rank, size = world()
    with torch.no_grad():
        norm = []
        for job in jobs:
            pts, patches, all_patches = job[0], job[1], job[2]
            weights = job[3] if len(job) > 3 else None
            start_patch = job[4] if len(job) > 4 else None
            norm.append((pts, patches, all_patches, weights, start_patch))
        live_ids = [i for i, j in enumerate(norm) if len(j[2]) > 0]
        live = [norm[i] for i in live_ids]
        traces = [None] * len(norm)
        if not live:
            return traces
        prepared = [fu._prepare_work(pts, weights) for pts, _, _, weights, _ in live]
        starts = torch.cat([fu._start_tensor(work, all_patches, start_patch).reshape(1)
                            for (work, _), (_, _, all_patches, _, start_patch) in zip(prepared, live)])
        starts = agree_on_start(starts)

        def finish(item):
            i, bw, W, handle = item
            pts, patches, all_patches, _, _ = live[i]
            work, w = prepared[i]
            if handle is not None:
                handle.wait()                       # a stream-side wait: the host does not block
            start_t = starts[i:i + 1]
            st = fu._batched_end(bw, W, start_t)
            listed = fu._listed_patches(patches, all_patches, work.device) if diffuse else None
            _finish_sharded(fu, pts, work, w, st, patches, diffuse, listed, start_t)
            traces[live_ids[i]] = _staging.last_trace("sharded")

        pending = None
        for i, (pts, patches, all_patches, _, _) in enumerate(live):
            bw = fu._batched_begin(prepared[i][0], all_patches, diffuse, rank=rank, world=size)
            W, handle = gather_rows_async(bw.W_local, bw.bounds, force=force_async)
            if pending is not None:
                finish(pending)
            pending = (i, bw, W, handle)
        finish(pending)
    return traces
