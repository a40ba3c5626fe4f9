"""Multi-GPU sharding of the per-patch field evaluations (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The path shards by PATCH: field_grad is linear in the dipoles and a flip negates a whole patch, so
the field dE_k of every patch on every other point can be evaluated before the greedy order is
known (include/dnp.h, dnp_patch_fields_f32).  Every rank holds the full cloud (2.4 MB at 100 k
points), evaluates a contiguous, size-balanced block of patches and its rows of the P x P
interaction matrix W; ONE all-gather of the W rows (P*P*8 bytes = 512 KB at P = 256) gives every
rank what rank 0 needs to run the whole sequential greedy loop as host arithmetic.  For the
diffuse per-point pass the sign vector sigma[P] is then known on every rank (same W, same
deterministic loop), each rank combines its own slabs, and one all-reduce (N*12 bytes) sums the
partial fields.  No other collective is on the path.

Reference lines: the loop being distributed is field_utils.py:308-335.
"""
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


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


def reduce_field(E_partial: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank partial fields (each rank combined its own slabs) on every rank."""
    _, size = world()
    if size > 1:
        if _host_staged(E_partial):
            host = E_partial.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            E_partial.copy_(host)
        else:
            dist.all_reduce(E_partial, op=dist.ReduceOp.SUM, group=group)
    return E_partial


def sharded_patch_propagation(pts: torch.Tensor, patches, all_patches, diffuse=False, weights=None,
                              start_patch: Optional[int] = None):
    """strongest_field_propagation (field_utils.py:286-348) with the per-patch fields sharded over
    the ranks of the default process group.  Every rank must call it with the same arguments; every
    rank ends with the same oriented normals in `pts` (in place)."""
    from . import field_utils as fu

    rank, size = world()
    with torch.no_grad():
        if len(all_patches) == 0:
            return
        work, w = fu._prepare_work(pts, weights)
        dev = work.device
        if start_patch is None:
            start_patch = fu._flattest_patch(work, [p.to(dev) for p in all_patches])
        order, sigma, chosen, E, point_patch = fu._batched_patch_propagation(
            work, list(all_patches), int(start_patch), diffuse, shard=(rank, size, gather_rows))
        if diffuse and E is not None:
            E = reduce_field(E)
        sig = torch.tensor(sigma, dtype=torch.float32, device=dev)
        flip = torch.ones(work.shape[0], dtype=torch.float32, device=dev)
        inpatch = point_patch >= 0
        flip[inpatch] = sig[point_patch[inpatch]]
        work[:, 3:] = work[:, 3:] * flip[:, None]
        if diffuse:
            fu._diffuse_sign_pass(work, E, [patch for _, patch in patches])
        if w is not None:
            work[:, 3:] = work[:, 3:] / w[:, None]
        pts[:, 3:] = work[:, 3:].to(device=pts.device, dtype=pts.dtype)
        sharded_patch_propagation.last_trace = dict(order=order, sigma=sigma, chosen=chosen, start=int(start_patch))
