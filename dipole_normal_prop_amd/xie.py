"""The fork's "xie" pair functions of the reference's field_utils: xie_field / xie_intersaction / xie_distance
(field_utils.py:431-526), the ordered sign propagation xie_propagation_points_in_order (:569-605) and the BFS-route propagation
with its vote, xie_propagation_points_onbfstree (:657-710) - same names, argument order and defaults; the pair matrix, the ordered
loop, its diffuse pass and the kNN mask run in csrc/dnp_xie.hip behind the C ABI (dnp_xie_pairs_*, dnp_xie_knn_*, dnp_xie_pairs_knn_*,
dnp_xie_order_blocked_*, dnp_xie_rowdots_*), in the
cloud's own precision (float64 clouds in float64).  Split out of field_utils.py in round 5; every public name is re-exported there."""
import ctypes

import numpy as np
import torch

from . import _lib
from . import util
from ._staging import _compute_device, _ld, _on_device, _set_trace, _stage, _work_dtype, _workspace


def _xie_pairs(source, target, C, vector_out, knn_mask=-1):
    """dnp_xie_pairs on staged tensors; knn_mask > 0: dnp_xie_knn (the k-th nearest target of every source, fp64 distances on
    the exact coordinates as the reference's KDTree computes them, field_utils.py:451-460) and dnp_xie_pairs_knn (entries
    outside the k nearest multiplied by 0, :467-468) - no T x S mask tensor."""
    lib = _lib.require_device()
    if source.dim() != 2 or source.shape[1] < 6 or target.dim() != 2 or target.shape[1] < 6:
        raise ValueError("xie pair functions need [S,6] sources and [T,6] targets")
    in_dev, in_dtype = target.device, torch.result_type(source, target)
    dev = source.device if source.is_cuda else (target.device if target.is_cuda else _compute_device())
    wd = _work_dtype(source, target)
    src = _stage(source.detach(), dev, wd)
    tgt = _stage(target.detach(), dev, wd)
    S, T = src.shape[0], tgt.shape[0]
    out = torch.empty((T, S, 3) if vector_out else (T, S), dtype=wd, device=dev)
    if S and T:
        f64 = wd == torch.float64
        with _on_device(dev):
            if knn_mask > 0:
                k = min(T, int(knn_mask))                                  # k = min(len(xyz), knn_mask), :456
                kth_d2 = torch.empty(S, dtype=torch.float64, device=dev)
                kth_idx = torch.empty(S, dtype=torch.int64, device=dev)
                _lib.check((lib.dnp_xie_knn_f64 if f64 else lib.dnp_xie_knn_f32)(
                    _lib.ptr(src), S, _ld(src), _lib.ptr(tgt), T, _ld(tgt), k, _lib.ptr(kth_d2), _lib.ptr(kth_idx),
                    _lib.current_stream()))
                rc = (lib.dnp_xie_pairs_knn_f64 if f64 else lib.dnp_xie_pairs_knn_f32)(
                    _lib.ptr(src), S, _ld(src), _lib.ptr(tgt), T, _ld(tgt), float(C), int(vector_out), _lib.ptr(kth_d2),
                    _lib.ptr(kth_idx), _lib.ptr(out), _lib.current_stream())
            else:
                rc = (lib.dnp_xie_pairs_f64 if f64 else lib.dnp_xie_pairs_f32)(
                    _lib.ptr(src), S, _ld(src), _lib.ptr(tgt), T, _ld(tgt), float(C), int(vector_out), _lib.ptr(out),
                    _lib.current_stream())
        _lib.check(rc)
    return out.to(device=in_dev, dtype=in_dtype) if (out.device != in_dev or out.dtype != in_dtype) else out


def xie_field(source: torch.Tensor, target: torch.Tensor, eps, max_pts=5000, knn_mask=-1, C=3):
    """Reflected-normal pair field [T,S,3] (field_utils.py:431-469): (n_s - C (n_s.r^) r^)/|r|^3 with
    r = x_s - x_t, left undivided for coincident pairs; optionally masked to the knn_mask nearest targets of
    each source.  `eps` and `max_pts` are accepted and unused (the reference ignores eps; its recursion above
    max_pts**2 pairs only bounds temporaries - and drops C / knn_mask on the way, which this does not)."""
    with torch.no_grad():
        return _xie_pairs(source, target, C, True, knn_mask)


def xie_intersaction(source: torch.Tensor, target: torch.Tensor, eps, knn_mask, C):
    """[T,S] interaction matrix xie_field . n_t with NaN/Inf zeroed (field_utils.py:509-519)."""
    with torch.no_grad():
        return _xie_pairs(source, target, C, False, knn_mask)


def xie_distance(source: torch.Tensor, target: torch.Tensor, eps):
    """sum_s |n_s * (x_s - x_t)| per target (field_utils.py:522-526); O(T*S*3) torch temporaries as there."""
    R = source[None, :, :3] - target[:, None, :3]
    return (source[None, :, 3:] * R).norm(dim=-1).sum(dim=-1)


def xie_propagation_points_in_order(pts: torch.Tensor, eps, order, diffuse=False, verbose=False, points_weight=None,
                                    knn_mask=-1, C=3):
    """Ordered sign propagation (field_utils.py:569-605): for each of the T visiting orders in `order[T,N]`,
    visit the points in that order, give every point the sign of the summed interaction with the points
    visited before it, and return the [T,N] bool tensor `interactions < 0` (True = flipped).  `points_weight`
    is accepted and has no effect, as in the reference (it multiplies by a tensor of ones)."""
    lib = _lib.require_device()
    with torch.no_grad():
        dev = pts.device if pts.is_cuda else _compute_device()
        # the reference computes in pts.dtype (interactions / weights are created with .type(pts.dtype), :581-586): a float64
        # cloud (its socket path, socket_server_para.py:68-83) gets a float64 matrix and float64 row sums
        wd = torch.float64 if pts.dtype == torch.float64 else torch.float32
        work = pts.detach().to(device=dev, dtype=wd).contiguous()
        order_t = torch.as_tensor(np.asarray(order)).to(device=dev, dtype=torch.int64).contiguous()
        T, N = order_t.shape
        M = xie_intersaction(work, work, eps, knn_mask, C).to(wd).contiguous()     # [N, N]
        # no initialisation needed: the kernels zero `inter` themselves (an order row that repeats an index leaves points
        # unvisited, and the reference's interactions start as torch.zeros) and write every weight
        weights = torch.empty((T, N), dtype=wd, device=dev)
        inter = torch.empty((T, N), dtype=wd, device=dev)
        f64 = wd == torch.float64
        # the blocked form (csrc/dnp_xie.hip): 256 steps per block - one HBM-bound launch for the block's row sums, one wavefront
        # per order for its dependent steps; rows of `order` that are not permutations go to the row-per-step kernel inside the call
        stream_handle = torch.cuda.current_stream(dev).cuda_stream
        ws = _workspace(lib.dnp_xie_order_workspace_bytes(N, T, 8 if f64 else 4), dev, stream_handle)
        with _on_device(dev):
            rc = (lib.dnp_xie_order_blocked_f64 if f64 else lib.dnp_xie_order_blocked_f32)(
                _lib.ptr(M), N, _lib.ptr(order_t), T, _lib.ptr(weights), _lib.ptr(inter), _lib.ptr(ws), ws.numel(),
                ctypes.c_void_p(stream_handle))
        _lib.check(rc)
        if diffuse:
            # interactions[t][i] = sum_j M[i][j] * w[t][j] (:597-603): one pass over M for all T weight vectors
            with _on_device(dev):
                rc = (lib.dnp_xie_rowdots_f64 if f64 else lib.dnp_xie_rowdots_f32)(_lib.ptr(M), N, _lib.ptr(weights), T,
                                                                                  _lib.ptr(inter), _lib.current_stream())
            _lib.check(rc)
        return (inter < 0).to(pts.device)


def align_votes(flips: torch.Tensor) -> torch.Tensor:
    """The vote alignment of field_utils.xie_propagation_points_onbfstree (field_utils.py:693-702), where the
    reference calls gurobi (MIQP, :620-646) on a problem with one binary variable per visiting order: choose
    x in {0,1}^T minimising  sum_ij [ H_ij if x_i == x_j else N - H_ij ]  with H_ij the Hamming distance between
    the flip vectors of orders i and j (cal_w, :674-677; cal_loss, :606-617).  T is a handful (the `times` of the
    caller, odd), so the optimum is found by enumerating all 2^(T-1) assignments with x_0 = 0 - the objective only
    depends on which x are equal, so x and its complement tie and x_0 = 0 picks one of the two; further ties go
    to the smallest assignment in binary order.  flips: [T, N] bool.  Returns x as a [T] bool tensor."""
    T, N = flips.shape
    if T > 20:
        raise ValueError(f"{T} visiting orders: the exhaustive vote is meant for the reference's handful of orders")
    f = flips.to(torch.float64)
    H = (f[:, None, :] - f[None, :, :]).abs().sum(dim=-1).cpu().numpy()           # [T, T] Hamming distances
    best_x, best_cost = 0, None
    for code in range(1 << max(T - 1, 0)):
        x = np.array([0] + [(code >> b) & 1 for b in range(T - 1)])
        same = x[:, None] == x[None, :]
        cost = float(np.where(same, H, N - H).sum())
        if best_cost is None or cost < best_cost:
            best_x, best_cost = x, cost
    return torch.from_numpy(np.asarray(best_x, dtype=bool)).to(flips.device)


def xie_propagation_points_onbfstree(pts: torch.Tensor, eps, diffuse=False, starting_point=0, verbose=False, k=10,
                                     treshold=0.1, times=1, use_pw=False, knn_mask=-1, C=3):
    """Propagation along breadth-first routes of the kNN graph with a vote over `times` routes
    (field_utils.py:657-710): routes start at starting_point and at times-1 further points drawn with
    np.random.seed(0) / randint as in the reference; every route is propagated in order
    (xie_propagation_points_in_order: one interaction matrix, one persistent workgroup per route), the routes'
    flip vectors are aligned (align_votes: the reference's MIQP, solved exactly without gurobi), and a point is
    flipped when more than half of the aligned routes flip it.  `pts` normals are updated in place; returns the
    [N] bool tensor of flipped points.  use_pw is accepted and has no effect (as points_weight in the reference)."""
    assert times % 2 == 1 and times > 0
    with torch.no_grad():
        starts = [int(starting_point)]
        np.random.seed(0)
        while len(np.unique(starts)) < times:
            cand = np.random.randint(0, pts.shape[0])
            if cand not in starts:
                starts.append(cand)
        adj, _ = util.knn_graph(pts[:, :3].detach().cpu().numpy(), k, treshold)
        orders = np.zeros((times, pts.shape[0]), dtype=np.int64)
        for i in range(times):
            orders[i] = util.bfs_route(adj, starts[i])
        flips = xie_propagation_points_in_order(pts.clone(), eps, orders, diffuse, verbose=False, knn_mask=knn_mask,
                                                C=C)                                   # [times, N]
        status = align_votes(flips)
        aligned = flips ^ status[:, None].to(flips.device)
        cnts = aligned.sum(dim=0)
        flipped = cnts > times / 2
        sel = flipped.to(pts.device)
        pts[sel, 3:] = pts[sel, 3:] * -1
        _set_trace("bfstree", orders=orders, flips=flips, status=status, starts=np.array(starts))
        return sel
