"""The greedy patch drivers of the reference's field_utils on the device library: strongest_field_propagation
(field_utils.py:286-348) and strongest_field_propagation_reps (:207-282) - the batched form (all per-patch fields in one launch with
the interaction partials out of the pair kernel's epilogue, device greedy loop, fp64 signed combine, fused tails), the literal
step-by-step form for overlapping patch lists, and the helpers parallel.py shards with (_batched_begin / _batched_end,
_balanced_blocks, _pick_source_split).  Split out of field_utils.py in round 5; every public name and every knob is re-exported
there for callers - tests and tools that PATCH a knob or a helper do it on this module, where the drivers look them up."""
from typing import List, Optional, Tuple

import ctypes

import numpy as np
import torch

from . import _lib
from . import util
from ._staging import (_NULL, _compute_device, _idx, _ld, _on_device, _pairs_into, _set_trace, _stage, _tls,  # noqa: F401
                       _work_dtype, _workspace)

__all__ = ["strongest_field_propagation", "strongest_field_propagation_reps", "greedy_order_from_interactions"]

# "auto": batched (all per-patch fields in one launch, greedy loop as P x P host arithmetic) when the
# patches are disjoint, else the step-by-step form; "sequential" / "batched" force one.
PATCH_MODE = "auto"
# device bytes the batched drivers may spend on the [P, N, 3] slab before they fall back to two passes
SLAB_BUDGET_BYTES = 48 << 30
# block size of the slab evaluation once the slabs do not fit that budget at once
SLAB_BLOCK_BYTES = 16 << 30
# slab sets up to this size are allocated without asking the driver how much memory is free (the query costs more
# than a small propagation)
SLAB_FREE_CHECK_BYTES = 1 << 30


def _flattest_patch(pts: torch.Tensor, patches) -> torch.Tensor:
    """argmin_k |lambda_min(cov(patch k))| (field_utils.py:303-306 / :230-233; util.pca_eigen_values per patch
    in the reference) as a 1-element int64 tensor on pts' device - no host round trip.  The covariances come
    from util.patch_pca (fp64, deterministic), the one shared implementation: among near-planar patches the
    reference's own fp32 choice is decided by BLAS rounding; fp64 picks the patch that is actually flattest,
    which is the reference's choice on every golden cloud (G6, G7, G13, G15)."""
    _, evals, _, _ = util.patch_pca(pts, patches)
    return torch.argmin(evals[:, 0].abs()).reshape(1)


def _start_tensor(work, patches, start_patch) -> torch.Tensor:
    """The start patch as a 1-element device tensor.  A Python int is range-checked here (IndexError, as the host
    loop greedy_order_from_interactions would raise); a TENSOR start is not synchronised on: the greedy kernels clamp
    an out-of-range value to patch 0 (documented in include/dnp.h)."""
    if start_patch is None:
        return _flattest_patch(work, patches)
    if isinstance(start_patch, torch.Tensor):
        return start_patch.to(device=work.device, dtype=torch.int64).reshape(1)
    if not 0 <= int(start_patch) < len(patches):
        raise IndexError(f"start_patch {int(start_patch)} out of range for {len(patches)} patches")
    return torch.tensor([int(start_patch)], dtype=torch.int64, device=work.device)


def _point_patch_ids(idx: torch.Tensor, sizes: np.ndarray, n: int) -> torch.Tensor:
    """patch id of every point (-1 = in no patch) from the CSR form of disjoint patches."""
    dev = idx.device
    point_patch = torch.full((n,), -1, dtype=torch.int64, device=dev)
    if idx.numel():
        point_patch[idx] = torch.repeat_interleave(torch.arange(len(sizes), device=dev),
                                                   util.to_device(sizes, dev), output_size=int(idx.numel()))
    return point_patch


def _listing_ids(lists_csr) -> torch.Tensor:
    """patch id of every entry of a CSR index list (device)."""
    off, idx, sizes = lists_csr
    return torch.repeat_interleave(torch.arange(len(sizes), device=idx.device), util.to_device(sizes, idx.device),
                                   output_size=int(idx.numel()))


def _flip_by_listing(work: torch.Tensor, neg: torch.Tensor, idx: torch.Tensor, pid: torch.Tensor) -> None:
    """work[idx, 3:] *= -1 once per listing of a point in a patch k with neg[k] (the reference's loop flips a
    point every time it is listed): parity of the listing count, all on the device.  pid = _listing_ids(...),
    built BEFORE the long kernels are enqueued: a host->device copy behind them would stall the host until
    they have finished."""
    if idx.numel() == 0:
        return
    cnt = torch.zeros(work.shape[0], dtype=torch.int32, device=work.device)
    cnt.index_add_(0, idx, neg[pid].to(torch.int32))
    s = (1 - 2 * (cnt % 2)).to(work.dtype)
    work[:, 3:] = work[:, 3:] * s[:, None]


def _diffuse_sign_pass(work: torch.Tensor, E: torch.Tensor, index_lists, listed: Optional[torch.Tensor] = None) -> None:
    """sign = (E.n > 0) * 2 - 1 applied to the normals of every listed point (field_utils.py:337-342,
    :267-271), all lists at once.  A point listed twice gets the same result as in the reference's loop:
    after the first visit E.n > 0, so the second visit multiplies by +1.  `listed` = a ready-made bool mask of
    the listed points (the drivers derive it from the patch ids when the lists are the patches themselves)."""
    if listed is None:
        if len(index_lists) == 0:
            return
        listed = torch.zeros(work.shape[0], dtype=torch.bool, device=work.device)
        if isinstance(index_lists, util.PatchList):
            listed[index_lists.flat.to(work.device)] = True
        else:
            listed[torch.cat([p.to(work.device) for p in index_lists])] = True
    pos = (E * work[:, 3:]).sum(dim=-1) > 0
    s = torch.where(listed & ~pos, -1.0, 1.0).to(work.dtype)
    work[:, 3:] = work[:, 3:] * s[:, None]


def _csr(patches: List[torch.Tensor], dev) -> Tuple[torch.Tensor, torch.Tensor]:
    off, idx, _ = util.patch_csr(patches, dev)
    return off, idx


def _disjoint(idx: torch.Tensor, n: int) -> bool:
    if idx.numel() == 0:
        return True
    return bool(torch.bincount(idx, minlength=n).max().item() <= 1)


def _patch_boxes(work: torch.Tensor, off, idx) -> torch.Tensor:
    """[P, 6] bounding boxes (min xyz, max xyz) of the patches' points (dnp_patch_boxes_f32 / _f64, in the cloud's precision):
    what the pair kernel's far-field test compares a wavefront's targets with; computed once per cloud."""
    lib = _lib.require_device()
    P = off.shape[0] - 1
    boxes = torch.empty((P, 6), dtype=work.dtype, device=work.device)
    fn = lib.dnp_patch_boxes_f64 if work.dtype == torch.float64 else lib.dnp_patch_boxes_f32
    with _on_device(work.device):
        rc = fn(_lib.ptr(work), work.shape[0], work.stride(0), _lib.ptr(off), _lib.ptr(idx), P, _lib.ptr(boxes),
                _lib.current_stream())
    _lib.check(rc)
    return boxes


# exchange buffers up to this size stay cached per (thread, device, stream); a larger one (12 416 bytes per split patch and
# 128-row tile: 29 MB for 3 split patches at 100 000 points, 780 MB for 8 at a million) is allocated zeroed for the call and freed
EXCHANGE_CACHE_MAX_BYTES = 256 << 20      # (12 416 bytes per split patch and 128-row tile)


def _exchange(nbytes: int, dev: torch.device) -> torch.Tensor:
    """Exchange buffer of the split forms (include/dnp.h, dnp_patch_fields_tiled_f32): zero at its first use, left zero
    by every launch that uses it; cached per (thread, device, stream) like the workspace, because one buffer serves one
    stream at a time.  A launch that FAILS may leave arrival counters behind: _exchange_drop forgets the cached buffer then
    (round-4 advisor), the next call gets a fresh zeroed one."""
    if nbytes > EXCHANGE_CACHE_MAX_BYTES:
        return torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    cache = getattr(_tls, "xch", None)
    if cache is None:
        cache = _tls.xch = {}
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = cache[key] = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    return buf


def _exchange_drop(dev: torch.device) -> None:
    cache = getattr(_tls, "xch", None)
    if cache is not None:
        cache.pop((dev.index, torch.cuda.current_stream(dev).cuda_stream), None)


def _patch_slabs(work: torch.Tensor, off, idx, point_patch, p0: int, p1: int, eps: float, boxes=None, tile_boxes=None,
                 w_part: Optional[torch.Tensor] = None, source_split: int = 1, order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dE[p1 - p0, N, 3]: the fields of patches p0..p1 on every point (dnp_patch_fields_tiled_f32 / _ordered_f32).  boxes /
    tile_boxes: the per-cloud box tables of the far-field test (_patch_boxes, _tile_boxes); w_part: receives the per-tile
    interaction partials [p1 - p0, n_tiles, 2 or 3] (see _TileTables; the last dimension = the group slots per tile).
    source_split = -k: the last k ROWS of the launch as split items whose run terms travel through the exchange buffer (needs
    both box tables; without them the launch is the plain one).  order (device int32 [p1 - p0], patch-sorted layout only): the
    patch, relative to p0, that launch row i evaluates - _launch_plan puts the longest patches first; slab k stays patch p0 + k."""
    lib = _lib.require_device()
    N = work.shape[0]
    dE = torch.empty((p1 - p0, N, 3), dtype=work.dtype, device=work.device)
    ordered = order is not None and idx is None and eps > 0 and p1 - p0 <= 65535
    if work.dtype == torch.float64:
        # a float64 cloud: double-precision slabs (the fp64 far chain when both box tables are given; no split tail)
        both = boxes is not None and tile_boxes is not None
        with _on_device(work.device):
            if ordered:
                rc = lib.dnp_patch_fields_ordered_f64(_lib.ptr(work), N, work.stride(0), _lib.ptr(off), off.shape[0] - 1,
                                                      _lib.ptr(point_patch), _lib.ptr(boxes if both else None),
                                                      _lib.ptr(tile_boxes if both else None), p0, p1, _lib.ptr(order), float(eps),
                                                      _lib.ptr(dE), _lib.ptr(w_part), 2 if w_part is None else int(w_part.shape[-1]),
                                                      _lib.current_stream())
            else:
                rc = lib.dnp_patch_fields_tiled_f64(_lib.ptr(work), N, work.stride(0), _lib.ptr(off), _lib.ptr(idx),
                                                    off.shape[0] - 1, _lib.ptr(point_patch), _lib.ptr(boxes if both else None),
                                                    _lib.ptr(tile_boxes if both else None), p0, p1, float(eps), _lib.ptr(dE),
                                                    _lib.ptr(w_part), 2 if w_part is None else int(w_part.shape[-1]),
                                                    _lib.current_stream())
        _lib.check(rc)
        return dE
    xch, xch_bytes = None, 0
    if source_split < 0 and boxes is not None and tile_boxes is not None:
        xch_bytes = int(lib.dnp_patch_exchange_bytes(N, min(-source_split, p1 - p0)))
        xch = _exchange(xch_bytes, work.device)
    with _on_device(work.device):
        if ordered:
            rc = lib.dnp_patch_fields_ordered_f32(_lib.ptr(work), N, work.stride(0), _lib.ptr(off), off.shape[0] - 1,
                                                  _lib.ptr(point_patch), _lib.ptr(boxes), _lib.ptr(tile_boxes), p0, p1, _lib.ptr(order),
                                                  float(eps), _lib.ptr(dE), _lib.ptr(w_part),
                                                  2 if w_part is None else int(w_part.shape[-1]), int(source_split),
                                                  _lib.ptr(xch), xch_bytes, _lib.current_stream())
        else:
            rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(work), N, work.stride(0), _lib.ptr(off), _lib.ptr(idx),
                                                off.shape[0] - 1, _lib.ptr(point_patch), _lib.ptr(boxes), _lib.ptr(tile_boxes),
                                                p0, p1, float(eps), _lib.ptr(dE), _lib.ptr(w_part),
                                                2 if w_part is None else int(w_part.shape[-1]), int(source_split),
                                                _lib.ptr(xch), xch_bytes, _lib.current_stream())
    if rc != 0 and xch is not None:
        _exchange_drop(work.device)               # its counters may not be re-armed: never reuse it
    _lib.check(rc)
    return dE


def _tile_group_slots(sizes, n_rows: int, rows_per_tile: int) -> int:
    """How many group slots per tile the pair kernel's interaction partials need on this patch-sorted cloud: 2 when every
    tile of `rows_per_tile` consecutive rows lies inside at most two groups (group = a patch; the rows behind the last patch
    form one more group), 3 when inside at most three CONSECUTIVE groups (round 5: patches of 64..127 points - the reference's
    grid partitions start at 100), 0 when neither (then W comes from the K3 pass over the slabs).  From the patch sizes alone
    (host, no sync): the group index of a tile's last row minus that of its first must be <= slots - 1; an empty patch
    between two patches of a tile makes the index jump and is refused, conservatively."""
    n_tiles = -(-n_rows // rows_per_tile)
    if n_tiles == 0:
        return 2
    sizes = np.asarray(sizes, dtype=np.int64)
    ends = np.cumsum(sizes)
    first = np.arange(n_tiles, dtype=np.int64) * rows_per_tile
    last = np.minimum(first + rows_per_tile, n_rows) - 1
    span = np.searchsorted(ends, last, side="right") - np.searchsorted(ends, first, side="right")
    worst = int(span.max())
    if worst <= 1:
        return 2
    if worst == 2 and not bool(np.any(sizes == 0)):
        return 3
    return 0


def _tiles_within_two_groups(sizes, n_rows: int, rows_per_tile: int) -> bool:
    """Does every tile lie inside at most two groups (the 2-slot form of the interaction partials)?"""
    return _tile_group_slots(sizes, n_rows, rows_per_tile) == 2


class _TileTables:
    """Per-cloud tables of the patch-sorted layout for the scalar-unit pair kernel: the boxes of its target tiles
    (tile i = sorted rows [i R, (i+1) R), R = dnp_patch_tile_rows() = the 128 targets one wavefront owns) and whether
    every tile lies inside at most two groups (patches; rows in no patch form the last group) - then the kernel's
    epilogue can leave the interaction sums per (slab, tile, group slot) and W needs no second pass over the slabs."""
    __slots__ = ("rows", "n_tiles", "boxes", "fused", "slots")

    def __init__(self, swork: torch.Tensor, sizes: np.ndarray):
        lib = _lib.require_device()
        N = swork.shape[0]
        self.rows = int(lib.dnp_patch_tile_rows())
        self.n_tiles = -(-N // self.rows)
        self.boxes = torch.empty((self.n_tiles, 6), dtype=swork.dtype, device=swork.device)      # in the cloud's precision
        fn = lib.dnp_tile_boxes_f64 if swork.dtype == torch.float64 else lib.dnp_tile_boxes_f32
        with _on_device(swork.device):
            _lib.check(fn(_lib.ptr(swork), N, swork.stride(0), self.rows, _lib.ptr(self.boxes), _lib.current_stream()))
        self.slots = _tile_group_slots(sizes, N, self.rows)        # 2 / 3 group slots per tile, 0 = not fusable
        self.fused = self.slots != 0


# Split tail of the pair kernel's launches (include/dnp.h, dnp_patch_fields_tiled_f32).  A launch ends with ~50 us of a
# chip that is emptying, and that tail scales with the item length (profiles/r03_timeline.txt); as a split item - four
# wavefronts on ONE target tile, one 128-source run of the patch each, the run terms added in run order by whichever
# arrives last (exchange buffer, no LDS) - a (tile, patch) evaluation is a third as long.  So the LAST patches of a launch
# are split (source_split = -k: one launch, its last resident set made of short items).  100 000-point sphere, the first K
# of the 256 patches, ms per launch plain / last 3 split (tools/gpu_xch_ab.py, profiles/r04_xch_ab.txt): K = 4 0.119 / 0.097,
# 16 0.319 / 0.281, 32 (a rank's share of 8) 0.559 / 0.538, 64 1.053 / 1.042, 128 2.047 / 2.040, 256 4.045 / 4.038 - it
# pays at every size, by less and less; above TAIL_BELOW_PAIRS the launch stays the plain one (the gain is inside the
# box-to-box noise there and the bench's kernel keeps its name).  Round 3's form of the tail (run terms in LDS) was worth
# -1 % at 32 patches and cost +3 % at 256.  Results do not depend on the choice (bit-identical slabs and partials).
TAIL_BELOW_PAIRS = 8e9
TAIL_PATCHES = 3
# Round 5: the tail is sized by its SOURCES, not by a patch count and not only for launches whose every patch has 129..512 points
# (round 4's rule, which never fired on the reference's own grid partitions: patches of 100..677 points): the last k patches of
# the launch with at least TAIL_SOURCES points between them - 3 of the bench's ~390-point patches, 2..8 of a grid partition's -,
# at most TAIL_MAX_PATCHES.  A tail patch of <= 128 points is one run (its items are short as they are), one of more than 512
# points stays with one wavefront per tile (an eight-wavefront item for those was built and measured: slower, see
# profiles/r05_xch_eight_wavefronts.patch).  One rank's share of eight (tools/gpu_rank_share.py, profiles/r05_rank_share_partitions.txt):
# bench partition 0.930 (plain launch 0.903), the reference's grid partition of the same sphere 0.910 (0.876), boxunion's 369
# patches of config 3 0.925 (0.839).
TAIL_SOURCES = 1000
TAIL_MAX_PATCHES = 8


def _pick_source_split(sizes_block: np.ndarray, n_targets: int) -> int:
    """source_split for a launch over patches of these sizes: 1, or -k (the last k patches split) when the launch is below
    TAIL_BELOW_PAIRS: k = the fewest trailing patches that hold TAIL_SOURCES points, at most TAIL_MAX_PATCHES - and none of them
    larger than 512 points: the tail ends in front of the last such patch (it would stay one wavefront per tile inside a
    four-wavefront item; with those counted in, rank 3 of 8 on the reference's grid partition fell from 0.90 to 0.84 of ideal,
    profiles/r05_tail_sweep.txt).  No tail when none of its patches has 129..512 points (nothing would be split)."""
    n = len(sizes_block)
    if n == 0 or float(sizes_block.sum()) * float(n_targets) >= TAIL_BELOW_PAIRS:
        return 1
    rev = np.asarray(sizes_block, dtype=np.int64)[::-1]
    big = np.flatnonzero(rev > 512)
    run = int(big[0]) if big.size else n                  # trailing patches of <= 512 points
    run = min(run, TAIL_MAX_PATCHES)
    if run == 0:
        return 1
    csum = np.cumsum(rev[:run])
    k = min(int(np.searchsorted(csum, TAIL_SOURCES, side="left")) + 1, run)
    tail = rev[:k]
    if not bool(np.any(tail > 128)):
        return 1
    return -k


# With the longest patches first (_launch_plan) a launch ends on its SHORTEST patches, and a split tail only pays when even those
# are long: on the reference's grid partition and on boxunion (last patches of 100..230 points) every tail costs 1-3 % of a rank's
# share of eight, with last patches of ~300 points (and on the bench's 343..439-point patches) it gains 1-3 % (profiles/r05_tail_sweep.txt).
TAIL_MIN_LAST = 256
_plan_cache = {}


def _launch_plan(sizes_block, n_targets: int, dev, tail: bool = True):
    """(order, source_split) of one pair-kernel launch over patches of these sizes (in patch order).  order: device int32
    permutation, longest patch first (stable; None when the patches already come that way) - workgroups are dispatched in launch
    order, so the shortest items drain the chip at the end (longest-processing-time-first): one rank's share of eight 0.91-0.95 of
    ideal on the reference's grid partition where patch order gave 0.84-0.92 with the best split tail (profiles/r05_tail_sweep.txt).
    source_split: the split tail for the last rows of THAT order, only when the shortest patch has more than TAIL_MIN_LAST points."""
    sizes_block = np.ascontiguousarray(sizes_block, dtype=np.int64)
    key = (sizes_block.tobytes(), int(n_targets), str(dev), bool(tail))
    hit = _plan_cache.get(key)
    if hit is not None:
        return hit
    perm = np.argsort(-sizes_block, kind="stable")
    in_order = sizes_block[perm]
    split = 1
    if tail and len(in_order) and int(in_order[-1]) > TAIL_MIN_LAST:
        split = _pick_source_split(in_order, n_targets)
    order = None if bool(np.all(perm == np.arange(len(perm)))) else util.to_device(perm.astype(np.int32), dev)
    if len(_plan_cache) > 256:
        _plan_cache.clear()
    _plan_cache[key] = (order, split)
    return order, split


def _slabs_and_rows(swork, off, point_patch, b0: int, b1: int, eps: float, boxes, tiles: "_TileTables", sizes=None):
    """One evaluation of patches b0..b1 on the patch-sorted cloud: (dE[b1-b0, N, 3], W rows [b1-b0, P] fp64).  When the
    tiles allow it the interaction rows come out of the pair kernel's epilogue (+ a tiny gather kernel), otherwise from
    the K3 pass over the slabs."""
    P = off.shape[0] - 1
    f64 = swork.dtype == torch.float64
    order, split = (None, 1) if sizes is None else _launch_plan(np.asarray(sizes)[b0:b1], swork.shape[0], swork.device,
                                                               tail=not (tiles is None or boxes is None or f64))
    if tiles is not None and tiles.fused and ((boxes is not None and eps >= 1e-30) or (f64 and eps > 0)):
        lib = _lib.require_device()
        K, N = b1 - b0, swork.shape[0]
        w_part = torch.empty((K, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=swork.device)
        dE = _patch_slabs(swork, off, None, point_patch, b0, b1, eps, boxes, tiles.boxes, w_part, split, order)
        W = torch.empty((K, P), dtype=torch.float64, device=swork.device)            # (tile geometry and sums: the same in both precisions)
        with _on_device(swork.device):
            _lib.check(lib.dnp_interactions_from_tiles(_lib.ptr(w_part), tiles.slots, K, N, _lib.ptr(point_patch), _lib.ptr(off), P,
                                                       _lib.ptr(W), _lib.current_stream()))
        return dE, W
    dE = _patch_slabs(swork, off, None, point_patch, b0, b1, eps, boxes, None if tiles is None else tiles.boxes, None, split, order)
    return dE, _interaction_rows(dE, swork, off, None)


def _interaction_rows(dE, work, off, idx) -> torch.Tensor:
    lib = _lib.require_device()
    K, N = dE.shape[0], dE.shape[1]
    P = off.shape[0] - 1
    W = torch.empty((K, P), dtype=torch.float64, device=work.device)
    fn = lib.dnp_interactions_f64 if dE.dtype == torch.float64 else lib.dnp_interactions_f32
    with _on_device(work.device):
        rc = fn(_lib.ptr(dE), K, N, _lib.ptr(work), work.stride(0), _lib.ptr(off), _lib.ptr(idx), P, _lib.ptr(W),
                _lib.current_stream())
    _lib.check(rc)
    return W


def _combine(dE, coef: torch.Tensor, slab: torch.Tensor, E: torch.Tensor, accumulate: bool):
    """E (+)= sum_i coef_i dE[slab_i] as a sequential fp32 chain in the order given (dnp_combine_fields_f32)."""
    lib = _lib.require_device()
    K, N = dE.shape[0], dE.shape[1]
    with _on_device(E.device):
        rc = lib.dnp_combine_fields_f32(_lib.ptr(dE), K, N, _lib.ptr(coef), _lib.ptr(slab), coef.shape[0],
                                        _lib.ptr(E), int(accumulate), _lib.current_stream())
    _lib.check(rc)


def _combine_signed(dE, sigma: torch.Tensor, p_lo: int, E64: torch.Tensor, accumulate: bool):
    """E64 (+)= sum_k sigma[p_lo + k] dE[k], accumulated in fp64 (dnp_combine_signed_f32 / _f64 by the slabs' precision)."""
    lib = _lib.require_device()
    K, N = dE.shape[0], dE.shape[1]
    fn = lib.dnp_combine_signed_f64 if dE.dtype == torch.float64 else lib.dnp_combine_signed_f32
    with _on_device(E64.device):
        rc = fn(_lib.ptr(dE), K, N, _lib.ptr(sigma), sigma.shape[0], int(p_lo), _lib.ptr(E64), int(accumulate),
                _lib.current_stream())
    _lib.check(rc)


def greedy_order_from_interactions(W: np.ndarray, start: int):
    """The greedy loop of field_utils.py:314-324 / :242-254 on the P x P interaction matrix (host form; the
    drivers run the same loop on the device, dnp_patch_greedy):
    I_j = sum_{k visited} sigma_k W[k, j]; pick argmax |I_j| over the remaining patches (first
    maximum in patch order, as torch.argmax over the `remaining` list), flip when I_j < 0.
    Returns (order[P], sigma[P], chosen_interaction[P-1])."""
    P = W.shape[0]
    sigma = np.ones(P)
    visited = np.zeros(P, dtype=bool)
    order = [start]
    visited[start] = True
    inter = W[start].astype(np.float64).copy()
    chosen = []
    for _ in range(P - 1):
        mag = np.where(visited, -np.inf, np.abs(inter))
        j = int(np.argmax(mag))
        chosen.append(inter[j])
        if inter[j] < 0:
            sigma[j] = -1.0
        visited[j] = True
        order.append(j)
        inter += sigma[j] * W[j]
    return np.array(order), sigma, np.array(chosen)


# Capacity cliffs of the device-side drivers (both have a slower, tested way around them):
#   * patch greedy loop: one wavefront up to 2048 patches, one workgroup up to dnp_patch_greedy_max_patches() = 16 384
#     (2.5-2.9 us per step); beyond that the loop runs on the host over a copy of W (greedy_order_from_interactions:
#     ~20 us per step in numpy plus one P x P device -> host copy).  PATCH_GREEDY_MAX lowers the limit (tests force the
#     host loop with 0).
#   * per-point greedy (K4): the persistent kernels hold N < 2^20 points (the index field of their 8-byte granules) and
#     at most 512 x 20 (fp32) / 512 x 8 (fp64) points per CU; beyond either, _points_stepwise launches one field kernel
#     and three small torch kernels per step (~140 us per step).  POINT_GREEDY_MAX_PER_GROUP lowers the limit.
PATCH_GREEDY_MAX = None


def _greedy_on_device(W: torch.Tensor, start_t: torch.Tensor):
    """(order[P] int64, sigma[P] fp64, chosen[P-1] fp64) device tensors from the full W[P,P] (fp64, device)."""
    lib = _lib.require_device()
    P = W.shape[0]
    dev = W.device
    if P > (lib.dnp_patch_greedy_max_patches() if PATCH_GREEDY_MAX is None else PATCH_GREEDY_MAX):
        order, sigma, chosen = greedy_order_from_interactions(W.cpu().numpy(), int(start_t.item()))
        return (torch.from_numpy(order).to(dev), torch.from_numpy(sigma).to(dev), torch.from_numpy(chosen).to(dev))
    order = torch.empty(P, dtype=torch.int64, device=dev)
    sigma = torch.empty(P, dtype=torch.float64, device=dev)
    chosen = torch.empty(max(P - 1, 0), dtype=torch.float64, device=dev)
    Wc = W if W.is_contiguous() else W.contiguous()
    with _on_device(dev):
        rc = lib.dnp_patch_greedy(_lib.ptr(Wc), P, _lib.ptr(start_t), _lib.ptr(order), _lib.ptr(sigma),
                                  _lib.ptr(chosen), _lib.current_stream())
    _lib.check(rc)
    return order, sigma, chosen


class _Batched:
    __slots__ = ("order", "sigma", "chosen", "Es", "perm", "swork", "sorted_patch")

    def __init__(self, order, sigma, chosen, Es, perm, swork, sorted_patch):
        self.order, self.sigma, self.chosen, self.Es = order, sigma, chosen, Es
        self.perm, self.swork, self.sorted_patch = perm, swork, sorted_patch

    @property
    def point_patch(self) -> torch.Tensor:
        """patch id of every point in the caller's row order (-1 = in no patch)."""
        out = torch.empty_like(self.sorted_patch)
        out[self.perm] = self.sorted_patch
        return out

    def field(self) -> Optional[torch.Tensor]:
        """E64 in the caller's row order."""
        if self.Es is None:
            return None
        E = torch.empty_like(self.Es)
        E[self.perm] = self.Es
        return E


def _finish_batched(pts: torch.Tensor, st: "_Batched", diffuse: bool, listed_patches, w) -> None:
    """The tail of a batched patch driver in one launch (dnp_patch_finish_f32): patch flips, diffuse sign pass on
    the listed patches, weight un-scaling, and the store into the caller's tensor (any float dtype/device)."""
    lib = _lib.require_device()
    dev = st.swork.device
    N = st.swork.shape[0]
    direct = pts.is_cuda and pts.device == dev and pts.dtype in (torch.float32, torch.float64) and pts.stride(1) == 1
    out = pts if direct else torch.empty((N, 6), dtype=st.swork.dtype, device=dev)
    w_sorted = None if w is None else w[st.perm].contiguous()
    fn = lib.dnp_patch_finish_f64 if st.swork.dtype == torch.float64 else lib.dnp_patch_finish_f32
    with _on_device(dev):
        rc = fn(_lib.ptr(st.swork), st.swork.stride(0), N, _lib.ptr(st.sorted_patch), _lib.ptr(st.sigma),
                _lib.ptr(st.Es if diffuse else None), _lib.ptr(listed_patches), _lib.ptr(w_sorted), _lib.ptr(st.perm),
                _lib.ptr(out), out.stride(0), int(out.dtype == torch.float64), _lib.current_stream())
    _lib.check(rc)
    if not direct:
        _store_normals(pts, out[:, 3:])


def _listed_patches(patches, all_patches, dev) -> Optional[torch.Tensor]:
    """uint8[P] flags of the filtered patches [(i, idx)] when every idx IS all_patches[i] (what the callers pass:
    inference_utils.fix_n_filter returns the patch objects it was given); None when the lists are something else."""
    if any(not (0 <= i < len(all_patches)) or patch is not all_patches[i] for i, patch in patches):
        return None
    flags = np.zeros(len(all_patches), dtype=np.uint8)
    if len(patches):
        flags[[i for i, _ in patches]] = 1
    return util.to_device(flags, dev)


class _BatchedWork:
    """What _batched_begin leaves for _batched_end: the sorted cloud, the tables, this rank's W rows and the slab blocks
    kept for the diffuse combine."""
    __slots__ = ("swork", "perm", "point_patch", "off", "sizes", "boxes", "tiles", "bounds", "p_lo", "p_hi", "batch", "kept",
                 "W_local", "diffuse", "want_E", "eps")


def _batched_begin(work: torch.Tensor, patches, diffuse: bool, eps: float = 1e-5, want_E: bool = True,
                   rank: int = 0, world: int = 1, subset: bool = False) -> "_BatchedWork":
    """First half of the batched drivers on a device cloud `work[N,6]` (normals already weight-scaled): the patch-sorted
    layout, the box tables, this rank's slabs and its rows of W - everything that does not need the other ranks.  No host
    synchronisation; the second half is _batched_end.  subset = True: the propagation runs on the LISTED points only (the
    representatives driver: targets are representatives) - the sorted cloud has one row per listed point, perm maps it to the
    row of `work`, and points in no patch take no part (the caller guarantees disjoint lists)."""
    dev = work.device
    N = work.shape[0]
    off, idx, sizes = util.patch_csr(patches, dev)
    P = len(sizes)
    # data layout for the kernels: the cloud sorted by patch (points in no patch last), so that a patch
    # is a contiguous row range - sources stream linearly and K3 reads its slab rows coalesced
    covered = int(sizes.sum())
    if subset:
        N = covered
    if covered == N and work.is_cuda and work.dtype in (torch.float32, torch.float64):
        # every point is in a patch (the callers' case): one launch builds the sorted cloud and its patch ids
        perm = idx
        swork = torch.empty((N, 6), dtype=work.dtype, device=dev)
        point_patch = torch.empty(N, dtype=torch.int64, device=dev)
        lib = _lib.require_device()
        layout = lib.dnp_patch_layout_f64 if work.dtype == torch.float64 else lib.dnp_patch_layout_f32
        with _on_device(dev):
            _lib.check(layout(_lib.ptr(work), work.stride(0), _lib.ptr(off), _lib.ptr(idx), P, _lib.ptr(swork),
                              _lib.ptr(point_patch), _lib.current_stream()))
    else:
        point_patch = torch.repeat_interleave(torch.arange(P, device=dev), util.to_device(sizes, dev),
                                              output_size=covered)
        if covered == N:
            perm = idx
        else:
            seen = torch.zeros(N, dtype=torch.bool, device=dev)
            seen[idx] = True
            loose = torch.nonzero(~seen).flatten()
            perm = torch.cat([idx, loose])
            point_patch = torch.cat([point_patch, torch.full((loose.shape[0],), -1, dtype=torch.int64, device=dev)])
        swork = work[perm].contiguous()

    # contiguous blocks of patches per rank, balanced by pair count |patch| * N
    bounds = _balanced_blocks(sizes, world)
    p_lo, p_hi = int(bounds[rank]), int(bounds[rank + 1])

    # Slab memory: every patch's field on all N points is 12 N bytes.  Within SLAB_BUDGET_BYTES all slabs of this rank
    # are evaluated by one launch and kept for the diffuse combine.  Beyond it the budget is raised to 80 % of what
    # the device has free (288 GB of HBM on an MI355X), the patches go through in blocks of at most SLAB_BLOCK_BYTES,
    # as many blocks as fit are kept, and only the others are evaluated a second time for the combine.
    boxes = _patch_boxes(swork, off, None)               # for the far-field test of the pair kernel
    tiles = _TileTables(swork, sizes)                    # target-tile boxes; can W come out of the kernel's epilogue?
    per_slab = N * 3 * swork.element_size()
    n_local = max(p_hi - p_lo, 1)
    budget = SLAB_BUDGET_BYTES
    if n_local * per_slab > SLAB_FREE_CHECK_BYTES:
        # anything sizeable is checked against what the device really has free (a GPU shared by several ranks, a
        # smaller part): within min(SLAB_BUDGET_BYTES, 80 % of free) one pass; beyond SLAB_BUDGET_BYTES the budget
        # becomes 80 % of free
        free = int(0.8 * _free_device_bytes(dev))
        budget = min(budget, free) if n_local * per_slab <= budget else free
    if n_local * per_slab <= budget:
        batch = n_local
    else:
        batch = max(1, min(n_local, min(SLAB_BLOCK_BYTES, budget // 2) // max(per_slab, 1)))
    W_rows, kept, kept_bytes = [], {}, 0
    for b0 in range(p_lo, p_hi, batch):
        b1 = min(b0 + batch, p_hi)
        dE, rows = _slabs_and_rows(swork, off, point_patch, b0, b1, eps, boxes, tiles, sizes)
        W_rows.append(rows)
        # keep this block if it and one more working block still fit
        if want_E and diffuse and kept_bytes + (b1 - b0) * per_slab + (batch * per_slab if b1 < p_hi else 0) <= budget:
            kept[b0] = dE
            kept_bytes += (b1 - b0) * per_slab
        del dE
    bw = _BatchedWork()
    bw.swork, bw.perm, bw.point_patch, bw.off, bw.sizes, bw.boxes, bw.tiles = swork, perm, point_patch, off, sizes, boxes, tiles
    bw.bounds, bw.p_lo, bw.p_hi, bw.batch, bw.kept = bounds, p_lo, p_hi, batch, kept
    bw.diffuse, bw.want_E, bw.eps = diffuse, want_E, eps
    if len(W_rows) == 1:
        bw.W_local = W_rows[0]
    else:
        bw.W_local = torch.cat(W_rows, dim=0) if W_rows else torch.zeros((0, P), dtype=torch.float64, device=dev)
    return bw


def _batched_end(bw: "_BatchedWork", W_full: torch.Tensor, start_t: torch.Tensor) -> "_Batched":
    """Second half: the greedy loop on the full W and, for the diffuse form, this rank's fp64 partial field."""
    dev = bw.swork.device
    N = bw.swork.shape[0]
    order, sigma, chosen = _greedy_on_device(W_full, start_t)
    E64 = None
    if bw.want_E and bw.diffuse:
        # E = sum_k sigma_k dE_k (field_utils.py:330-331).  The reference adds the fp32 slabs one by one in
        # visit order; here the signed slabs are summed in fp64 and rounded to fp32 once, which is closer to the
        # exact sum and does not depend on the visit order or on how the patches are split over GPUs.
        Es = torch.empty((N, 3), dtype=torch.float64, device=dev)
        first = True
        for b0 in range(bw.p_lo, bw.p_hi, bw.batch):
            b1 = min(b0 + bw.batch, bw.p_hi)
            dE = bw.kept.pop(b0, None)
            if dE is None:
                dE = _patch_slabs(bw.swork, bw.off, None, bw.point_patch, b0, b1, bw.eps, bw.boxes,
                                  None if bw.tiles is None else bw.tiles.boxes)
            _combine_signed(dE, sigma, b0, Es, not first)
            first = False
            del dE
        if first:
            Es.zero_()
        E64 = Es
    return _Batched(order, sigma, chosen, E64, bw.perm, bw.swork, bw.point_patch)


def _batched_patch_propagation(work: torch.Tensor, patches, start_t: torch.Tensor, diffuse: bool,
                               eps: float = 1e-5, want_E: bool = True, shard=None, subset: bool = False):
    """Core of the batched drivers on a device cloud `work[N,6]` (normals already weight-scaled); everything
    stays on the device - no host synchronisation between the launches.

    Returns a _Batched: device tensors order / sigma / chosen; Es[N,3] = this rank's part of the accumulated
    field of the diffuse form (fp64 sum of the sigma-signed fp32 slabs) in PATCH-SORTED row order (None unless
    want_E and diffuse); perm (sorted row -> caller's row), the sorted working cloud and patch ids, and

    `shard` = (rank, world, gather_fn): patches are split over ranks in contiguous size-balanced blocks, each
    rank computes its slabs and W rows, gather_fn(rows, bounds) returns the full W on every rank."""
    rank, world, gather = (0, 1, None) if shard is None else shard
    bw = _batched_begin(work, patches, diffuse, eps, want_E, rank, world, subset)
    W_full = bw.W_local if gather is None else gather(bw.W_local, bw.bounds)
    return _batched_end(bw, W_full, start_t)


def _free_device_bytes(dev) -> int:
    """Bytes a new tensor could take: what the driver reports free plus what torch's allocator holds unused."""
    return int(torch.cuda.mem_get_info(dev)[0]) + int(torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))


def _balanced_blocks(sizes: np.ndarray, world: int) -> np.ndarray:
    """Cut 0..P into `world` contiguous blocks of patches; returns world+1 bounds.  Equal counts when that
    leaves the pair work within 5 % of balanced (uniform patch sizes: the all-gather then needs no padding),
    otherwise cuts at equal cumulative size."""
    P = len(sizes)
    if world <= 1:
        return np.array([0, P])
    csum = np.concatenate([[0], np.cumsum(sizes)])
    if P % world == 0:
        even = np.arange(world + 1) * (P // world)
        work = np.diff(csum[even])
        if work.max() <= 1.05 * csum[-1] / world:
            return even.astype(np.int64)
    targets = csum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(csum, targets, side="left")
    return np.concatenate([[0], np.clip(cuts, 0, P), [P]]).astype(np.int64)


def _sequential_patch_propagation(work, patches: List[torch.Tensor], start: int, diffuse: bool, eps=1e-5):
    """Step-by-step form: one field launch per greedy step, exactly the loop of
    field_utils.py:308-335 with the masks expressed as index lists.  Works for overlapping
    patches too.  Returns (order, sigma, chosen, E)."""
    dev = work.device
    N, P = work.shape[0], len(patches)
    pidx = [p.to(device=dev, dtype=torch.int64) for p in patches]
    E = torch.zeros((N, 3), dtype=work.dtype, device=dev)
    oriented = torch.zeros(N, dtype=torch.bool, device=dev)
    all_rows = torch.arange(N, device=dev)

    def add_field(k):
        src_idx = pidx[k]
        if diffuse:
            m = torch.ones(N, dtype=torch.bool, device=dev)
            m[src_idx] = False
        else:
            m = ~oriented
        tgt_idx = all_rows[m]
        if tgt_idx.numel():
            _pairs_into("field", work, src_idx, work, tgt_idx, eps, 15000, E, out_scatter=True, accumulate=True)

    # per-step interaction of every patch by ONE segmented sum over the concatenated index lists (works for
    # overlapping lists too), instead of one gather per remaining patch
    sizes = torch.tensor([int(p.shape[0]) for p in pidx], device=dev)
    cat_idx = torch.cat(pidx) if P else torch.zeros(0, dtype=torch.int64, device=dev)
    pending = torch.ones(P, dtype=torch.bool, device=dev)
    pending[start] = False
    oriented[pidx[start]] = True
    # E[~mask] = field_grad(pts[start], pts[~start])
    tgt0 = all_rows[~oriented]
    if tgt0.numel():
        _pairs_into("field", work, pidx[start], work, tgt0, eps, 15000, E, out_scatter=True, accumulate=True)
    order, sigma, chosen = [start], np.ones(P), []
    neg_inf = torch.full((P,), float("-inf"), dtype=torch.float64, device=dev)
    for _ in range(P - 1):
        dots = (E * work[:, 3:]).sum(dim=-1)
        inter = torch.segment_reduce(dots[cat_idx].double(), "sum", lengths=sizes)   # deterministic, in patch order
        # first maximum in patch order among the pending ones == argmax over the reference's `remaining` list
        k = int(torch.where(pending, inter.abs(), neg_inf).argmax().item())
        v = float(inter[k])
        chosen.append(v)
        if v < 0:
            work[pidx[k], 3:] *= -1
            sigma[k] = -1.0
        pending[k] = False
        oriented[pidx[k]] = True
        order.append(k)
        add_field(k)
    return np.array(order), sigma, np.array(chosen), E


def _store_normals(pts: torch.Tensor, normals: torch.Tensor) -> None:
    """pts[:, 3:] = normals for the caller's tensor on any device / float dtype.  For a HOST tensor the rows go over
    PCIe as one contiguous copy and into the strided column view through numpy: torch's CPU copy kernel splits these
    300 000 elements over every core it sees, and in a container with a CPU quota (16 of a box's 192+) that is 22 ms of
    oversubscribed threads for a 1.2 MB copy (tools/gpu_host_boundary.py) - 5x the whole propagation."""
    if pts.is_cuda or pts.dtype not in (torch.float32, torch.float64) or pts.requires_grad:
        pts[:, 3:] = normals.to(device=pts.device, dtype=pts.dtype)
        return
    pts.numpy()[:, 3:] = normals.contiguous().cpu().numpy()


def _prepare_work(pts: torch.Tensor, weights):
    """Device working copy of pts (normals scaled by clamp(weights, 0.1, 1)) in the precision the propagation runs in: float64
    for a float64 cloud - the reference's drivers compute in pts.dtype (field_utils.py:286-348, :207-282; rounds 1-4 narrowed
    such a cloud to fp32 silently) -, float32 for everything else."""
    dev = pts.device if pts.is_cuda else _compute_device()
    wd = torch.float64 if pts.dtype == torch.float64 else torch.float32
    work = pts.detach().to(device=dev, dtype=wd).contiguous()
    if work.data_ptr() == pts.data_ptr():
        work = work.clone()
    w = None
    if weights is not None:
        w = weights.detach().to(device=dev, dtype=wd).clamp(0.1, 1)
        work[:, 3:] = work[:, 3:] * w[:, None]
    return work, w


def _finish_patch_driver(pts, work, w):
    if w is not None:
        work[:, 3:] = work[:, 3:] / w[:, None]
    _store_normals(pts, work[:, 3:])


def strongest_field_propagation(pts, patches, all_patches, diffuse=False, weights=None, start_patch=None):
    """Greedy patch orientation (field_utils.py:286-348).  `pts[N,6]` normals are updated in place.

    patches:      list of (i, index tensor) - the filtered patches that receive the per-point
                  diffuse sign pass
    all_patches:  list of index tensors - every patch takes part in the greedy ordering
    start_patch:  (extension, default None = the reference's rule: the flattest patch) pin the starting patch.

    The visit order / flips are available afterwards from last_trace("patches")."""
    with torch.no_grad():
        if len(all_patches) == 0:
            return
        work, w = _prepare_work(pts, weights)
        dev = work.device
        start_t = _start_tensor(work, all_patches, start_patch)
        mode = PATCH_MODE
        if mode == "auto":
            known = isinstance(all_patches, util.PatchList) and all_patches.disjoint
            mode = "batched" if known or _disjoint(_csr(all_patches, dev)[1], work.shape[0]) else "sequential"
        if mode == "batched":
            listed = _listed_patches(patches, all_patches, dev) if diffuse else None   # host->device copy: before the long kernels
            st = _batched_patch_propagation(work, all_patches, start_t, diffuse)
            if not diffuse or listed is not None:
                _finish_batched(pts, st, diffuse, listed, w)
            else:                                       # diffuse lists that are not the patches themselves
                flip = torch.where(st.point_patch >= 0, st.sigma[st.point_patch.clamp(min=0)], 1.0).to(work.dtype)
                work[:, 3:] = work[:, 3:] * flip[:, None]
                _diffuse_sign_pass(work, st.field().to(work.dtype), [patch for _, patch in patches])
                _finish_patch_driver(pts, work, w)
            _set_trace("patches", order=st.order, sigma=st.sigma, chosen=st.chosen, start=start_t)
            return
        order, sigma, chosen, E = _sequential_patch_propagation(work, list(all_patches), int(start_t.item()), diffuse)
        if diffuse:
            _diffuse_sign_pass(work, E, [patch for _, patch in patches])
        _finish_patch_driver(pts, work, w)
        _set_trace("patches", order=order, sigma=sigma, chosen=chosen, start=start_t)


def strongest_field_propagation_reps(input_pc, reps, diffuse=False, weights=None, start_patch=None):
    """Greedy orientation on <=500 representatives per patch (field_utils.py:207-282): `reps` is a
    list of (rep_idx, rest_idx); a flipped patch flips its rest points too; afterwards every
    non-representative point takes the sign of the field of all representatives (:273-276).
    Trace: last_trace("reps")."""
    input_pc = input_pc.detach()
    with torch.no_grad():
        if len(reps) == 0:
            return
        work, w = _prepare_work(input_pc, weights)
        dev = work.device
        N = work.shape[0]
        known_disjoint = False
        if isinstance(reps, util.RepLists):
            rep_csr, rest_csr = util.patch_csr(reps.reps, dev), util.patch_csr(reps.rests, dev)
            known_disjoint = reps.reps.disjoint
        else:
            rep_csr = util.patch_csr([r for r, _ in reps], dev)
            rest_csr = util.patch_csr([r for _, r in reps], dev)
        _, all_reps, rep_sizes = rep_csr
        rep_lists = util.PatchList(all_reps, rep_sizes)
        start_t = _start_tensor(work, rep_lists, start_patch)
        mode = PATCH_MODE
        if mode == "auto":
            mode = "batched" if known_disjoint or _disjoint(all_reps, N) else "sequential"
        n_rest = int(rest_csr[2].sum())
        # representatives and rests partition the cloud (what the callers pass): every point is listed exactly once
        partition = known_disjoint and isinstance(reps, util.RepLists) and reps.rests.disjoint \
            and int(rep_sizes.sum()) + n_rest == N
        if mode == "batched" and partition and work.is_cuda:
            # The callers' case, everything in the library's kernels (round 5; rounds 2-4 ran ~40 small torch launches around
            # the pair kernel here - profiles/r05_config3_kernels.txt).  The loop's targets are representatives only: the
            # sorted cloud of the propagation is built straight from `work` through the representatives' index list
            # (subset), the fused tail stores their oriented normals back into `work`, the field of ALL representatives
            # is evaluated at the rest points, and one launch gives every rest point its patch's flip and the sign of that field.
            lib = _lib.require_device()
            f64 = work.dtype == torch.float64
            st = _batched_patch_propagation(work, rep_lists, start_t, diffuse, subset=True)
            order, sigma, chosen = st.order, st.sigma, st.chosen
            with _on_device(dev):
                rc = (lib.dnp_patch_finish_f64 if f64 else lib.dnp_patch_finish_f32)(
                    _lib.ptr(st.swork), st.swork.stride(0), st.swork.shape[0], _lib.ptr(st.sorted_patch), _lib.ptr(st.sigma),
                    _lib.ptr(st.Es if diffuse else None), None, None, _lib.ptr(st.perm), _lib.ptr(work), work.stride(0),
                    int(f64), _lib.current_stream())
            _lib.check(rc)
            if n_rest:
                rest = rest_csr[1]                                # targets are independent rows: their order is free
                # sources as a compact copy in PATCH order: contiguous rows go through the scalar-unit kernel and chunks of one or
                # two patches have tight boxes, so the far-field chain fires (the reference sums them in point order,
                # field_grad(pts[oriented_pts_mask], ...): the same sum up to fp32 rounding - the kernels add fp64 chunk sums - and
                # up to which 15 000-source leaf an Inf / NaN pair would blank; 392 -> 359 us, profiles/r05_rest_field_ab.txt)
                src = work[all_reps].contiguous()
                E2 = torch.empty((n_rest, 3), dtype=work.dtype, device=dev)
                _pairs_into("field", src, None, work, rest, 1e-5, 15000, E2)
                with _on_device(dev):
                    rc = (lib.dnp_rest_finish_f64 if f64 else lib.dnp_rest_finish_f32)(
                        _lib.ptr(work), work.stride(0), _lib.ptr(rest_csr[0]), _lib.ptr(rest), len(rest_csr[2]),
                        _lib.ptr(sigma), _lib.ptr(E2), _lib.current_stream())
                _lib.check(rc)
            _finish_patch_driver(input_pc, work, w)
            _set_trace("reps", order=order, sigma=sigma, chosen=chosen, start=start_t)
            return
        # ---- general lists (overlapping, not a partition, or the step-by-step mode): the loop runs on the compact sub-cloud
        # of the representatives (patch k = the contiguous row range of its representatives), the rest in torch
        rep_pid, rest_pid = _listing_ids(rep_csr), _listing_ids(rest_csr)
        sub = work[all_reps].contiguous()
        sub_patches = util.PatchList(torch.arange(all_reps.shape[0], device=dev), rep_sizes)
        if mode == "batched":
            st = _batched_patch_propagation(sub, sub_patches, start_t, diffuse)
            order, sigma, chosen = st.order, st.sigma, st.chosen
            neg = sigma < 0
        else:
            order, sigma, chosen, E_sub = _sequential_patch_propagation(sub, list(sub_patches), int(start_t.item()),
                                                                        diffuse)
            neg = torch.from_numpy(sigma < 0).to(dev)
        E = torch.zeros((N, 3), dtype=work.dtype, device=dev)
        if mode == "batched":
            if st.Es is not None:
                E[all_reps] = st.field().to(work.dtype)
        else:
            E[all_reps] = E_sub
        _flip_by_listing(work, neg, all_reps, rep_pid)
        _flip_by_listing(work, neg, rest_csr[1], rest_pid)
        if diffuse:
            _diffuse_sign_pass(work, E, rep_lists)
        # every non-representative point: sign of the field of all representatives
        # (field_grad(pts[oriented_pts_mask], pts[~oriented_pts_mask]): both sides in point order)
        is_rep = torch.zeros(N, dtype=torch.bool, device=dev)
        is_rep[all_reps] = True
        rest = torch.nonzero(~is_rep).flatten()
        rest = rest if rest.numel() else None
        src_rows = torch.nonzero(is_rep).flatten()
        if rest is not None:
            src = work[src_rows].contiguous()
            E2 = torch.empty((rest.shape[0], 3), dtype=work.dtype, device=dev)
            _pairs_into("field", src, None, work, rest, 1e-5, 15000, E2)
            s = ((E2 * work[rest, 3:]).sum(dim=-1) > 0).to(work.dtype) * 2 - 1
            work[rest, 3:] = work[rest, 3:] * s[:, None]
        _finish_patch_driver(input_pc, work, w)
        _set_trace("reps", order=order, sigma=sigma, chosen=chosen, start=start_t)
