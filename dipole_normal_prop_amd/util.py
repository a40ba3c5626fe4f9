"""Host helpers the dipole hot path touches - own-code counterparts of the six `util.py`
helpers of the reference that sit on the path.  Everything here is
O(N) host/torch plumbing; none of it is the product's arithmetic.

Reference behaviour each function reproduces (file:line under the reference tree):
    gen_grid            util.py:26-36      10^3 lattice {-1,-0.8,...,0.8}^3
    orient_center       util.py:39-44      flip normals pointing towards the patch centroid
    export_pc           util.py:46-51      '.xyz' writer
    xyz2tensor          util.py:53-69      '.xyz' parser (3 or 6 columns, 'nan' lines dropped)
    Transform           util.py:577-609    centre + scale to the unit box, and back
    divide_pc           util.py:110-150 + util.py:448-492   voxel partition, then merge of small patches
    pca_eigen_values    util.py:495-500    smallest covariance eigen-pair of a patch
    timer_factory       util.py:612-649    wall-clock stage timer
"""
import ctypes
import time
from typing import List

import numpy as np
import torch


def gen_grid(n: int = 10) -> torch.Tensor:
    """The n^3 probe lattice of measure_mean_potential: coordinates (i/n - 0.5) * 2, x slowest."""
    axis = torch.arange(n, dtype=torch.float32)
    gx, gy, gz = torch.meshgrid(axis, axis, axis, indexing="ij")
    pts = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=1)
    pts = pts / n
    pts -= 0.5
    pts *= 2
    return pts


def orient_center(pred: torch.Tensor) -> torch.Tensor:
    """Flip (in place) the normals of `pred[N,6]` that point towards the centroid."""
    rel = pred[:, :3] - pred[:, :3].mean(dim=0)
    inward = (rel * pred[:, 3:]).sum(dim=-1) < 0
    pred[inward, 3:] *= -1
    return pred


def export_pc(pc: torch.Tensor, dest) -> None:
    """Write a [C,N] tensor (the callers pass pc.transpose(0,1)) as N lines of C numbers.
    Format as the reference: str(float) joined by ' ', lines joined by '\n', no trailing newline.  float32
    clouds are formatted by the native library (dnp_xyz_format_f32: the same bytes, ~10x faster than 600 000
    str() calls); anything else takes the literal Python path."""
    rows = pc.transpose(0, 1).detach().cpu()
    text = _format_rows_native(rows) if rows.dtype == torch.float32 and rows.dim() == 2 else None
    if text is None:
        text = "\n".join(" ".join(str(v) for v in row) for row in rows.tolist()).encode()
    with open(dest, "wb") as fh:
        fh.write(text)


def _format_rows_native(rows: torch.Tensor):
    try:
        from . import _lib
        lib = _lib.load()
    except Exception:
        return None
    arr = np.ascontiguousarray(rows.numpy())
    cap = int(lib.dnp_xyz_format_bound(arr.shape[0], arr.shape[1]))
    buf = np.empty(cap, dtype=np.uint8)        # not create_string_buffer: that zero-fills 15 MB and .raw copies them again
    n = lib.dnp_xyz_format_f32(arr.ctypes.data, arr.shape[0], arr.shape[1], buf.ctypes.data, cap)
    return memoryview(buf)[:n] if n >= 0 else None       # written to the file as it is (bytes-like)


def _xyz_native(txt: str, append_normals: bool):
    """Regular text through the native parser (dnp_xyz_parse_f32); None when the text is not regular or the library
    is not available - the callers then fall back to the paths below."""
    try:
        from . import _lib
        lib = _lib.load()
    except Exception:
        return None
    raw = txt.encode()
    max_rows = len(raw) // 6 + 1               # a row is at least "1 2 3\n" (counting the newlines costs 4 ms per 7 MB; the
    out = np.empty((max_rows, 6), dtype=np.float32)   # pages of the over-sized buffer that are never written are never touched)
    ncol = ctypes.c_int32(0)
    n = lib.dnp_xyz_parse_f32(raw, len(raw), out.ctypes.data, max_rows, ctypes.byref(ncol))
    if n <= 0 or ncol.value not in (3, 6):
        return None
    arr = out.reshape(-1)[: n * ncol.value].reshape(n, ncol.value)
    if ncol.value == 3 and append_normals:
        arr = np.concatenate([arr, np.zeros((n, 3), dtype=np.float32)], axis=1)
    return torch.from_numpy(np.array(arr))      # a copy of the rows: the over-sized buffer goes


def _xyz_fast(txt: str, append_normals: bool):
    """Vectorised parse for the regular case - every non-empty line has the same number (3 or 6) of
    single-space separated numbers and there is no 'nan' anywhere.  Returns None when the text is not of
    that form; the caller then takes the line-by-line path, which defines the semantics (blank lines are
    skipped there as well: they split into one token)."""
    if "nan" in txt or "\t" in txt:
        return None
    rows = [ln for ln in (raw.strip() for raw in txt.split("\n")) if ln]
    if not rows:
        return None
    ncol = rows[0].count(" ") + 1
    if ncol not in (3, 6) or any(ln.count(" ") != ncol - 1 for ln in rows):
        return None
    try:
        vals = np.array(" ".join(rows).split(" "), dtype=np.float64)    # float(): correctly rounded, as float(x)
    except ValueError:
        return None
    if vals.size != len(rows) * ncol:
        return None
    arr = vals.reshape(len(rows), ncol).astype(np.float32)
    if ncol == 3 and append_normals:
        arr = np.concatenate([arr, np.zeros((len(rows), 3), dtype=np.float32)], axis=1)
    return torch.from_numpy(arr)


def xyz2tensor(txt: str, append_normals: bool = True) -> torch.Tensor:
    """Parse '.xyz' text: space separated, 3 or 6 columns per line; lines containing 'nan' are
    dropped; 3-column lines get zero normals appended when append_normals."""
    fast = _xyz_native(txt, append_normals)
    if fast is None:
        fast = _xyz_fast(txt, append_normals)
    if fast is not None:
        return fast
    rows = []
    for line in txt.split("\n"):
        line = line.strip()
        if "nan" in line:
            continue
        cols = line.split(" ")
        if len(cols) == 6:
            rows.append([float(c) for c in cols])
        elif len(cols) == 3:
            vals = [float(c) for c in cols]
            if append_normals:
                vals += [0.0, 0.0, 0.0]
            rows.append(vals)
    if not rows:
        raise RuntimeError("stack expects a non-empty TensorList")   # what torch.stack([]) raises
    width = len(rows[0])
    if any(len(r) != width for r in rows):
        raise RuntimeError("stack expects each tensor to be equal size")
    return torch.tensor(rows, dtype=torch.float32)


def npxyz2tensor(np_pc, append_normals: bool = True) -> torch.Tensor:
    """numpy cloud -> tensor of the same dtype (the socket path feeds float64), zero normals appended to a
    3-column cloud (util.py:71-77)."""
    np_pc = np.asarray(np_pc)
    if np_pc.shape[1] == 3 and append_normals:
        np_pc = np.concatenate([np_pc, np.zeros((np_pc.shape[0], 3), dtype=np_pc.dtype)], axis=1)
    return torch.tensor(np_pc)


def load_xyz(path, append_normals: bool = True) -> torch.Tensor:
    with open(path, "r") as fh:
        return xyz2tensor(fh.read(), append_normals=append_normals)


class Transform:
    """Centre by the mean and divide by the largest bounding-box extent ('reg'), or the 'bb' variant
    that uses the two extreme points along (1,1,1)."""

    def __init__(self, pc: torch.Tensor, ttype: str = "reg"):
        xyz = pc[:, :3]
        if ttype == "reg":
            self.center = xyz.mean(dim=0)
            self.scale = (xyz.max(dim=0)[0] - xyz.min(dim=0)[0]).max()
        elif ttype == "bb":
            self.center = xyz.mean(dim=0)
            rel = xyz - self.center
            diag = xyz.sum(dim=-1)
            lo, hi = diag.argmin(), diag.argmax()
            self.scale = (rel[hi] - rel[lo]).norm()
            self.center = self.center + (rel[lo] + rel[hi]) / 2
        else:
            raise ValueError(f"unknown transform type {ttype!r}")

    def apply(self, pc: torch.Tensor) -> torch.Tensor:
        out = pc.clone()
        out[:, :3] -= self.center[None, :]
        out[:, :3] = out[:, :3] / self.scale
        return out

    def inverse(self, pc: torch.Tensor) -> torch.Tensor:
        out = pc.clone()
        out[:, :3] = out[:, :3] * self.scale
        out[:, :3] += self.center[None, :]
        return out

    @staticmethod
    def trans(pc: torch.Tensor, ttype: str = "reg"):
        t = Transform(pc, ttype=ttype)
        return t.apply(pc), t


# ---- patch lists -----------------------------------------------------------------------------------
class PatchList(list):
    """A list of per-patch index tensors (what the reference's callers pass around) that remembers its CSR
    form: `flat` = the concatenated indices, `sizes` = the host-side patch sizes.  The entries are views into
    `flat`, so drivers handed a PatchList need neither a torch.cat of hundreds of small tensors nor a device
    round trip to learn the sizes."""

    def __init__(self, flat: torch.Tensor, sizes, disjoint: bool = False):
        self.flat = flat
        self.sizes = [int(n) for n in sizes]
        self.disjoint = bool(disjoint)        # True when the producer guarantees no point is listed twice
        super().__init__(torch.split(flat, self.sizes) if len(self.sizes) else [])


def to_device(arr: np.ndarray, dev) -> torch.Tensor:
    """Small host array -> device tensor without stalling the host: through pinned memory with a non-blocking
    copy (a copy from pageable memory makes the host wait for everything already enqueued on the stream)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if torch.device(dev).type != "cuda":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


class RepLists(list):
    """The `reps` argument of strongest_field_propagation_reps - a list of (representatives, rest) index pairs, one
    per patch (orient_large.py:48-52) - that remembers both sides as PatchLists, so the driver needs no
    concatenation of hundreds of small tensors.  When both PatchLists are flagged disjoint and their sizes add up to
    the cloud, the driver takes representatives and rests as a partition of the cloud (what splitting a partition's
    patches gives) and needs no mask round trip to find the non-representative points."""

    def __init__(self, reps: PatchList, rests: PatchList):
        self.reps, self.rests = reps, rests
        super().__init__(zip(reps, rests))


def patch_csr(patches, dev):
    """(off[P+1] int64 on dev, idx[M] int64 on dev, sizes[P] numpy int64) of a list of index tensors."""
    if isinstance(patches, PatchList) and len(patches) == len(patches.sizes):
        sizes = np.asarray(patches.sizes, dtype=np.int64)
        idx = patches.flat.to(device=dev, dtype=torch.int64)
        cached = getattr(patches, "_off", None)           # the offsets travel to a device once per object (sizes are fixed at construction)
        if cached is None or cached.device != torch.device(dev) or cached.shape[0] != len(sizes) + 1:
            cached = patches._off = to_device(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64), dev)
        return cached, idx, sizes
    else:
        sizes = np.array([int(p.shape[0]) for p in patches], dtype=np.int64)
        idx = torch.cat([p.to(device=dev, dtype=torch.int64) for p in patches]) if len(patches) else \
            torch.zeros(0, dtype=torch.int64, device=dev)
    off = to_device(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64), dev)
    return off, idx, sizes


# ---- voxel partition ---------------------------------------------------------------------------
def _axis_bins(x: torch.Tensor, n_part: int, ranges) -> torch.Tensor:
    """Bin of every coordinate under the reference's test  lo_i < x <= hi_i  with
    lo_i = edge*i + ranges[0], hi_i = lo_i + edge (python doubles compared against the float32
    column, i.e. rounded to float32 first), i = 0..n_part.  -1 = in no bin.  Runs on x's device."""
    edge = (ranges[1] - ranges[0]) / n_part
    lo64 = np.array([edge * i + ranges[0] for i in range(n_part + 1)], dtype=np.float64)
    hi64 = lo64 + edge
    lo = torch.from_numpy(lo64).to(device=x.device, dtype=x.dtype)
    hi = torch.from_numpy(hi64).to(device=x.device, dtype=x.dtype)
    cand = torch.searchsorted(lo, x.contiguous(), right=False) - 1     # largest i with lo_i < x
    ok = cand >= 0
    cand_c = cand.clamp(0, n_part)
    ok &= x <= hi[cand_c]
    return torch.where(ok, cand_c, torch.full_like(cand_c, -1))


def _voxel_cells(pc_in: torch.Tensor, n_part: int, ranges=(-1.5, 1.5)):
    """(order, keys, counts): `order` = the indices of the points that fall into a cell, sorted by cell in
    lexicographic (i, j, k) order and ascending inside a cell (the order the reference's triple loop
    produces, util.py:133-149); keys / counts = one entry per non-empty cell (host numpy).  Binning and
    sorting run on pc_in's device; only the cell table crosses to the host."""
    bx = _axis_bins(pc_in[:, 0], n_part, ranges)
    by = _axis_bins(pc_in[:, 1], n_part, ranges)
    bz = _axis_bins(pc_in[:, 2], n_part, ranges)
    inside = (bx >= 0) & (by >= 0) & (bz >= 0)
    m = n_part + 1
    key = (bx * m + by) * m + bz
    pts = torch.nonzero(inside).flatten()
    skey, perm = torch.sort(key[pts], stable=True)
    order = pts[perm]
    ukeys, counts = torch.unique_consecutive(skey, return_counts=True)
    table = torch.stack([ukeys, counts]).cpu().numpy()                 # ONE small device->host copy
    return order, table[0].astype(np.int64), table[1].astype(np.int64)


def _keys_to_ijk(keys: np.ndarray, n_part: int) -> np.ndarray:
    m = n_part + 1
    return np.stack([keys // (m * m), (keys // m) % m, keys % m], axis=1).astype(np.int32)


def _divide_pc(pc_in: torch.Tensor, n_part: int, ranges=(-1.5, 1.5), min_patch: int = 0):
    """Voxel partition (util.py:110-150): returns (indices, ijk) with one entry per non-empty cell, cells in
    lexicographic (i, j, k) order and point indices ascending inside a cell."""
    order, keys, counts = _voxel_cells(pc_in, n_part, ranges)
    if keys.size == 0:
        return [], []
    return PatchList(order, counts, disjoint=True), [tuple(int(v) for v in row) for row in _keys_to_ijk(keys, n_part)]


def merge_cells(ijk: np.ndarray, sizes: np.ndarray, min_patch: int):
    """The reference's order-dependent merge of small cells (util.merge_nodes, util.py:448-492) on the cell
    table: (seq, seq_off) = original cell ids grouped by surviving patch, in concatenation order.  Runs in
    the native library (host code: the procedure is sequential by definition)."""
    from . import _lib
    lib = _lib.load()
    C = int(len(sizes))
    ijk = np.ascontiguousarray(ijk, dtype=np.int32).reshape(C, 3)
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    seq = np.empty(max(C, 1), dtype=np.int64)
    seq_off = np.empty(C + 1, dtype=np.int64)
    n_out = np.zeros(1, dtype=np.int64)
    sweeps = np.zeros(1, dtype=np.int32)
    _lib.check(lib.dnp_merge_cells(ijk.ctypes.data, sizes.ctypes.data, C, int(min_patch), seq.ctypes.data,
                                   seq_off.ctypes.data, n_out.ctypes.data, sweeps.ctypes.data))
    if int(sweeps[0]) == 10:
        print("recursive merge failed to merge some patches")
    n = int(n_out[0])
    return seq[: int(seq_off[n])], seq_off[: n + 1]


def merge_nodes(indices: List[torch.Tensor], ijk, min_patch: int):
    """Merge cells with fewer than min_patch points into a neighbouring cell (26-neighbourhood of any of its
    constituent cells), as util.merge_nodes (util.py:448-492) including its order dependence: up to 10 sweeps
    over the cells in order; a small cell is appended to the LAST (highest index) live cell that touches it;
    cells still below min_patch at the end are dropped.  Returns (patches, constituent cells per patch)."""
    sizes = np.array([int(t.shape[0]) for t in indices], dtype=np.int64)
    cells = np.asarray(ijk, dtype=np.int32).reshape(len(indices), 3)
    seq, seq_off = merge_cells(cells, sizes, min_patch)
    out_idx = [torch.cat([indices[c] for c in seq[seq_off[g]:seq_off[g + 1]]]) for g in range(len(seq_off) - 1)]
    out_cells = [[tuple(int(v) for v in cells[c]) for c in seq[seq_off[g]:seq_off[g + 1]]]
                 for g in range(len(seq_off) - 1)]
    return out_idx, out_cells


def divide_pc(pc_in: torch.Tensor, n_part: int, ranges=(-1.5, 1.5), min_patch: int = 0) -> List[torch.Tensor]:
    """Voxel partition followed by the merge of small patches: the behaviour the reference's
    callers rely on (`[x.clone() for x in patch_indices]`).  NOTE: the reference's divide_pc as
    committed (util.py:338-341) skips the merge and returns list-wrapped tensors, which makes
    its own callers raise; this is the evidently intended composition.

    The points are binned and sorted by cell on pc_in's device, the cell table (a few thousand rows) is
    merged on the host, and ONE gather assembles every patch's index list; the result is a PatchList."""
    order, keys, counts = _voxel_cells(pc_in, n_part, ranges)
    if keys.size == 0:
        return PatchList(order, [])
    seq, seq_off = merge_cells(_keys_to_ijk(keys, n_part), counts, min_patch)
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    sz = counts[seq]
    sizes = np.add.reduceat(sz, seq_off[:-1]) if len(seq_off) > 1 else np.zeros(0, dtype=np.int64)
    total = int(sz.sum())
    # position p of the output reads order[starts[cell] + (p - first output position of that cell)]
    out_first = np.concatenate([[0], np.cumsum(sz)])[:-1]
    shift = torch.from_numpy((starts[seq] - out_first).astype(np.int64)).to(order.device)
    reps = torch.from_numpy(sz.astype(np.int64)).to(order.device)
    pos = torch.arange(total, device=order.device) + torch.repeat_interleave(shift, reps, output_size=total)
    return PatchList(order[pos], sizes, disjoint=True)        # a partition: every point in at most one patch


def pca_eigen_values(x: torch.Tensor):
    """(smallest covariance eigenvalue as a 1-element tensor, its eigenvector) of x[:, :3]
    (util.pca_eigen_values, util.py:495-500), in x's precision like the reference."""
    rel = x[:, :3] - x.mean(dim=0)[None, :3]
    cov = (rel.transpose(0, 1) @ rel) / x.shape[0]
    e, v = torch.linalg.eigh(cov)   # ascending; the reference's torch.symeig is gone in torch >= 2
    return e[0:1], v[:, 0]


def patch_pca(pc: torch.Tensor, patches):
    """(mean[P,3], evals[P,3] ascending, evecs[P,3,3], csr) of cov = (x - mean)^T (x - mean) / n of every
    listed patch, in fp64 from pc's coordinates, deterministic.  evecs[p][:, k] is eigenvector k with its
    largest-magnitude component positive (the sign is arbitrary in the reference: LAPACK's).  This is the ONE
    place patch covariances are formed: the start-patch rule (field_utils.py:230-233, :303-306), the flatness
    filter (inference_utils.py:52-71) and orient_center (util.py:39-44) all read it.  Device clouds run the
    dnp_patch_pca kernel (one workgroup per patch); CPU clouds the same arithmetic in torch."""
    dev = pc.device
    off, idx, sizes = patch_csr(patches, dev)
    P = len(sizes)
    mean = torch.empty((P, 3), dtype=torch.float64, device=dev)
    evals = torch.empty((P, 3), dtype=torch.float64, device=dev)
    evecs = torch.empty((P, 3, 3), dtype=torch.float64, device=dev)
    if P == 0:
        return mean, evals, evecs, (off, idx, sizes)
    if pc.is_cuda:
        from . import _lib
        lib = _lib.require_device()
        src = pc if pc.dtype in (torch.float32, torch.float64) else pc.float()
        if src.stride(1) != 1:
            src = src.contiguous()
        fn = lib.dnp_patch_pca_f64 if src.dtype == torch.float64 else lib.dnp_patch_pca_f32
        with torch.cuda.device(dev):
            _lib.check(fn(_lib.ptr(src), src.stride(0), _lib.ptr(off), _lib.ptr(idx), P, _lib.ptr(mean),
                          _lib.ptr(evals), _lib.ptr(evecs), _lib.current_stream()))
        return mean, evals, evecs, (off, idx, sizes)
    pid = torch.repeat_interleave(torch.arange(P), torch.from_numpy(sizes))
    xyz = pc[idx, :3].double()
    cnt = torch.from_numpy(sizes).double().clamp(min=1)[:, None]
    mean = torch.zeros((P, 3), dtype=torch.float64).index_add_(0, pid, xyz) / cnt
    rel = xyz - mean[pid]
    outer = (rel[:, :, None] * rel[:, None, :]).reshape(-1, 9)
    cov = (torch.zeros((P, 9), dtype=torch.float64).index_add_(0, pid, outer) / cnt).reshape(P, 3, 3)
    evals, evecs = torch.linalg.eigh(cov)
    big = evecs.abs().argmax(dim=1, keepdim=True)                       # [P,1,3]: row of the largest component
    sgn = torch.where(torch.gather(evecs, 1, big) < 0, -1.0, 1.0)
    return mean, evals, evecs * sgn, (off, idx, sizes)


def fix_n_filter(input_pc: torch.Tensor, patch_indices: List[torch.Tensor], threshold: float):
    """Keep patches whose flatness ratio e0 / (e1 + e2/2) exceeds threshold as (i, idx) pairs; the
    others get their normals aligned with their own PCA normal in place
    (inference_utils.py:52-71 - pure torch, on the callers' path).  One patch_pca call for all patches;
    only the P keep flags cross to the host (the return value is a Python list)."""
    if len(patch_indices) == 0:
        return []
    _, e, v, (off, idx, sizes) = patch_pca(input_pc, patch_indices)
    keep = (e[:, 0] / (e[:, 1] + e[:, 2] / 2)) > threshold
    keep_host = keep.cpu().numpy()
    kept = [(i, patch) for i, patch in enumerate(patch_indices) if bool(keep_host[i])]
    if not keep_host.all():
        P = len(sizes)
        pid = torch.repeat_interleave(torch.arange(P, device=input_pc.device),
                                      to_device(sizes, input_pc.device), output_size=int(sizes.sum()))
        sel = ~keep[pid]
        rows, normal = idx[sel], v[:, :, 0].to(input_pc.dtype)[pid[sel]]
        s = ((input_pc[rows, 3:] * normal).sum(dim=-1) > 0).to(input_pc.dtype) * 2 - 1
        input_pc[rows, 3:] = input_pc[rows, 3:] * s[:, None]
    return kept


def orient_center_patches(input_pc: torch.Tensor, patches: List[torch.Tensor]) -> None:
    """`for p in patches: input_pc[p] = orient_center(input_pc[p])` (orient_pointcloud.py:36-38) for disjoint
    patches in one pass: flip the normals that point towards their patch's centroid."""
    if len(patches) == 0:
        return
    mean, _, _, (off, idx, sizes) = patch_pca(input_pc, patches)
    if torch.bincount(idx, minlength=input_pc.shape[0]).max() > 1:      # overlapping lists: literal loop
        for p in patches:
            input_pc[p] = orient_center(input_pc[p])
        return
    pid = torch.repeat_interleave(torch.arange(len(sizes), device=input_pc.device),
                                  to_device(sizes, input_pc.device), output_size=int(sizes.sum()))
    rel = input_pc[idx, :3] - mean.to(input_pc.dtype)[pid]
    inward = (rel * input_pc[idx, 3:]).sum(dim=-1) < 0
    rows = idx[inward]
    input_pc[rows, 3:] = -input_pc[rows, 3:]


def estimate_normals(pc: torch.Tensor, max_nn: int = 30) -> torch.Tensor:
    """Unoriented PCA normals from the max_nn nearest neighbours (the point itself included), returned
    as [N,6] = (xyz, n).  The reference delegates this to open3d's KD-tree estimator
    (util.py:551-567), which is not available offline; this is an own brute-force counterpart (chunked
    distance matrix + top-k + batched 3x3 eigh, on whatever device pc lives on).  The sign of each
    normal is arbitrary, as with open3d - fixing it is the job of the propagation."""
    xyz = pc[:, :3].contiguous()
    n = xyz.shape[0]
    k = min(max_nn, n)
    normals = torch.empty_like(xyz)
    sq = (xyz * xyz).sum(dim=1)
    step = max(1, min(n, (1 << 26) // max(n, 1)))
    for i in range(0, n, step):
        q = xyz[i:i + step]
        d2 = sq[i:i + step, None] - 2.0 * (q @ xyz.T) + sq[None, :]
        nn = d2.topk(k, dim=1, largest=False).indices            # [m, k]
        nb = xyz[nn]                                             # [m, k, 3]
        rel = nb - nb.mean(dim=1, keepdim=True)
        cov = rel.transpose(1, 2) @ rel / k
        _, v = torch.linalg.eigh(cov)
        normals[i:i + step] = v[:, :, 0]
    return torch.cat([xyz, normals], dim=1)


# ---- kNN graph + breadth-first visiting order (graph.getEMSTfromPC / LinkedListGraph.get_bfs_route) -------------
def knn_graph(xyz: np.ndarray, k: int = 10, threshold: float = 0.1):
    """Directed k-nearest-neighbour graph of graph.getEMSTfromPC (graph.py:380-392; despite its name it builds no
    spanning tree): an edge i -> j for each of i's k nearest neighbours j != i closer than `threshold`.  Returns
    (adjacency: list of sets of ints, mean distance to the k neighbours).  The adjacency sets are real Python
    sets filled in neighbour order, because the reference iterates sets of objects hashed by their end point:
    the same insertion sequence of the same hash values gives the same iteration order, and that order decides
    the BFS route."""
    xyz = np.asarray(xyz, dtype=np.float64)
    try:
        from sklearn.neighbors import KDTree               # what the reference queries (graph.py:345, :382)
        dist, idx = KDTree(xyz).query(xyz, k)
    except ImportError:                                     # same neighbours, ties possibly in another order
        from scipy.spatial import cKDTree
        dist, idx = cKDTree(xyz).query(xyz, k)
        dist, idx = dist.reshape(len(xyz), -1), idx.reshape(len(xyz), -1)
    # the same insertion sequence as one adj[i].add(...) per accepted neighbour, without 10^5 numpy scalar look-ups (65 -> 15 ms
    # at 10 000 points): a set comprehension inserts in iteration order
    keep = ((idx != np.arange(len(xyz))[:, None]) & (dist < threshold)).tolist()
    adj = [{j for j, ok in zip(row, krow) if ok} for row, krow in zip(idx.tolist(), keep)]
    return adj, dist.mean(axis=1)


def bfs_route(adj, start: int):
    """Unweighted breadth-first visiting order from `start` (LinkedListGraph.get_bfs_route, graph.py:293-317):
    FIFO queue, neighbours in the adjacency set's iteration order; when the queue runs dry before every node
    is visited, the unvisited node of smallest index seeds the next component."""
    n = len(adj)
    visited = [False] * n                       # a Python list: the loop below indexes it ~10 n times (numpy scalars cost 3x)
    route, queue, head, units = [], [int(start)], 0, 1
    visited[start] = True
    first_unvisited = 0                          # every node below it is visited: the next component's seed is found in O(n) in all
    while head < len(queue):
        u = queue[head]
        head += 1
        route.append(u)
        for v in adj[u]:
            if not visited[v]:
                visited[v] = True
                queue.append(v)
        if head == len(queue):
            while first_unvisited < n and visited[first_unvisited]:
                first_unvisited += 1
            if first_unvisited == n:
                break
            queue.append(first_unvisited)       # the unvisited node of smallest index
            visited[first_unvisited] = True
            units += 1
    if units != 1:
        print("bfs warning::unit= ", units)
    return route


def timer_factory():
    """A fresh timer class whose instances are `with` blocks printing their wall time; the class
    keeps a running total (print_total_time)."""

    class MyTimer:
        total = 0.0

        def __init__(self, msg="", count=True):
            self.msg, self.count = msg, count

        def __enter__(self):
            self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            dt = time.perf_counter() - self.t0
            if self.count:
                MyTimer.total += dt
            print(f"{self.msg} -- {dt:.3f}s")

        @classmethod
        def print_total_time(cls):
            print(f"total time: {cls.total:.3f}s")

    return MyTimer
