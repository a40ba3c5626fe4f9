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
import time
from typing import List, Tuple

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
    Format as the reference: str(float) joined by ' ', lines joined by '\\n', no trailing newline."""
    rows = pc.transpose(0, 1).detach().cpu().tolist()
    text = "\n".join(" ".join(str(v) for v in row) for row in rows)
    with open(dest, "w+") as fh:
        fh.write(text)


def xyz2tensor(txt: str, append_normals: bool = True) -> torch.Tensor:
    """Parse '.xyz' text: space separated, 3 or 6 columns per line; lines containing 'nan' are
    dropped; 3-column lines get zero normals appended when append_normals."""
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


# ---- voxel partition ---------------------------------------------------------------------------
def _axis_bins(x: torch.Tensor, n_part: int, ranges) -> np.ndarray:
    """Bin of every coordinate under the reference's test  lo_i < x <= hi_i  with
    lo_i = edge*i + ranges[0], hi_i = lo_i + edge (python doubles compared against the float32
    column, i.e. rounded to float32 first), i = 0..n_part.  -1 = in no bin."""
    edge = (ranges[1] - ranges[0]) / n_part
    lo64 = np.array([edge * i + ranges[0] for i in range(n_part + 1)], dtype=np.float64)
    hi64 = lo64 + edge
    xv = x.detach().cpu().numpy()
    lo = lo64.astype(xv.dtype)
    hi = hi64.astype(xv.dtype)
    cand = np.searchsorted(lo, xv, side="left") - 1          # largest i with lo_i < x
    ok = cand >= 0
    cand_c = np.clip(cand, 0, n_part)
    ok &= xv <= hi[cand_c]
    return np.where(ok, cand_c, -1)


def _divide_pc(pc_in: torch.Tensor, n_part: int, ranges=(-1.5, 1.5), min_patch: int = 0):
    """Voxel partition: returns (indices, ijk) with one entry per non-empty cell, cells in
    lexicographic (i, j, k) order and point indices ascending inside a cell - the order the
    reference's triple loop produces."""
    bx = _axis_bins(pc_in[:, 0], n_part, ranges)
    by = _axis_bins(pc_in[:, 1], n_part, ranges)
    bz = _axis_bins(pc_in[:, 2], n_part, ranges)
    inside = (bx >= 0) & (by >= 0) & (bz >= 0)
    m = n_part + 1
    key = (bx.astype(np.int64) * m + by) * m + bz
    pts = np.nonzero(inside)[0]
    order = pts[np.argsort(key[pts], kind="stable")]
    skey = key[order]
    starts = np.nonzero(np.r_[True, skey[1:] != skey[:-1]])[0] if len(order) else np.array([], dtype=np.int64)
    ends = np.r_[starts[1:], len(order)] if len(order) else starts
    dev = pc_in.device
    indices, ijk = [], []
    for s, e in zip(starts, ends):
        kk = int(skey[s])
        indices.append(torch.from_numpy(order[s:e].copy()).to(dev))
        ijk.append((kk // (m * m), (kk // m) % m, kk % m))
    return indices, ijk


def merge_nodes(indices: List[torch.Tensor], ijk: List[Tuple[int, int, int]], min_patch: int):
    """Merge cells with fewer than min_patch points into a neighbouring cell (26-neighbourhood of
    any of its constituent cells).  Mirrors the reference's procedure including its order
    dependence: up to 10 sweeps over the cells in order; a small cell is appended to the LAST
    (highest index) live cell that touches it; cells still below min_patch at the end are dropped."""
    pts = [[t] for t in indices]               # list of tensors per live cell (concatenated at the end)
    size = [int(t.shape[0]) for t in indices]
    cells = [[c] for c in ijk]                 # constituent voxel coordinates per live cell
    live = [True] * len(indices)

    def touches(a, b):
        for ca in a:
            for cb in b:
                if abs(ca[0] - cb[0]) <= 1 and abs(ca[1] - cb[1]) <= 1 and abs(ca[2] - cb[2]) <= 1:
                    return True
        return False

    sweeps, again = 0, True
    while again and sweeps < 10:
        again = False
        sweeps += 1
        for i in range(len(cells)):
            if not live[i] or size[i] >= min_patch:
                continue
            target = -1
            for j in range(len(cells)):
                if j != i and live[j] and touches(cells[i], cells[j]):
                    target = j
            if target < 0:
                continue
            pts[target].extend(pts[i])
            size[target] += size[i]
            cells[target].extend(cells[i])
            live[i] = False
            pts[i], cells[i], size[i] = [], [], 0
            if size[target] < min_patch:
                again = True
    if sweeps == 10:
        print("recursive merge failed to merge some patches")
    out_idx, out_cells = [], []
    for i in range(len(cells)):
        if live[i] and size[i] >= min_patch:
            out_idx.append(torch.cat(pts[i]))
            out_cells.append(cells[i])
    return out_idx, out_cells


def divide_pc(pc_in: torch.Tensor, n_part: int, ranges=(-1.5, 1.5), min_patch: int = 0) -> List[torch.Tensor]:
    """Voxel partition followed by the merge of small patches: the behaviour the reference's
    callers rely on (`[x.clone() for x in patch_indices]`).  NOTE: the reference's divide_pc as
    committed (util.py:338-341) skips the merge and returns list-wrapped tensors, which makes
    its own callers raise; this is the evidently intended composition."""
    indices, ijk = _divide_pc(pc_in, n_part, ranges, min_patch)
    merged, _ = merge_nodes(indices, ijk, min_patch)
    return merged


def pca_eigen_values(x: torch.Tensor):
    """(smallest covariance eigenvalue as a 1-element tensor, its eigenvector) of x[:, :3]."""
    rel = x[:, :3] - x.mean(dim=0)[None, :3]
    cov = (rel.transpose(0, 1) @ rel) / x.shape[0]
    e, v = torch.linalg.eigh(cov)   # ascending; the reference's torch.symeig is gone in torch >= 2
    return e[0:1], v[:, 0]


def fix_n_filter(input_pc: torch.Tensor, patch_indices: List[torch.Tensor], threshold: float):
    """Keep patches whose flatness ratio e0 / (e1 + e2/2) exceeds threshold as (i, idx) pairs; the
    others get their normals aligned with their own PCA normal in place
    (inference_utils.py:52-71 - pure torch, on the callers' path)."""
    kept = []
    for i, patch in enumerate(patch_indices):
        x = input_pc[patch]
        rel = x[:, :3] - x.mean(dim=0)[None, :3]
        cov = (rel.transpose(0, 1) @ rel) / x.shape[0]
        e, v = torch.linalg.eigh(cov)
        if (e[0] / (e[1] + e[2] / 2)).item() > threshold:
            kept.append((i, patch))
        else:
            s = ((input_pc[patch, 3:] * v[:, 0][None, :]).sum(dim=-1) > 0).to(input_pc.dtype) * 2 - 1
            input_pc[patch, 3:] = input_pc[patch, 3:] * s[:, None]
    return kept


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
