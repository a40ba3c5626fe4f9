"""CPU oracle for the dipole hot path - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A dense-broadcast PyTorch restatement of the reference's algorithm, written from the reference's
behaviour (file:line cited per function, all under /root/reference).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(dipole_normal_prop_amd/) never does and has no CPU fallback.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against the
fixtures under tests/golden/, which tools/gen_golden.py captured by importing the reference
itself in the build container (G1..G12, GH).

The same op-sequence class as the reference on purpose ([S,T,3] broadcast, norm, masked
normalise, divide, sum over S, leaf recursion at 15 000 rows): this is also what bench.py times
as the `cpu_baseline` ("kind": "port") on the GPU box's host cores.
"""
import numpy as np
import torch


# ---- leaves ----------------------------------------------------------------------------------------
def source_leaves(S: int, max_pts: int):
    """Source ranges the reference's recursion ends up summing (field_utils.py:79-94): halve at
    int(n/2) while a range is longer than max_pts.  Returned left to right; the reference adds
    them as a balanced tree, which differs from left-to-right only in fp rounding."""
    out, stack = [], [(0, S)]
    while stack:
        lo, hi = stack.pop()
        if max_pts > 0 and hi - lo > max_pts:
            mid = lo + (hi - lo) // 2
            stack.append((mid, hi))
            stack.append((lo, mid))
        else:
            out.append((lo, hi))
    return out


def _target_blocks(T: int, max_pts: int):
    return source_leaves(T, max_pts)   # same halving rule (field_utils.py:74-77), rows just concatenate


# ---- field / potential -----------------------------------------------------------------------------
def _field_leaf(src: torch.Tensor, tgt: torch.Tensor, eps: float) -> torch.Tensor:
    """One dense leaf (field_utils.py:96-116)."""
    mom = src[:, 3:6]
    sep = src[:, None, :3] - tgt[None, :, :3]                 # [S,T,3]  r = x_s - x_t
    dist = sep.norm(dim=-1)                                    # [S,T]
    coincident = dist == 0
    unit = sep / torch.where(coincident, torch.ones_like(dist), dist)[:, :, None]
    unit = torch.where(coincident[:, :, None], torch.zeros_like(unit), unit)
    along = (mom[:, None, :] * unit).sum(dim=-1)               # p . r^
    contrib = 3 * along[:, :, None] * unit - mom[:, None, :]
    contrib = torch.where(coincident[:, :, None], torch.zeros_like(contrib), contrib)
    contrib = contrib / (dist ** 3 + eps)[:, :, None]
    total = contrib.sum(dim=0) * -1
    bad = total.isinf() | total.isnan()
    return torch.where(bad, torch.zeros_like(total), total)


def _potential_leaf(src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """One dense leaf (field_utils.py:46-55): no eps, no zero mask, Inf/NaN zeroed after the sum."""
    mom = src[:, 3:6]
    sep = src[:, None, :3] - tgt[None, :, :3]
    num = (mom[:, None, :] * sep).sum(dim=-1)
    total = (num / sep.norm(dim=-1) ** 3).sum(dim=0)
    bad = total.isinf() | total.isnan()
    return torch.where(bad, torch.zeros_like(total), total)


def _leafwise(fn, sources, means, recursive, max_pts, width):
    S, T = sources.shape[0], means.shape[0]
    mp = max_pts if recursive else 0
    rows = []
    for (t0, t1) in _target_blocks(T, mp) if T > 0 else []:
        acc = None
        for (s0, s1) in source_leaves(S, mp):
            part = fn(sources[s0:s1], means[t0:t1])
            acc = part if acc is None else acc + part
        rows.append(acc)
    if not rows:
        shape = (0, 3) if width == 3 else (0,)
        return torch.zeros(shape, dtype=torch.result_type(sources, means))
    return torch.cat(rows, dim=0)


def field_grad(sources, means, eps=1e-5, recursive=True, max_pts=15000):
    """field_utils.py:61-116."""
    return _leafwise(lambda s, t: _field_leaf(s, t, eps), sources, means, recursive, max_pts, 3)


def potential(sources, means, eps=1e-5, recursive=True, max_pts=15000):
    """field_utils.py:12-55 (eps unused there as well)."""
    return _leafwise(_potential_leaf, sources, means, recursive, max_pts, 1)


def gen_grid(n=10):
    """util.py:26-36."""
    i = torch.arange(n ** 3)
    pts = torch.stack([i // n // n, (i // n) % n, i % n], dim=1).float()
    return (pts / n - 0.5) * 2


def measure_mean_potential(pc):
    """field_utils.py:7-9."""
    return potential(pc, gen_grid().to(pc.dtype)).mean()


def reference_field(pc1, pc2):
    """field_utils.py:188-201 (returns a new tensor in both cases; the reference mutates the
    6-column input in place)."""
    E = field_grad(pc1, pc2)
    if pc2.shape[1] == 3:
        n = E.norm(dim=-1)
        E = torch.where((n != 0)[:, None], E / torch.where(n != 0, n, torch.ones_like(n))[:, None], E)
        return torch.cat([pc2, E], dim=1)
    sign = ((E * pc2[:, 3:]).sum(dim=-1) >= 0).to(pc2.dtype) * 2 - 1
    out = pc2.clone()
    out[:, 3:] = out[:, 3:] * sign[:, None]
    return out


def field_edge_calculator(sources, means):
    """field_utils.py:145-160: (w, -w) with w = 2 * sum(E.n) / |S| * |T|."""
    inter = (field_grad(sources, means) * means[:, 3:]).sum(dim=-1).sum()
    w = ((inter * 2) / sources.shape[0] * means.shape[0]).numpy()
    return w, -w


# ---- host helpers ----------------------------------------------------------------------------------
def pca_min_eigen(x):
    """util.py:495-500: smallest eigenvalue of the covariance of x[:, :3]."""
    rel = x[:, :3] - x.mean(dim=0)[None, :3]
    cov = rel.transpose(0, 1) @ rel / x.shape[0]
    return torch.linalg.eigvalsh(cov)[0]


# ---- greedy drivers --------------------------------------------------------------------------------
def strongest_field_propagation(pts, patches, all_patches, diffuse=False, weights=None, start_patch=None):
    """field_utils.py:286-348.  Works on a clone; returns (new_pts, trace) with trace = dict(order,
    flipped, chosen)."""
    pts = pts.clone()
    if weights is not None:
        w = weights.clamp(0.1, 1)
        pts[:, 3:] = pts[:, 3:] * w[:, None]
    N, P = pts.shape[0], len(all_patches)
    E = torch.zeros(N, 3, dtype=pts.dtype)
    if start_patch is None:
        start_patch = int(np.argmin([abs(float(pca_min_eigen(pts[p]))) for p in all_patches]))
    todo = [k for k in range(P) if k != start_patch]
    done = torch.zeros(N, dtype=torch.bool)
    done[all_patches[start_patch]] = True
    E[~done] = field_grad(pts[done], pts[~done])
    order, flipped, chosen = [start_patch], [False], []
    while todo:
        scores = torch.stack([(E[all_patches[k]] * pts[all_patches[k], 3:]).sum(dim=-1).sum() for k in todo])
        m = int(scores.abs().argmax())
        k = todo.pop(m)
        chosen.append(float(scores[m]))
        flip = bool(scores[m] < 0)
        if flip:
            pts[all_patches[k], 3:] *= -1
        done[all_patches[k]] = True
        order.append(k)
        flipped.append(flip)
        if diffuse:
            others = torch.ones(N, dtype=torch.bool)
            others[all_patches[k]] = False
        else:
            others = ~done
        E[others] = E[others] + field_grad(pts[all_patches[k]], pts[others])
    if diffuse:
        for _, p in patches:
            s = ((E[p] * pts[p, 3:]).sum(dim=-1) > 0).to(pts.dtype) * 2 - 1
            pts[p, 3:] = pts[p, 3:] * s[:, None]
    if weights is not None:
        pts[:, 3:] = pts[:, 3:] / w[:, None]
    return pts, dict(order=np.array(order), flipped=np.array(flipped), chosen=np.array(chosen))


def strongest_field_propagation_reps(pts, reps, diffuse=False, start_patch=None):
    """field_utils.py:207-282 (weights omitted: the callers never pass them)."""
    pts = pts.clone()
    N, P = pts.shape[0], len(reps)
    E = torch.zeros(N, 3, dtype=pts.dtype)
    is_rep = torch.zeros(N, dtype=torch.bool)
    for r, _ in reps:
        is_rep[r] = True
    pending = is_rep.clone()
    if start_patch is None:
        start_patch = int(np.argmin([abs(float(pca_min_eigen(pts[r]))) for r, _ in reps]))
    todo = [k for k in range(P) if k != start_patch]
    done = torch.zeros(N, dtype=torch.bool)
    done[reps[start_patch][0]] = True
    pending[reps[start_patch][0]] = False
    E[pending] = field_grad(pts[done], pts[pending])
    order, flipped, chosen = [start_patch], [False], []
    while todo:
        scores = torch.stack([(E[reps[k][0]] * pts[reps[k][0], 3:]).sum(dim=-1).sum() for k in todo])
        m = int(scores.abs().argmax())
        k = todo.pop(m)
        rep, rest = reps[k]
        chosen.append(float(scores[m]))
        flip = bool(scores[m] < 0)
        if flip:
            pts[rep, 3:] *= -1
            pts[rest, 3:] *= -1
        done[rep] = True
        pending[rep] = False
        order.append(k)
        flipped.append(flip)
        if diffuse:
            others = is_rep.clone()
            others[rep] = False
        else:
            others = pending.clone()
        E[others] = E[others] + field_grad(pts[rep], pts[others])
    if diffuse:
        for rep, _ in reps:
            s = ((E[rep] * pts[rep, 3:]).sum(dim=-1) > 0).to(pts.dtype) * 2 - 1
            pts[rep, 3:] = pts[rep, 3:] * s[:, None]
    if (~done).any():
        E2 = field_grad(pts[done], pts[~done])
        s = ((E2 * pts[~done, 3:]).sum(dim=-1) > 0).to(pts.dtype) * 2 - 1
        pts[~done, 3:] = pts[~done, 3:] * s[:, None]
    return pts, dict(order=np.array(order), flipped=np.array(flipped), chosen=np.array(chosen))


def strongest_field_propagation_points(pts, diffuse=False, starting_point=0):
    """field_utils.py:353-388 (eps = 1e-6 per step).  Returns (new_pts, visit order)."""
    pts = pts.clone()
    N = pts.shape[0]
    E = torch.zeros(N, 3, dtype=pts.dtype)
    seen = torch.zeros(N, dtype=torch.bool)
    ids = torch.arange(N)
    cur = int(starting_point)
    order = []
    for step in range(N):
        seen[cur] = True
        order.append(cur)
        rest = ids != cur
        E[rest] += field_grad(pts[cur:cur + 1], pts[rest, :3], eps=1e-6)
        if step + 1 == N:
            break
        score = (E[~seen] * pts[~seen, 3:]).sum(dim=-1)
        m = score.abs().argmax()
        cur = int(ids[~seen][m])
        if score[m] < 0:
            pts[cur, 3:] *= -1
    if diffuse:
        s = ((E * pts[:, 3:]).sum(dim=-1) > 0).to(pts.dtype) * 2 - 1
        pts[:, 3:] = pts[:, 3:] * s[:, None]
    return pts, np.array(order)


# ---- the fork's "xie" pair functions ----------------------------------------------------------------
def xie_field(source, target, C=3):
    """field_utils.py:431-469 without the kNN mask (eps is unused there): [T,S,3]."""
    sep = source[None, :, :3] - target[:, None, :3]
    dist = sep.norm(dim=-1)
    apart = dist != 0
    unit = torch.where(apart[:, :, None], sep / torch.where(apart, dist, torch.ones_like(dist))[:, :, None], sep)
    n = source[None, :, 3:6]
    refl = n - C * (n * unit).sum(dim=-1)[:, :, None] * unit
    return torch.where(apart[:, :, None], refl / torch.where(apart, dist, torch.ones_like(dist))[:, :, None] ** 3, refl)


def xie_knn_mask(source, target, k):
    """field_utils.py:451-460: 1 where target t is among the k nearest targets (scipy KDTree) of source s."""
    from scipy.spatial import KDTree
    k = min(target.shape[0], k)
    _, idx = KDTree(target[:, :3].numpy()).query(source[:, :3].numpy(), k=k)
    idx = np.asarray(idx).reshape(source.shape[0], -1)
    mask = torch.zeros(target.shape[0], source.shape[0], dtype=torch.float64)
    for s in range(source.shape[0]):
        mask[idx[s], s] = 1.0
    return mask


def xie_intersaction(source, target, knn_mask=-1, C=3):
    """field_utils.py:509-519."""
    f = xie_field(source, target, C)
    if knn_mask > 0:
        f = f * xie_knn_mask(source, target, knn_mask)[:, :, None]
    m = (f * target[:, None, 3:6]).sum(dim=-1)
    return torch.where(m.isnan() | m.isinf(), torch.zeros_like(m), m)


def xie_distance(source, target):
    """field_utils.py:522-526."""
    return (source[None, :, 3:] * (source[None, :, :3] - target[:, None, :3])).norm(dim=-1).sum(dim=-1)


def xie_propagation_points_in_order(pts, order, diffuse=False, knn_mask=-1, C=3):
    """field_utils.py:569-605: [T,N] bool, True where the summed interaction is negative."""
    order = torch.as_tensor(np.asarray(order)).long()
    T, N = order.shape
    M = xie_intersaction(pts, pts, knn_mask, C).to(pts.dtype)
    w = torch.zeros(T, N, dtype=pts.dtype)
    inter = torch.zeros(T, N, dtype=pts.dtype)
    rows = torch.arange(T)
    for i in range(N):
        idx = order[:, i]
        inter[rows, idx] = (M[idx] * w).sum(dim=-1)
        w[rows, idx] = torch.where(inter[rows, idx] < 0, -1.0, 1.0).to(pts.dtype)
    if diffuse:
        inter = (M[None, :, :] * w[:, None, :]).sum(dim=-1)
    return inter < 0
