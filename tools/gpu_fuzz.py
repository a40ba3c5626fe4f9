#!/usr/bin/env python3
"""Randomised cross-check of the round-3 entry points against the fp64 C oracle and against each other, for a time budget
(default 180 s):  random clouds (clustered, with coincident points), random patch cuts (sizes 1 .. 700, empty patches, rows in
no patch, ragged last tiles), random eps;
  * dnp_patch_fields_tiled_f32 with / without the tile table, with / without interaction partials, source_split 1 / -k (split tail through the exchange buffer) / all split:
    slabs bit-identical in every combination, each slab row within 1e-5 of the oracle (rows of cancellation-heavy random
    clouds get the documented 16 u sum-of-terms allowance),
  * dnp_interactions_from_tiles against dnp_interactions_f32 where the tiles allow the fused form,
  * dnp_reference_field_* against field_grad + the torch tail (both forms, fp32 / fp64),
  * field_grad on ragged shapes against the oracle,
  * every 40th case one of the heavier kinds: field_grad / reference_field at 4 10^8 .. 2.7 10^9 pairs (scalar kernel, far
    launches, source split), the patch driver against the oracle's driver (visit order, normals), potential on ragged
    shapes, the per-point driver in both forms, the blocked xie ordered propagation against the row-per-step kernels (with / without a
    kNN mask and a non-permutation row).
Prints one line per failure and a summary; exit code 1 on any failure.

    python tools/gpu_fuzz.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import _lib  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from dipole_normal_prop_amd import point_driver as ptd  # noqa: E402
from oracle import c_oracle  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def cloud(n):
    k = int(rng.integers(1, 12))
    centres = rng.uniform(-0.5, 0.5, (k, 3))
    x = centres[rng.integers(0, k, n)] + rng.normal(0, 10.0 ** rng.uniform(-3, -1), (n, 3))
    if n > 8 and rng.random() < 0.3:                     # some coincident points
        dup = rng.integers(0, n, max(1, n // 50))
        x[dup] = x[rng.integers(0, n, len(dup))]
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return torch.from_numpy(np.concatenate([x, nrm], 1).astype(np.float32))


def rel_rows(a, b, allow=None):
    den = np.linalg.norm(b, axis=-1)
    den = np.where(den == 0, 1.0, den)
    err = np.linalg.norm(a - b, axis=-1)
    if allow is not None:
        err = np.maximum(err - allow, 0)
    return float((err / den).max()) if len(den) else 0.0


def term_allowance(src, tgt, eps):
    """16 u * sum over sources of |term|: the documented allowance for cancellation residues."""
    r = src[None, :, :3].astype(np.float64) - tgt[:, None, :3].astype(np.float64)
    d = np.linalg.norm(r, axis=-1)
    mag = 4.0 / (d ** 3 + eps)
    mag[d == 0] = 0
    return 16 * 6e-8 * mag.sum(axis=1)


def run(budget=180.0, seed=0):
    """Returns (number of cases, list of failure descriptions)."""
    global rng
    rng = np.random.default_rng(seed)
    fails, cases = [], 0
    t_end = time.time() + budget
    t_note = time.time() + 60.0
    stop_at = int(os.environ.get("FUZZ_STOP_CASE", "0"))
    while time.time() < t_end and not (stop_at and cases >= stop_at):
        if time.time() > t_note:                             # a sign of life once a minute (long runs under gpurun)
            print(f"  ... {cases} cases, {len(fails)} failures", flush=True)
            t_note = time.time() + 60.0
        cases += 1
        kind = cases % 3 if cases % 40 else 3 + (cases // 40) % 5      # every 40th case: one of the heavier kinds
        try:
            if kind == 0:                                    # patch slabs
                sizes = []
                n_p = int(rng.integers(1, 14))
                for _ in range(n_p):
                    r = rng.random()
                    sizes.append(0 if r < 0.05 else int(rng.integers(1, 130)) if r < 0.35 else int(rng.integers(129, 513)) if r < 0.85
                                 else int(rng.integers(513, 700)))
                sizes = np.array(sizes, dtype=np.int64)
                loose = int(rng.integers(0, 200)) if rng.random() < 0.5 else 0
                N = int(sizes.sum()) + loose
                if N == 0:
                    continue
                pc = cloud(N)
                sw = pc.to(dev)
                off_np = np.concatenate([[0], np.cumsum(sizes)])
                off = torch.from_numpy(off_np).to(dev)
                pp = torch.cat([torch.repeat_interleave(torch.arange(n_p, device=dev), off[1:] - off[:-1]),
                                torch.full((loose,), -1, dtype=torch.int64, device=dev)])
                eps = float(10.0 ** rng.uniform(-7, -4))
                boxes, tiles = fu._patch_boxes(sw, off, None), fu._TileTables(sw, sizes)
                base = fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes)
                tail = -int(rng.integers(1, n_p + 2))             # one launch, its last k patches split (k > n_p: all of them)
                allp = -(n_p + 1)                                 # every patch of the launch split
                slots = tiles.slots if tiles.slots else 2        # (partials are written for ANY cut and only used where the tiles allow)
                wp = {ss: torch.full((n_p, tiles.n_tiles, slots), float("nan"), dtype=torch.float64, device=dev) for ss in (1, allp, tail)}
                variants = {"tile table": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes),
                            "partials": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes, wp[1], 1),
                            "all split": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes, None, allp),
                            "all split + partials": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes, wp[allp], allp),
                            f"split {tail}": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes, None, tail),
                            f"split {tail} + partials": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps, boxes, tiles.boxes, wp[tail], tail),
                            "no boxes": fu._patch_slabs(sw, off, None, pp, 0, n_p, eps)}
                for name, v in variants.items():
                    if not torch.equal(v, base):
                        fails.append(f"case {cases}: slabs differ with '{name}' (sizes {sizes.tolist()}, loose {loose})")
                if not torch.equal(wp[1], wp[allp]) or not torch.equal(wp[1], wp[tail]):
                    fails.append(f"case {cases}: partials differ between source_split 1, all split and {tail} (sizes {sizes.tolist()})")
                if tiles.fused:
                    W3 = fu._interaction_rows(base, sw, off, None)
                    Wt = torch.empty_like(W3)
                    lib = _lib.require_device()
                    _lib.check(lib.dnp_interactions_from_tiles(_lib.ptr(wp[1]), slots, n_p, N, _lib.ptr(pp), _lib.ptr(off), n_p,
                                                               _lib.ptr(Wt), _lib.current_stream()))
                    scale = float(W3.abs().max())
                    if scale > 0 and float((Wt - W3).abs().max()) > 1e-11 * scale:
                        fails.append(f"case {cases}: W from tiles differs from K3 by {float((Wt - W3).abs().max()) / scale:.2e}")
                k = int(rng.integers(0, n_p))
                if sizes[k] > 0:
                    others = (pp != k).cpu().numpy()
                    src = pc[off_np[k]:off_np[k + 1]].numpy()
                    ref = c_oracle.field_grad_f64(src, pc[others].numpy(), eps=eps)
                    e = rel_rows(base[k].cpu().numpy()[others].astype(np.float64), ref, term_allowance(src, pc[others].numpy(), eps))
                    if e > 1e-5:
                        fails.append(f"case {cases}: slab {k} off by {e:.2e} (sizes {sizes.tolist()}, eps {eps:.1e})")
                    if float(base[k][off_np[k]:off_np[k + 1]].abs().max()) != 0:
                        fails.append(f"case {cases}: own rows of slab {k} are not zero")
            elif kind == 1:                                  # reference_field, fused against two-step
                S, T = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
                src, tgt = cloud(S), cloud(T)
                for dt in (torch.float32, torch.float64):
                    a = src.to(dev).to(dt)
                    for cols in (3, 6):
                        b1 = tgt[:, :cols].contiguous().to(dev).to(dt)
                        b2 = b1.clone()
                        one = fu.reference_field(a, b1)
                        saved = fu._reference_field_fused
                        fu._reference_field_fused = lambda p, q: None
                        try:
                            two = fu.reference_field(a, b2)
                        finally:
                            fu._reference_field_fused = saved
                        if cols == 6:
                            E = fu.field_grad(a, tgt.to(dev).to(dt))
                            margin = ((E * tgt[:, 3:].to(dev).to(dt)).sum(-1).abs() / E.norm(dim=1).clamp(min=1e-300)).cpu().numpy()
                            differ = (one != two).any(dim=1).cpu().numpy()
                            if differ.any() and margin[differ].max() > 1e-5:
                                fails.append(f"case {cases}: fused reference_field signs differ at margin {margin[differ].max():.2e} ({dt})")
                        else:
                            tol = 5e-6 if dt == torch.float32 else 1e-12
                            if float((one[:, 3:] - two[:, 3:]).abs().max()) > tol:
                                fails.append(f"case {cases}: fused reference_field normals off by {float((one[:, 3:] - two[:, 3:]).abs().max()):.2e} ({dt})")
            else:                                            # plain field_grad on ragged shapes
                S, T = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
                src, tgt = cloud(S), cloud(T)
                eps = float(10.0 ** rng.uniform(-7, -4))
                E = fu.field_grad(src.to(dev), tgt.to(dev), eps=eps).cpu().numpy().astype(np.float64)
                ref = c_oracle.field_grad_f64(src.numpy(), tgt.numpy(), eps=eps)
                e = rel_rows(E, ref, term_allowance(src.numpy(), tgt.numpy(), eps))
                if e > 1e-5:
                    fails.append(f"case {cases}: field_grad {S}x{T} eps {eps:.1e} off by {e:.2e}")
            if kind == 3:                                    # field_grad around / above 10^9 pairs (scalar kernel, far launches, source split)
                S, T = int(rng.integers(20000, 52000)), int(rng.integers(20000, 52000))
                src, tgt = cloud(S), cloud(T)
                if rng.random() < 0.5:                       # spatially sorted: the far chains are really taken
                    src = src[np.argsort(np.floor((src[:, 0].numpy() + 1) * 16) * 1024 + np.floor((src[:, 1].numpy() + 1) * 16))]
                    tgt = tgt[np.argsort(np.floor((tgt[:, 0].numpy() + 1) * 16) * 1024 + np.floor((tgt[:, 1].numpy() + 1) * 16))]
                eps = float(10.0 ** rng.uniform(-6, -4))
                E = fu.field_grad(src.to(dev), tgt.to(dev), eps=eps).cpu().numpy().astype(np.float64)
                rows = rng.choice(T, 96, replace=False)
                ref = c_oracle.field_grad_f64(src.numpy(), tgt.numpy()[rows], eps=eps)
                e = rel_rows(E[rows], ref, term_allowance(src.numpy(), tgt.numpy()[rows], eps))
                if e > 1e-5 or not np.isfinite(E).all():
                    fails.append(f"case {cases}: field_grad {S}x{T} (large) eps {eps:.1e} off by {e:.2e}")
                out = fu.reference_field(src.to(dev), tgt[:, :3].contiguous().to(dev))     # the fused tail on a large call
                nrm = ref / np.linalg.norm(ref, axis=1, keepdims=True)
                if eps == 1e-5 and np.abs(out.cpu().numpy()[rows, 3:] - nrm).max() > 1e-4:
                    fails.append(f"case {cases}: reference_field {S}x{T} normals off")
            elif kind == 4:                                  # the patch driver against the oracle's driver
                from oracle import dipole_oracle as O
                n_p = int(rng.integers(2, 9))
                sizes = rng.integers(20, 260, n_p)
                N = int(sizes.sum())
                pc = cloud(N)
                cuts = np.concatenate([[0], np.cumsum(sizes)])
                perm = rng.permutation(N)
                patches = [torch.from_numpy(np.sort(perm[cuts[k]:cuts[k + 1]])) for k in range(n_p)]
                diffuse = bool(rng.integers(0, 2))
                start = int(rng.integers(0, n_p))
                a = pc.clone().to(dev)
                fu.strongest_field_propagation(a, [(i, p.to(dev)) for i, p in enumerate(patches)], [p.to(dev) for p in patches],
                                               diffuse=diffuse, start_patch=start)
                tr = fu.last_trace("patches")
                ref, rtr = O.strongest_field_propagation(pc.clone(), list(enumerate(patches)), patches, diffuse=diffuse, start_patch=start)
                if not np.array_equal(tr["order"], rtr["order"]):
                    # a different order is a failure only when the oracle's own decision was not a near-tie
                    ch = np.abs(np.asarray(rtr["chosen"], dtype=np.float64))
                    fails.append(f"case {cases}: patch driver order differs (P={n_p}, N={N}, diffuse={diffuse}, min |chosen| {ch.min():.3e})")
                elif not diffuse and not torch.equal(a.cpu()[:, 3:], ref[:, 3:]):
                    fails.append(f"case {cases}: patch driver normals differ (P={n_p}, N={N})")
            elif kind == 5:                                  # potential on ragged shapes
                S, T = int(rng.integers(1, 5000)), int(rng.integers(1, 3000))
                src, tgt = cloud(S), cloud(T)
                tgt[:, :3] += 0.37                            # keep targets off the sources: the potential has no eps
                phi = fu.potential(src.to(dev), tgt.to(dev)).cpu().numpy().astype(np.float64)
                ref = c_oracle.potential_f64(src.numpy(), tgt.numpy())
                # fp32 terms n.r / |r|^3 carry 6e-8 |n||r| of rounding in n.r whatever its own size: the same 16 u sum-of-|term|
                # allowance as for the field rows (seed 2025 met a near-perpendicular close pair: 1.6e-5 of max |phi| without it)
                rr = src.numpy()[None, :, :3].astype(np.float64) - tgt.numpy()[:, None, :3].astype(np.float64)
                dd = np.linalg.norm(rr, axis=-1)
                allow = 16 * 6e-8 * np.where(dd > 0, np.linalg.norm(src.numpy()[None, :, 3:], axis=-1) / np.maximum(dd, 1e-300) ** 2, 0).sum(axis=1)
                err = np.maximum(np.abs(phi - ref) - allow, 0)
                if os.environ.get("FUZZ_DETAIL_CASE") == str(cases):      # replay of a reported case: what the numbers are
                    from oracle import dipole_oracle as O
                    cpu32 = O.potential(src, tgt).numpy().astype(np.float64)
                    i = int(np.argmax(np.abs(phi - ref)))
                    print(f"  detail case {cases}: max |phi64| {np.abs(ref).max():.4e}; worst target {i}: phi64 {ref[i]:.6e}, HIP error "
                          f"{abs(phi[i] - ref[i]):.3e}, reference-class torch fp32 error there {abs(cpu32[i] - ref[i]):.3e} (its own max "
                          f"{np.abs(cpu32 - ref).max():.3e}), closest source at {dd[i].min():.3e}, sum of |term| bounds {allow[i] / (16 * 6e-8):.4e}, "
                          f"allowance {allow[i]:.3e}", flush=True)
                if err.max() > 1e-5 * max(np.abs(ref).max(), 1e-30):
                    fails.append(f"case {cases}: potential {S}x{T} off by {err.max() / np.abs(ref).max():.2e} beyond the term allowance")
            elif kind == 6:                                  # per-point driver, both forms: a complete visit order, forms agree
                N = int(rng.integers(2, 260))
                pc = cloud(N)
                outs = []
                for form in (1, 2):
                    ptd.POINT_GREEDY_FORM = form
                    try:
                        b = pc.clone().to(dev)
                        fu.strongest_field_propagation_points(b, diffuse=True, starting_point=0)
                    finally:
                        ptd.POINT_GREEDY_FORM = 0
                    if sorted(fu.last_trace("points")["order"].tolist()) != list(range(N)):
                        fails.append(f"case {cases}: per-point order is not a permutation (N={N}, form {form})")
                    outs.append((b.cpu(), fu.last_trace("points")["order"].copy()))
                if not (torch.equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
                    fails.append(f"case {cases}: the two per-point forms disagree (N={N})")
            elif kind == 7:                                  # xie: blocked ordered propagation against the row-per-step kernels, kNN
                N, R = int(rng.integers(512, 3000)), int(rng.integers(1, 7))
                f64 = bool(rng.random() < 0.4)
                pc = cloud(N).to(dev)
                pc = pc.double() if f64 else pc
                lib = _lib.require_device()
                knn = int(rng.integers(1, 80)) if rng.random() < 0.5 else -1
                M = fu.xie_intersaction(pc, pc, 0.1, knn, 3).contiguous()
                orders = np.stack([rng.permutation(N) for _ in range(R)]).astype(np.int64)
                if rng.random() < 0.3:                       # one row that is not a permutation
                    orders[0, N // 2] = orders[0, N // 3]
                ot = torch.from_numpy(orders).to(dev)
                w1, i1 = torch.empty((R, N), dtype=M.dtype, device=dev), torch.empty((R, N), dtype=M.dtype, device=dev)
                w2, i2 = torch.empty_like(w1), torch.empty_like(i1)
                nb = lib.dnp_xie_order_workspace_bytes(N, R, 8 if f64 else 4)
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                rc1 = (lib.dnp_xie_order_f64 if f64 else lib.dnp_xie_order_f32)(_lib.ptr(M), N, _lib.ptr(ot), R, _lib.ptr(w1), _lib.ptr(i1), _lib.current_stream())
                rc2 = (lib.dnp_xie_order_blocked_f64 if f64 else lib.dnp_xie_order_blocked_f32)(_lib.ptr(M), N, _lib.ptr(ot), R, _lib.ptr(w2), _lib.ptr(i2),
                                                                                               _lib.ptr(ws), nb, _lib.current_stream())
                if rc1 or rc2:
                    fails.append(f"case {cases}: xie order rc {rc1} / {rc2}")
                elif not torch.equal(w1, w2):
                    # a sign may differ only where the sum itself is rounding noise of its terms
                    bad = (w1 != w2)
                    mag = (M.abs().double() @ torch.ones(N, dtype=torch.float64, device=dev))
                    r_, c_ = torch.nonzero(bad, as_tuple=True)
                    worst = float((i1[bad].abs().double() / mag[c_]).max())
                    if worst > 1e-12:
                        fails.append(f"case {cases}: blocked xie order signs differ (N={N}, R={R}, f64={f64}, knn={knn}, {int(bad.sum())} entries, |inter|/sum|M| up to {worst:.2e})")
        except Exception as exc:                             # a library error is a failure too
            fails.append(f"case {cases} (kind {kind}): {type(exc).__name__}: {exc}")
        if len(fails) > 20:
            break
    fu.flush_warnings()
    return cases, fails


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_cases, failures = run(budget, seed)
    for f in failures:
        print("FAIL", f)
    print(f"{n_cases} cases in {budget:.0f} s (seed {seed}): {len(failures)} failures")
    sys.exit(1 if failures else 0)
