#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation (build container only).

Imports /root/reference/field_utils.py *in place* (nothing is copied) with empty stub
modules for the three optional dependencies that are absent offline (open3d, pymeshlab,
gurobipy - none of them is touched by the field arithmetic), plus two API shims the
reference needs on torch >= 2 / CPU-only hosts:
    torch.symeig  -> torch.linalg.eigh      (util.py:498, inference_utils.py:57)
    Tensor.cuda() -> identity               (field_utils.py:355 hard-codes .cuda())
and writes small .npz fixtures under tests/golden/.  The fixtures (inputs + expected
outputs) are data; they travel to the GPU box, the reference does not.

Usage:  python tools/gen_golden.py [--only G1,G5,...]
Fixture ids follow SURVEY.md section 8c (G1..G12) plus GH (host helpers).
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    for name in ("open3d", "pymeshlab", "gurobipy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.symeig = lambda a, eigenvectors=True: torch.linalg.eigh(a)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import field_utils  # noqa
    import util  # noqa
    import inference_utils_shim  # noqa  (see below)
    return field_utils, util


def load_cloud(util, name):
    return util.xyz2tensor(open(f"{REF}/data/{name}.xyz").read())


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def scramble_signs(pc, seed=0):
    """Deterministic sign scrambling of the normals."""
    g = torch.Generator().manual_seed(seed)
    flip = torch.rand(pc.shape[0], generator=g) < 0.5
    out = pc.clone()
    out[flip, 3:] *= -1
    return out, flip


def ref_patches(util, pc, n_part, min_patch):
    """The evidently intended divide_pc: grid partition followed by the merge
    (util.divide_pc as committed omits the merge and returns list-wrapped tensors)."""
    indices, ijk = util._divide_pc(pc[:, :3], n_part, min_patch=min_patch)
    merged, _ = util.merge_nodes(pc, indices, ijk, min_patch)
    return merged


# ----------------------------------------------------------------------------------------
def g1(fu, util):
    g = torch.Generator().manual_seed(0)
    src = torch.randn(64, 6, generator=g)
    tgt = torch.randn(48, 6, generator=g)
    out = dict(src=src, tgt=tgt)
    for tag, eps in (("e5", 1e-5), ("e6", 1e-6)):
        out[f"E6_{tag}"] = fu.field_grad(src, tgt, eps=eps)
        out[f"E3_{tag}"] = fu.field_grad(src, tgt[:, :3].contiguous(), eps=eps)
        out[f"E64_{tag}"] = fu.field_grad(src.double(), tgt.double(), eps=eps)
        out[f"phi_{tag}"] = fu.potential(src, tgt, eps=eps)
        out[f"phi64_{tag}"] = fu.potential(src.double(), tgt.double(), eps=eps)
    # degenerate shapes
    out["E_S1"] = fu.field_grad(src[:1], tgt)
    out["E_T1"] = fu.field_grad(src, tgt[:1])
    out["E_S0"] = fu.field_grad(src[:0], tgt)
    save("G1_field_grad_small", **out)


def g2(fu, util):
    pc, _ = util.Transform.trans(load_cloud(util, "fandisk"))
    s = pc[:300].clone()
    t = torch.cat([pc[100:200], pc[1000:1100]], dim=0).clone()  # 100 coincident + 100 distinct
    save("G2_zero_distance",
         src=s, tgt=t,
         E_self=fu.field_grad(s, s), E_self64=fu.field_grad(s.double(), s.double()),
         E_part=fu.field_grad(s, t), E_part64=fu.field_grad(s.double(), t.double()))


def g3(fu, util):
    src = torch.tensor([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    tgt = torch.tensor([[0.0, 0.0, 0.5], [0.0, 0.0, -0.25], [0.7, 0.0, 0.0], [0.0, -0.3, 0.0],
                        [0.3, 0.4, 0.0], [0.0, 0.0, 0.0], [0.1, 0.2, 0.3]])
    out = dict(src=src, tgt=tgt)
    for tag, eps in (("e5", 1e-5), ("e0", 0.0)):
        out[f"E_{tag}"] = fu.field_grad(src, tgt, eps=eps)
        out[f"phi_{tag}"] = fu.potential(src, tgt, eps=eps)
    neg = src.clone()
    neg[:, 3:] *= -1
    out["E_neg"] = fu.field_grad(neg, tgt)
    save("G3_analytic", **out)


def g4(fu, util):
    out = dict(grid=util.gen_grid())
    for name in ("ok", "fandisk", "hand"):
        pc, _ = util.Transform.trans(load_cloud(util, name))
        phi = fu.potential(pc, util.gen_grid())
        out[f"phi_{name}"] = phi
        out[f"mean_{name}"] = fu.measure_mean_potential(pc)
        out[f"phi64_{name}"] = fu.potential(pc.double(), util.gen_grid().double())
    # a source exactly on a lattice node -> 0/0 = NaN in that column -> zeroed after the sum
    pc, _ = util.Transform.trans(load_cloud(util, "ok"))
    pc = pc[:500].clone()
    pc[7, :3] = util.gen_grid()[345]
    out["node_src"] = pc
    out["node_phi"] = fu.potential(pc, util.gen_grid())
    out["node_mean"] = fu.measure_mean_potential(pc)
    save("G4_potential", **out)


def g5(fu, util):
    raw = load_cloud(util, "fandisk")
    pc, _ = util.Transform.trans(raw)
    g = torch.Generator().manual_seed(5)
    rows = torch.randperm(pc.shape[0], generator=g)[:256].sort()[0]
    t0 = time.time()
    E = fu.field_grad(pc, pc)
    print(f"  fandisk all-pairs reference: {time.time() - t0:.1f}s")
    E64_rows = fu.field_grad(pc.double(), pc[rows].double())
    nrm = E.norm(dim=-1)
    save("G5_fandisk_allpairs",
         raw=raw, pc=pc, rows=rows, E_rows=E[rows], E64_rows=E64_rows,
         sum_norm=nrm.double().sum(), max_norm=nrm.max(), norm_all=nrm,
         sign_all=((E * pc[:, 3:]).sum(-1) > 0))


class _TorchProxy:
    """Forwards to torch but records the python list handed to torch.tensor(...) so the
    per-step interaction lists of the greedy drivers can be captured unmodified."""

    def __init__(self, log):
        self._log = log

    def __getattr__(self, k):
        return getattr(torch, k)

    def tensor(self, data, *a, **k):
        t = torch.tensor(data, *a, **k)
        if isinstance(data, list):
            self._log.append(t.clone())
        return t


def _run_patch_driver(fu, util, which, pc0, patches_arg, all_arg, diffuse, weights):
    calls, inter = [], []
    orig_fg, orig_torch = fu.field_grad, fu.torch

    depth = [0]

    def rec_fg(sources, means, *a, **k):
        # the reference's recursion (field_utils.py:73-94) re-enters through the module global: record only
        # the drivers' own calls
        if depth[0] == 0:
            calls.append((sources.shape[0], means.shape[0], sources[0, :3].clone(), sources[0, 3:].clone()))
        depth[0] += 1
        try:
            return orig_fg(sources, means, *a, **k)
        finally:
            depth[0] -= 1

    fu.field_grad = rec_fg
    fu.torch = _TorchProxy(inter)
    pts = pc0.clone()
    try:
        if which == "patch":
            fu.strongest_field_propagation(pts, patches_arg, all_arg, diffuse=diffuse,
                                           weights=None if weights is None else weights.clone())
        else:
            fu.strongest_field_propagation_reps(pts, patches_arg, diffuse=diffuse,
                                                weights=None if weights is None else weights.clone())
    finally:
        fu.field_grad = orig_fg
        fu.torch = orig_torch
    return pts, calls, inter


def _order_from_calls(calls, pc0, patch_first_pts, min_pts=None):
    """Map each recorded field_grad call to the patch whose first point is sources[0]."""
    order, flipped = [], []
    for (S, T, x0, n0) in calls:
        # later calls pass pts[patch] (patch order); the first call passes pts[mask] (point order)
        hit = [k for k, (idx0, xyz) in enumerate(patch_first_pts) if torch.equal(xyz, x0)]
        if len(hit) != 1 and min_pts is not None:
            hit = [k for k, (idx0, xyz) in enumerate(min_pts) if torch.equal(xyz, x0)]
            if len(hit) == 1:
                order.append(hit[0])
                flipped.append(False)
                continue
        if len(hit) != 1:
            order.append(-1)
            flipped.append(False)
            continue
        k = hit[0]
        order.append(k)
        idx0 = patch_first_pts[k][0]
        flipped.append(bool((n0 * pc0[idx0, 3:]).sum() < 0))
    return np.array(order), np.array(flipped)


def g6(fu, util):
    pc, _ = util.Transform.trans(load_cloud(util, "fandisk"))
    allp = ref_patches(util, pc, 30, 100)
    print(f"  fandisk patches: {len(allp)}")
    pc0, flip = scramble_signs(pc, 0)
    # orient patches consistently inside themselves first (as orient_center does in the callers),
    # then scramble whole patches so that the propagation has real work to do
    g = torch.Generator().manual_seed(6)
    pc_patch = pc.clone()
    pflip = torch.rand(len(allp), generator=g) < 0.5
    for k, idx in enumerate(allp):
        if pflip[k]:
            pc_patch[idx, 3:] *= -1
    wts = torch.rand(pc.shape[0], generator=g)
    patches = [(i, p) for i, p in enumerate(allp) if i % 5 != 3]   # a filtered subset for the diffuse pass
    out = dict(pc_patchflip=pc_patch, pc_scrambled=pc0, weights=wts,
               patch_off=np.cumsum([0] + [len(p) for p in allp]),
               patch_idx=torch.cat(allp), filtered=np.array([i for i, _ in patches]),
               curv=np.array([util.pca_eigen_values(pc[p])[0].item() for p in allp]))
    firsts = None
    for cname, cloud in (("pf", pc_patch), ("sc", pc0)):
        for diffuse in (False, True):
            for wname, w in (("nw", None), ("w", wts)):
                tag = f"{cname}_{'d' if diffuse else 'n'}_{wname}"
                t0 = time.time()
                pts, calls, inter = _run_patch_driver(fu, util, "patch", cloud, patches, allp, diffuse, w)
                base = cloud if w is None else torch.cat([cloud[:, :3], cloud[:, 3:] * w.clamp(0.1, 1)[:, None]], 1)
                firsts = [(int(p[0]), base[int(p[0]), :3]) for p in allp]
                mins = [(int(p.min()), base[int(p.min()), :3]) for p in allp]
                order, flipped = _order_from_calls(calls, base, firsts, mins)
                chosen = [float(inter[i][inter[i].abs().argmax()]) for i in range(len(inter))]
                out[f"order_{tag}"] = order
                out[f"flipped_{tag}"] = flipped
                out[f"chosen_{tag}"] = np.array(chosen)
                out[f"sign_{tag}"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
                out[f"normals_{tag}"] = pts[:, 3:]
                print(f"  G6 {tag}: {time.time() - t0:.1f}s, start patch {order[0]}")
    save("G6_patch_propagation", **out)


def g7(fu, util):
    pc, _ = util.Transform.trans(load_cloud(util, "fandisk"))
    allp = ref_patches(util, pc, 30, 100)
    g = torch.Generator().manual_seed(6)
    pc_patch = pc.clone()
    pflip = torch.rand(len(allp), generator=g) < 0.5
    for k, idx in enumerate(allp):
        if pflip[k]:
            pc_patch[idx, 3:] *= -1
    out = dict(pc_patchflip=pc_patch, patch_off=np.cumsum([0] + [len(p) for p in allp]),
               patch_idx=torch.cat(allp))
    for cap in (500, 50):
        torch.manual_seed(1)
        reps = []
        for p in allp:
            perm = torch.randperm(p.shape[0])
            reps.append((p[perm[:cap]], p[perm[cap:]]))
        out[f"rep_off_{cap}"] = np.cumsum([0] + [len(r) for r, _ in reps])
        out[f"rep_idx_{cap}"] = torch.cat([r for r, _ in reps])
        out[f"rest_off_{cap}"] = np.cumsum([0] + [len(r) for _, r in reps])
        out[f"rest_idx_{cap}"] = torch.cat([r for _, r in reps])
        for diffuse in (False, True):
            tag = f"{cap}_{'d' if diffuse else 'n'}"
            t0 = time.time()
            pts, calls, inter = _run_patch_driver(fu, util, "reps", pc_patch, reps, None, diffuse, None)
            firsts = [(int(r[0]), pc_patch[int(r[0]), :3]) for r, _ in reps]
            mins = [(int(r.min()), pc_patch[int(r.min()), :3]) for r, _ in reps]
            order, flipped = _order_from_calls(calls[:len(allp)], pc_patch, firsts, mins)
            out[f"order_{tag}"] = order
            out[f"flipped_{tag}"] = flipped
            out[f"chosen_{tag}"] = np.array([float(t[t.abs().argmax()]) for t in inter])
            out[f"sign_{tag}"] = ((pts[:, 3:] * pc_patch[:, 3:]).sum(-1) > 0)
            print(f"  G7 {tag}: {time.time() - t0:.1f}s")
    save("G7_reps_propagation", **out)


def g8(fu, util):
    raw = load_cloud(util, "ok")
    pc, _ = util.Transform.trans(raw)
    out = dict(raw=raw)
    g = torch.Generator().manual_seed(8)
    sub = torch.randperm(pc.shape[0], generator=g)[:1000].sort()[0]
    for name, cloud in (("sub1000", pc[sub].clone()), ("full", pc)):
        cloud0, _ = scramble_signs(cloud, 0)
        for diffuse in (False, True):
            order = []
            orig_fg = fu.field_grad

            def rec_fg(sources, means, *a, **k):
                order.append(sources.storage_offset() // 6)
                return orig_fg(sources, means, *a, **k)

            fu.field_grad = rec_fg
            t0 = time.time()
            try:
                pts = fu.strongest_field_propagation_points(cloud0.clone(), diffuse=diffuse, starting_point=0)
            finally:
                fu.field_grad = orig_fg
            tag = f"{name}_{'d' if diffuse else 'n'}"
            out[f"order_{tag}"] = np.array(order)
            out[f"sign_{tag}"] = ((pts[:, 3:] * cloud0[:, 3:]).sum(-1) > 0)
            print(f"  G8 {tag}: {time.time() - t0:.1f}s")
        out[f"pc_{name}"] = cloud0
    out["sub_rows"] = sub
    save("G8_point_propagation", **out)


def g9(fu, util):
    src, _ = util.Transform.trans(load_cloud(util, "ok"))
    g = torch.Generator().manual_seed(3)
    tgt3 = (src[:, :3] + 1e-3 * torch.randn(src.shape[0], 3, generator=g)).contiguous()
    tgt6, _ = scramble_signs(torch.cat([tgt3, src[:, 3:]], dim=1), 0)
    out3 = fu.reference_field(src, tgt3.clone())
    t6 = tgt6.clone()
    out6 = fu.reference_field(src, t6)
    save("G9_reference_field", src=src, tgt3=tgt3, tgt6=tgt6, out3=out3, out6=out6,
         E64=fu.field_grad(src.double(), tgt3.double()))


def g10(fu, util):
    pc, _ = util.Transform.trans(load_cloud(util, "fandisk"))
    allp = ref_patches(util, pc, 30, 100)
    a, b = pc[allp[3]], pc[allp[4]]
    w, invw = fu.field_edge_calculator(a, b)
    wb = fu.field_edge_calculator_bool(a, b)
    wc = fu.field_edge_calculator_count(a, b)
    ws = fu.self_interaction_all(a)
    save("G10_edge", a=a, b=b, w=w, invw=invw, wbool=np.array(wb), wcount=np.array(wc), wself=ws)


def g11(fu, util):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16000, 3, generator=g)
    x = x / x.norm(dim=-1, keepdim=True) * 0.5
    n = torch.randn(16000, 3, generator=g)
    n = n / n.norm(dim=-1, keepdim=True)
    pc = torch.cat([x, n], dim=1)
    rows = torch.cat([torch.arange(64), torch.arange(8000 - 32, 8000 + 32), torch.arange(16000 - 64, 16000)])
    t0 = time.time()
    E = fu.field_grad(pc, pc)
    print(f"  16000^2 reference with recursion: {time.time() - t0:.1f}s")
    save("G11_recursion", seed=11, rows=rows, E_rows=E[rows],
         E64_rows=fu.field_grad(pc.double(), pc[rows].double()),
         pc_head=pc[:8], pc_tail=pc[-8:])


def g12(fu, util):
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(100000, 3, generator=g)
    n = x / x.norm(dim=-1, keepdim=True)
    pc = torch.cat([n, n], dim=1)
    pc, tr = util.Transform.trans(pc)
    rows = torch.arange(0, 100000, 100000 // 64)[:64]
    t0 = time.time()
    E_rows = fu.field_grad(pc, pc[rows])
    print(f"  100k sphere, 64 target rows: {time.time() - t0:.1f}s")
    save("G12_sphere100k", head=pc[:16], center=tr.center, scale=tr.scale,
         rows=rows, E_rows=E_rows, E64_rows=fu.field_grad(pc.double(), pc[rows].double()),
         mean_potential=fu.measure_mean_potential(pc))


def gh(fu, util):
    """Host helpers on the path."""
    raw = load_cloud(util, "fandisk")
    pc, tr = util.Transform.trans(raw)
    allp = ref_patches(util, pc, 30, 100)
    ev = [util.pca_eigen_values(pc[p]) for p in allp]
    oc = util.orient_center(pc[allp[0]].clone())
    inv = tr.inverse(pc)
    # fix_n_filter (inference_utils.py:52-71) on a copy, threshold 0.01 so that some patches are filtered
    import inference_utils_shim as iu
    pcf = pc.clone()
    g = torch.Generator().manual_seed(2)
    fl = torch.rand(pcf.shape[0], generator=g) < 0.5
    pcf[fl, 3:] *= -1
    pcf_in = pcf.clone()
    kept = iu.fix_n_filter(pcf, [p.clone() for p in allp], 0.01)
    # small grid partition without merge for a tiny cloud
    small = pc[::37].clone()
    ind_s, ijk_s = util._divide_pc(small[:, :3], 6)
    save("GH_host_helpers", raw_head=raw[:32], center=tr.center, scale=tr.scale, pc_head=pc[:32],
         inv_head=inv[:32], patch_off=np.cumsum([0] + [len(p) for p in allp]), patch_idx=torch.cat(allp),
         eig_min=np.array([e[0].item() for e, _ in ev]), eig_vec=torch.stack([v for _, v in ev]),
         oc_in=pc[allp[0]], oc_out=oc, filt_in=pcf_in, filt_out=pcf, filt_kept=np.array([i for i, _ in kept]),
         small=small, small_off=np.cumsum([0] + [len(i[0]) for i in ind_s]),
         small_idx=torch.cat([i[0] for i in ind_s]), small_ijk=torch.stack([j[0] for j in ijk_s]))
    txt = "1 2 3\n4 5 6 0 0 1\nnan 1 2\n\n7.5 8 9e-1"
    save("GH_xyz_parse", parsed=util.xyz2tensor(txt), parsed_noappend=util.xyz2tensor("1 2 3\n4 5 6", append_normals=False))


def gx(fu, util):
    """The fork's 'xie' pair functions that are pure torch given an order:
    xie_field / xie_intersaction / xie_distance (field_utils.py:431-519) and
    xie_propagation_points_in_order (field_utils.py:569-605)."""
    g = torch.Generator().manual_seed(31)
    src = torch.randn(50, 6, generator=g)
    tgt = torch.randn(40, 6, generator=g)
    tgt[:5, :3] = src[10:15, :3]                      # coincident pairs: not divided by |R|^3
    out = dict(src=src, tgt=tgt)
    for C in (3, 2):
        out[f"field_C{C}"] = fu.xie_field(src, tgt, eps=0.1, C=C)
        out[f"inter_C{C}"] = fu.xie_intersaction(src, tgt, eps=0.1, knn_mask=-1, C=C)
    out["inter_knn5"] = fu.xie_intersaction(src, tgt, eps=0.1, knn_mask=5, C=3)
    out["inter64"] = fu.xie_intersaction(src.double(), tgt.double(), eps=0.1, knn_mask=-1, C=3)
    out["distance"] = fu.xie_distance(src, tgt, eps=0.1)
    # ordered propagation on the 1000-point ok.xyz subsample, three visiting orders
    pc = torch.from_numpy(np.load(os.path.join(OUT, "G8_point_propagation.npz"))["pc_sub1000"])
    orders = np.stack([np.arange(1000), torch.randperm(1000, generator=g).numpy(), np.arange(1000)[::-1].copy()])
    out["pc"] = pc
    out["orders"] = orders
    for diffuse in (False, True):
        for knn in (-1, 20):
            t0 = time.time()
            res = fu.xie_propagation_points_in_order(pc.clone(), 0.1, orders, diffuse=diffuse, knn_mask=knn, C=3)
            out[f"flip_{'d' if diffuse else 'n'}_k{knn if knn > 0 else 0}"] = res
            print(f"  GX in_order diffuse={diffuse} knn={knn}: {time.time() - t0:.1f}s, flipped {int(res.sum())}")
    out["inter_pc"] = fu.xie_intersaction(pc, pc, eps=0.1, knn_mask=-1, C=3)[:64]
    save("GX_xie", **out)


def g13(fu, util):
    """A second cloud for the greedy drivers: hand.xyz (10 000 points) - full per-point visit order, and the
    patch driver at n_part 20 / min 50 (different partition than G6), both with scrambled signs."""
    raw = load_cloud(util, "hand")
    pc, _ = util.Transform.trans(raw)
    cloud0, _ = scramble_signs(pc, 13)
    out = dict(raw=raw, pc_scrambled=cloud0)
    order = []
    orig_fg = fu.field_grad

    def rec_fg(sources, means, *a, **k):
        order.append(sources.storage_offset() // 6)
        return orig_fg(sources, means, *a, **k)

    fu.field_grad = rec_fg
    t0 = time.time()
    try:
        pts = fu.strongest_field_propagation_points(cloud0.clone(), diffuse=True, starting_point=17)
    finally:
        fu.field_grad = orig_fg
    out["order_points"] = np.array(order)
    out["sign_points"] = ((pts[:, 3:] * cloud0[:, 3:]).sum(-1) > 0)
    print(f"  G13 per-point: {time.time() - t0:.1f}s")
    allp = ref_patches(util, pc, 20, 50)
    print(f"  hand patches: {len(allp)}")
    g = torch.Generator().manual_seed(14)
    pflip = torch.rand(len(allp), generator=g) < 0.5
    pc_patch = pc.clone()
    for k, idx in enumerate(allp):
        if pflip[k]:
            pc_patch[idx, 3:] *= -1
    patches = list(enumerate(allp))
    ptsp, calls, inter = _run_patch_driver(fu, util, "patch", pc_patch, patches, allp, True, None)
    firsts = [(int(p[0]), pc_patch[int(p[0]), :3]) for p in allp]
    mins = [(int(p.min()), pc_patch[int(p.min()), :3]) for p in allp]
    order_p, flipped_p = _order_from_calls(calls, pc_patch, firsts, mins)
    out.update(pc_patchflip=pc_patch, patch_off=np.cumsum([0] + [len(p) for p in allp]), patch_idx=torch.cat(allp),
               order_patch=order_p, flipped_patch=flipped_p,
               chosen_patch=np.array([float(t[t.abs().argmax()]) for t in inter]),
               sign_patch=((ptsp[:, 3:] * pc_patch[:, 3:]).sum(-1) > 0))
    save("G13_hand", **out)


def g14(fu, util):
    """Per-point propagation on a float64 cloud (the socket path, util.py:71-77 -> field_utils.py:353-388 in
    fp64): 1000-point ok.xyz subsample and a 3000-point one, visit order + signs."""
    base = np.load(os.path.join(OUT, "G8_point_propagation.npz"))
    out = {}
    for name, cloud in (("sub1000", torch.from_numpy(base["pc_sub1000"]).double()),
                        ("sub3000", torch.from_numpy(base["pc_full"])[::3][:3000].clone().double())):
        for diffuse in (False, True):
            order = []
            orig_fg = fu.field_grad

            def rec_fg(sources, means, *a, **k):
                order.append(sources.storage_offset() // 6)
                return orig_fg(sources, means, *a, **k)

            fu.field_grad = rec_fg
            try:
                pts = fu.strongest_field_propagation_points(cloud.clone(), diffuse=diffuse, starting_point=0)
            finally:
                fu.field_grad = orig_fg
            assert pts.dtype == torch.float64
            tag = f"{name}_{'d' if diffuse else 'n'}"
            out[f"order_{tag}"] = np.array(order)
            out[f"sign_{tag}"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
        out[f"pc_{name}"] = cloud
    save("G14_point_propagation_f64", **out)


def pca_normals_numpy(xyz, k):
    """Unoriented PCA normals (own numpy routine, NOT the reference's open3d estimator, which is absent
    offline): eigenvector of the smallest eigenvalue of the covariance of the k nearest neighbours
    (scipy cKDTree, the point itself included), fp64, sign as numpy.linalg.eigh returns it."""
    from scipy.spatial import cKDTree
    x = np.asarray(xyz, dtype=np.float64)
    _, nn = cKDTree(x).query(x, k=k)
    nb = x[nn]                                              # [N, k, 3]
    rel = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", rel, rel) / k
    _, v = np.linalg.eigh(cov)
    return v[:, :, 0]


def g15(fu, util):
    """BASELINE config 3 on a cloud the reference holds (data/boxunion.xyz stands in for the
    missing lion.xyz) with the flags of demos/lion.sh:13-19: the stages of orient_large.py:18-75 run with the
    reference's own functions - Transform, _divide_pc + merge_nodes (n_part 41, min 100), fix_n_filter
    (curvature_threshold 0.0), orient_center, torch.manual_seed(1) randperm representatives (cap 500),
    strongest_field_propagation_reps(diffuse=True), measure_mean_potential.  Normals come from
    pca_normals_numpy (k = 50, lion.sh --n 50); the PointCNN vote is skipped (models absent offline)."""
    import inference_utils_shim as iu
    raw = util.xyz2tensor(open(f"{REF}/data/boxunion.xyz").read(), append_normals=False)
    pc3, tr = util.Transform.trans(raw)
    t0 = time.time()
    nrm = torch.from_numpy(pca_normals_numpy(pc3.numpy(), 50)).float()
    pc = torch.cat([pc3, nrm], dim=1).contiguous()
    print(f"  normals: {time.time() - t0:.1f}s")
    t0 = time.time()
    allp = ref_patches(util, pc, 41, 100)
    print(f"  reference partition + merge: {len(allp)} patches, {time.time() - t0:.1f}s")
    pc_in = pc.clone()
    kept = iu.fix_n_filter(pc_in, [p.clone() for p in allp], 0.0)
    for _, p in kept:
        pc_in[p] = util.orient_center(pc_in[p])
    print(f"  kept {len(kept)}/{len(allp)}")
    torch.manual_seed(1)
    reps = []
    for p in allp:
        perm = torch.randperm(p.shape[0])
        reps.append((p[perm[:500]], p[perm[500:]]))
    t0 = time.time()
    pts, calls, inter = _run_patch_driver(fu, util, "reps", pc_in, reps, None, True, None)
    print(f"  reference reps propagation: {time.time() - t0:.1f}s")
    firsts = [(int(r[0]), pc_in[int(r[0]), :3]) for r, _ in reps]
    mins = [(int(r.min()), pc_in[int(r.min()), :3]) for r, _ in reps]
    order, flipped = _order_from_calls(calls[:len(allp)], pc_in, firsts, mins)
    mean_phi = fu.measure_mean_potential(pts)
    i32 = lambda t: torch.as_tensor(t).to(torch.int32)
    save("G15_boxunion_config3",
         pc=pc, center=tr.center, scale=tr.scale,
         patch_off=np.cumsum([0] + [len(p) for p in allp]), patch_idx=i32(torch.cat(allp)),
         kept=np.array([i for i, _ in kept]),
         prefilter_sign=((pc_in[:, 3:] * pc[:, 3:]).sum(-1) > 0),          # cloud handed to the propagation
         rep_off=np.cumsum([0] + [len(r) for r, _ in reps]), rep_idx=i32(torch.cat([r for r, _ in reps])),
         rest_off=np.cumsum([0] + [len(r) for _, r in reps]), rest_idx=i32(torch.cat([r for _, r in reps])),
         curv=np.array([util.pca_eigen_values(pc_in[r])[0].item() for r, _ in reps]),
         order=order, flipped=flipped, chosen=np.array([float(t[t.abs().argmax()]) for t in inter]),
         sign=((pts[:, 3:] * pc_in[:, 3:]).sum(-1) > 0), mean_potential=mean_phi)


def g16(fu, util):
    """Per-point propagation of the FULL 10 000-point ok.xyz as a float64 cloud (the socket path,
    util.py:71-77 -> field_utils.py:353-388 in fp64): complete visit order + final signs."""
    base = np.load(os.path.join(OUT, "G8_point_propagation.npz"))
    cloud = torch.from_numpy(base["pc_full"]).double()
    out = {}
    for diffuse in (True,):
        order = []
        orig_fg = fu.field_grad

        def rec_fg(sources, means, *a, **k):
            order.append(sources.storage_offset() // 6)
            return orig_fg(sources, means, *a, **k)

        fu.field_grad = rec_fg
        t0 = time.time()
        try:
            pts = fu.strongest_field_propagation_points(cloud.clone(), diffuse=diffuse, starting_point=0)
        finally:
            fu.field_grad = orig_fg
        assert pts.dtype == torch.float64
        out["order_full_d"] = np.array(order).astype(np.int32)
        out["sign_full_d"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
        print(f"  G16 full fp64 per-point: {time.time() - t0:.1f}s")
    save("G16_point_propagation_f64_full", **out)      # the cloud itself is G8's pc_full cast to float64


def gw(fu, util):
    """Byte-level vectors of the socket wire format (socket_server_para.py:137-195): the reference's own
    handle_client is driven through a fake connection object; the estimator it would call is replaced by a
    stub that records the decoded request and returns a fixed array, so the fixture pins the framing (header
    parse, ack, payload decode, reply encode, the ERROR reply), not an estimator."""
    import json
    import tempfile
    sys.modules.setdefault("graph_dipole", types.ModuleType("graph_dipole"))
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())                 # log_msg appends to ./error.log
    try:
        import socket_server_para as srv
    finally:
        pass

    class FakeConn:
        def __init__(self, chunks):
            self.chunks, self.sent = list(chunks), []

        def recv(self, n):
            if not self.chunks:
                return b""
            c = self.chunks[0]
            out, rest = c[:n], c[n:]
            if rest:
                self.chunks[0] = rest
            else:
                self.chunks.pop(0)
            return out

        def sendall(self, b):
            self.sent.append(bytes(b))

        def close(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    g = np.random.default_rng(5)
    xyz = g.standard_normal((7, 3))
    result = np.concatenate([xyz.astype(np.float32), g.standard_normal((7, 3)).astype(np.float32)], axis=1)
    seen = {}

    def stub(xyz_data, config):
        seen["xyz"], seen["config"] = np.array(xyz_data), dict(config)
        return result

    srv.simple_estimate = stub
    header = json.dumps({"function_name": "simple_estimate", "function_config": {"diffuse": True}, "data_size": 7}).encode()
    payload = xyz.astype(np.float64).tobytes()
    ok = FakeConn([header, payload[:50], payload[50:]])        # payload arriving in two pieces
    srv.handle_client(ok, ("test", 0))
    bad_fn = FakeConn([json.dumps({"function_name": "nope", "function_config": {}, "data_size": 7}).encode(), payload])
    srv.handle_client(bad_fn, ("test", 0))
    short = FakeConn([header, payload[:100]])                      # connection drops early: size mismatch
    srv.handle_client(short, ("test", 0))
    os.chdir(cwd)
    u8 = lambda b: np.frombuffer(b, dtype=np.uint8)
    assert np.array_equal(seen["xyz"], xyz) and seen["config"] == {"diffuse": True}
    save("GW_wire_format", header=u8(header), payload=u8(payload), xyz=xyz, result=result,
         ack=u8(ok.sent[0]), reply=u8(ok.sent[1]), n_sent_ok=len(ok.sent),
         unknown_ack=u8(bad_fn.sent[0]), unknown_reply=u8(bad_fn.sent[1]),
         short_ack=u8(short.sent[0]), short_reply=u8(short.sent[1]), decoded_xyz=seen["xyz"])


def gx2(fu, util):
    """BFS-route propagation with a vote (field_utils.xie_propagation_points_onbfstree, field_utils.py:657-710)
    on the 1000-point ok.xyz subsample, times = 1 and 5.  The reference's own graph.getEMSTfromPC /
    LinkedListGraph.get_bfs_route build the routes; its MIQP (gurobi, absent offline) is replaced IN THE HARNESS by
    an exhaustive search over the 2^times assignments of the same objective (cal_loss, field_utils.py:606-617),
    with x[0] = 0 choosing between an optimum and its complement (they tie by construction).  Captured: the
    starting points, the routes, the per-route flip vectors, the aligned vote and the final flip vector."""
    import itertools
    pc = torch.from_numpy(np.load(os.path.join(OUT, "G8_point_propagation.npz"))["pc_sub1000"])
    seen = {}
    orig_in_order, orig_miqp = fu.xie_propagation_points_in_order, fu.MIQP

    def rec_in_order(pts, eps, order, *a, **k):
        seen["orders"] = np.array(order)
        res = orig_in_order(pts, eps, order, *a, **k)
        seen["flips"] = res.clone()
        return res

    def exhaustive(A, B):
        n = len(A)
        best, best_cost = None, None
        for tail in itertools.product((0, 1), repeat=n - 1):
            x = np.array((0,) + tail[::-1])                      # binary order with x[0] fixed to 0
            cost = fu.cal_loss(x, A, B)
            if best_cost is None or cost < best_cost:
                best, best_cost = x, cost
        seen["status"] = best.astype(bool)
        return best.astype(float)

    fu.xie_propagation_points_in_order, fu.MIQP = rec_in_order, exhaustive
    out = dict(pc=pc)
    try:
        for times, diffuse in ((1, False), (5, False), (5, True)):
            pts = pc.clone()
            res = fu.xie_propagation_points_onbfstree(pts, 0.1, diffuse=diffuse, starting_point=0, k=10, treshold=0.1,
                                                      times=times, knn_mask=-1, C=3)
            tag = f"t{times}_{'d' if diffuse else 'n'}"
            out[f"orders_{tag}"] = seen["orders"]
            out[f"flips_{tag}"] = seen["flips"]
            out[f"status_{tag}"] = seen["status"]
            out[f"result_{tag}"] = res
            out[f"normals_{tag}"] = pts[:, 3:]
            print(f"  GX2 {tag}: flipped {int(res.sum())}, status {seen['status'].astype(int)}")
    finally:
        fu.xie_propagation_points_in_order, fu.MIQP = orig_in_order, orig_miqp
    save("GX2_xie_bfstree", **out)


def g17(fu, util):
    """The patch driver at the headline size on a reference-held cloud: strongest_field_propagation
    (field_utils.py:286-348) on the G15 boxunion cloud (100 000 points, the reference's 369 patches, its
    fix_n_filter / orient_center state and kept list), diffuse, no weights - 10^10 pair evaluations in the
    reference.  Inputs are G15's; only the trace and the signs are stored here."""
    g = np.load(os.path.join(OUT, "G15_boxunion_config3.npz"))
    pc = torch.from_numpy(g["pc"]).clone()
    pc[~torch.from_numpy(g["prefilter_sign"]), 3:] *= -1
    off, idx = g["patch_off"], torch.from_numpy(g["patch_idx"].astype(np.int64))
    allp = [idx[off[k]:off[k + 1]].clone() for k in range(len(off) - 1)]
    patches = [(int(i), allp[int(i)]) for i in g["kept"]]
    t0 = time.time()
    pts, calls, inter = _run_patch_driver(fu, util, "patch", pc, patches, allp, True, None)
    print(f"  reference patch propagation at 100k: {time.time() - t0:.1f}s")
    firsts = [(int(p[0]), pc[int(p[0]), :3]) for p in allp]
    mins = [(int(p.min()), pc[int(p.min()), :3]) for p in allp]
    order, flipped = _order_from_calls(calls, pc, firsts, mins)
    save("G17_boxunion_patch_propagation", order=order, flipped=flipped,
         chosen=np.array([float(t[t.abs().argmax()]) for t in inter]),
         sign=((pts[:, 3:] * pc[:, 3:]).sum(-1) > 0),
         curv=np.array([util.pca_eigen_values(pc[p])[0].item() for p in allp]))


def g18(fu, util):
    """BASELINE config 5 at its headline size on a reference-held cloud: reference_field (field_utils.py:188-201)
    with S = T = 100 000 - the G15 boxunion cloud as the oriented reference, the same cloud jittered by 1e-3
    (seed 5) as the cloud to orient, normals sign-scrambled (seed 5).  10^10 pair evaluations.  The reference's
    field_grad splits only the SOURCES into leaves, so one call would hold [100 000, 12 500, 3] temporaries
    (15 GB each); the harness feeds the targets in blocks of 4000 rows - every target row is an independent sum,
    so this is the call the reference makes, row block by row block.  Stored: the sign decision of all 100 000
    points (6-column form), E.n of every point (the margin of that decision), the field and the 3-column-form
    normals on 4000 sampled rows.  Inputs are rebuilt from G15 and the seed."""
    g = np.load(os.path.join(OUT, "G15_boxunion_config3.npz"))
    src = torch.from_numpy(g["pc"]).clone()
    gen = torch.Generator().manual_seed(5)
    tgt3 = (src[:, :3] + 1e-3 * torch.randn(src.shape[0], 3, generator=gen)).contiguous()
    tgt6, flip = scramble_signs(torch.cat([tgt3, src[:, 3:]], dim=1), 5)
    N = src.shape[0]
    rows = torch.sort(torch.randperm(N, generator=gen)[:4000]).values
    t0 = time.time()
    out6 = torch.empty_like(tgt6)
    E = torch.empty(N, 3)
    for lo in range(0, N, 4000):
        blk = tgt6[lo:lo + 4000].clone()
        E[lo:lo + 4000] = fu.field_grad(src, blk)                      # the field reference_field computes first
        out6[lo:lo + 4000] = fu.reference_field(src, blk)             # in place on the block, returned
        if (lo // 4000) % 5 == 0:
            print(f"    rows {lo}: {time.time() - t0:.0f}s", flush=True)
    out3_rows = fu.reference_field(src, tgt3[rows].clone())
    print(f"  reference_field at 100k x 100k: {time.time() - t0:.1f}s")
    keep = (out6[:, 3:] * tgt6[:, 3:]).sum(-1) > 0                     # sign = +1 rows
    save("G18_reference_field_100k", seed=5, tgt3_head=tgt3[:8], tgt3_sum=tgt3.double().sum(0), flip_count=int(flip.sum()),
         keep=np.packbits(keep.numpy()), e_dot_n=(E * tgt6[:, 3:]).sum(-1), rows=rows, E_rows=E[rows],
         out3_rows=out3_rows)


def g19(fu, util):
    """The HEADLINE workload itself (BASELINE config 4 = bench.py's cloud, tools/workloads.headline_workload): the
    100 000-point sphere (seed 1234), 256 Fibonacci patches, whole patches sign-scrambled (seed 0), run through the
    reference's strongest_field_propagation (field_utils.py:286-348) with diffuse=True and every patch in the
    filtered list - 10^10 pair evaluations in the reference.  Stored: its start patch, the complete visit order, the
    flip decisions, the chosen interaction of every step, the 100 000 final signs, the per-patch curvatures it chose
    the start from, and - as the pin of the bench's dominant kernel - the reference's own
    field_grad(pts[patch_k], pts[~patch_k]) for three patches (all ~99 600 rows each).  The inputs are rebuilt from
    the seeds by tools/workloads.py; a checksum of them is stored."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.workloads import headline_workload
    pc, allp, scramble = headline_workload()
    patches = list(enumerate(allp))
    out = dict(pc_head=pc[:16], pc_sum=pc.double().sum(0), scramble=scramble,
               sizes=np.array([len(p) for p in allp]))
    slab_patches = np.array([0, 100, 255])
    for k in slab_patches:
        mask = torch.zeros(len(pc), dtype=torch.bool)
        mask[allp[k]] = True
        out[f"dE_{k}"] = fu.field_grad(pc[allp[k]], pc[~mask])
    out["slab_patches"] = slab_patches
    out["curv"] = np.array([util.pca_eigen_values(pc[p])[0].item() for p in allp])
    t0 = time.time()
    pts, calls, inter = _run_patch_driver(fu, util, "patch", pc, patches, allp, True, None)
    print(f"  reference patch propagation on the headline cloud: {time.time() - t0:.1f}s")
    firsts = [(int(p[0]), pc[int(p[0]), :3]) for p in allp]
    mins = [(int(p.min()), pc[int(p.min()), :3]) for p in allp]
    order, flipped = _order_from_calls(calls, pc, firsts, mins)
    assert (np.sort(order) == np.arange(len(allp))).all(), "every patch visited exactly once"
    out.update(order=order, flipped=flipped, chosen=np.array([float(t[t.abs().argmax()]) for t in inter]),
               sign=np.packbits(((pts[:, 3:] * pc[:, 3:]).sum(-1) > 0).numpy()))
    save("G19_headline_sphere_patch_propagation", **out)


def g20(fu, util):
    """The caller pipelines on data/fandisk.xyz, every stage run by the REFERENCE's own functions (the network vote is
    skipped: models absent offline) - the fixtures tests/test_callers_gpu.py compares the caller counterparts with,
    so that no stage of the expected result comes from the product's own helpers:
      pointcloud: orient_pointcloud.py:14-76 with demos/fandisk.sh's flags (number_parts 30, minimum_points_per_patch
        100, diffuse): file normals sign-scrambled (seed 3), Transform, _divide_pc + merge_nodes, fix_n_filter (0.0),
        orient_center, strongest_field_propagation(diffuse), measure_mean_potential flip;
      large: orient_large.py:18-75 on the unscrambled file: the same prologue, torch.manual_seed(1) randperm
        representatives (cap 500), strongest_field_propagation_reps(diffuse=True), global flip.
    Stored: the scramble, the state handed to the propagation (as signs against the input), kept patch ids, start patch,
    the final normals' signs against the input, the mean potential before the global flip."""
    import inference_utils_shim as iu
    raw = load_cloud(util, "fandisk")
    out = {}
    for name in ("pointcloud", "large"):
        cloud = raw.clone()
        if name == "pointcloud":
            flip = torch.rand(raw.shape[0], generator=torch.Generator().manual_seed(3)) < 0.5
            cloud[flip, 3:] *= -1
            out["scramble"] = flip
        pc, tr = util.Transform.trans(cloud)
        allp = ref_patches(util, pc, 30, 100)
        pc_in = pc.clone()
        kept = iu.fix_n_filter(pc_in, [p.clone() for p in allp], 0.0)
        for _, p in kept:
            pc_in[p] = util.orient_center(pc_in[p])
        state = pc_in.clone()
        if name == "pointcloud":
            pts, calls, inter = _run_patch_driver(fu, util, "patch", pc_in, kept, [p.clone() for p in allp], True, None)
            firsts = [(int(p[0]), state[int(p[0]), :3]) for p in allp]
            mins = [(int(p.min()), state[int(p.min()), :3]) for p in allp]
            order, _ = _order_from_calls(calls, state, firsts, mins)
        else:
            torch.manual_seed(1)
            reps = []
            for p in allp:
                perm = torch.randperm(p.shape[0])
                reps.append((p[perm[:500]], p[perm[500:]]))
            pts, calls, inter = _run_patch_driver(fu, util, "reps", pc_in, reps, None, True, None)
            firsts = [(int(r[0]), state[int(r[0]), :3]) for r, _ in reps]
            mins = [(int(r.min()), state[int(r.min()), :3]) for r, _ in reps]
            order, _ = _order_from_calls(calls[:len(allp)], state, firsts, mins)
            out["large_rep_off"] = np.cumsum([0] + [len(r) for r, _ in reps])
            out["large_rep_idx"] = torch.cat([r for r, _ in reps]).to(torch.int32)
        phi = fu.measure_mean_potential(pts)
        if phi < 0:
            pts[:, 3:] *= -1
        out[f"{name}_n_patches"] = len(allp)
        out[f"{name}_kept"] = np.array([i for i, _ in kept])
        out[f"{name}_state_sign"] = ((state[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
        out[f"{name}_start"] = int(order[0])
        out[f"{name}_order"] = order
        out[f"{name}_mean_potential"] = phi
        out[f"{name}_final_sign"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
        out[f"{name}_final_xyz_head"] = tr.inverse(pts)[:8]
        print(f"  G20 {name}: {len(kept)}/{len(allp)} patches kept, start {int(order[0])}, mean potential {float(phi):.4e}")
    save("G20_fandisk_caller_pipelines", **out)


def g21(fu, util):
    """Round 5: the drivers on FLOAT64 clouds (the reference computes in the dtype it is handed, field_utils.py:96-109, :286-348,
    :207-282, :569-605; its socket path hands it float64, util.py:71-77).  The clouds, patches, representatives and routes are
    those of G6 / G7 / GX / GX2 cast to float64 - only the reference's float64 OUTPUTS are stored here: order / flipped / chosen /
    signs / normals of two G6 variants and one G7 variant, the ordered xie propagation (GX's three orders, plain and diffuse) and
    the BFS-route vote t5_d."""
    import itertools
    g6 = np.load(os.path.join(OUT, "G6_patch_propagation.npz"))
    off = g6["patch_off"]
    idx = torch.from_numpy(g6["patch_idx"])
    allp = [idx[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    filtered = g6["filtered"]
    patches = [(int(i), allp[int(i)]) for i in filtered]
    out = {}
    for cname, key, diffuse, use_w in (("pf", "pc_patchflip", True, True), ("sc", "pc_scrambled", False, False),
                                       ("pf", "pc_patchflip", True, False)):
        cloud = torch.from_numpy(g6[key]).double()
        w = torch.from_numpy(g6["weights"]).double() if use_w else None
        tag = f"g6_{cname}_{'d' if diffuse else 'n'}_{'w' if use_w else 'nw'}"
        t0 = time.time()
        pts, calls, inter = _run_patch_driver(fu, util, "patch", cloud, patches, allp, diffuse, w)
        assert pts.dtype == torch.float64 and inter[0].dtype == torch.float64
        base = cloud if w is None else torch.cat([cloud[:, :3], cloud[:, 3:] * w.clamp(0.1, 1)[:, None]], 1)
        firsts = [(int(p[0]), base[int(p[0]), :3]) for p in allp]
        mins = [(int(p.min()), base[int(p.min()), :3]) for p in allp]
        order, flipped = _order_from_calls(calls, base, firsts, mins)
        out[f"order_{tag}"] = order
        out[f"flipped_{tag}"] = flipped
        out[f"chosen_{tag}"] = np.array([float(t[t.abs().argmax()]) for t in inter])
        out[f"sign_{tag}"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
        out[f"normals_{tag}"] = pts[:, 3:]
        print(f"  G21 {tag}: {time.time() - t0:.1f}s, start patch {order[0]}")
    # G7, cap 50, diffuse (rests are non-empty), float64
    g7 = np.load(os.path.join(OUT, "G7_reps_propagation.npz"))
    cloud = torch.from_numpy(g7["pc_patchflip"]).double()
    ro, ri = g7["rep_off_50"], torch.from_numpy(g7["rep_idx_50"])
    so, si = g7["rest_off_50"], torch.from_numpy(g7["rest_idx_50"])
    reps = [(ri[ro[k]:ro[k + 1]], si[so[k]:so[k + 1]]) for k in range(len(ro) - 1)]
    t0 = time.time()
    pts, calls, inter = _run_patch_driver(fu, util, "reps", cloud, reps, None, True, None)
    assert pts.dtype == torch.float64
    firsts = [(int(r[0]), cloud[int(r[0]), :3]) for r, _ in reps]
    mins = [(int(r.min()), cloud[int(r.min()), :3]) for r, _ in reps]
    order, flipped = _order_from_calls(calls[:len(reps)], cloud, firsts, mins)
    out["order_g7_50_d"] = order
    out["flipped_g7_50_d"] = flipped
    out["chosen_g7_50_d"] = np.array([float(t[t.abs().argmax()]) for t in inter])
    out["sign_g7_50_d"] = ((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
    out["normals_g7_50_d"] = pts[:, 3:]
    print(f"  G21 g7_50_d: {time.time() - t0:.1f}s, start patch {order[0]}")
    # the ordered xie propagation in float64 (GX's cloud and orders), plain and diffuse; a float64 slice of the matrix
    gxf = np.load(os.path.join(OUT, "GX_xie.npz"))
    pc = torch.from_numpy(gxf["pc"]).double()
    for diffuse in (False, True):
        res = fu.xie_propagation_points_in_order(pc.clone(), 0.1, gxf["orders"], diffuse=diffuse, knn_mask=-1, C=3)
        out[f"xie_flip_{'d' if diffuse else 'n'}"] = res
        print(f"  G21 xie in_order f64 diffuse={diffuse}: flipped {int(res.sum())}")
    out["xie_inter_pc64"] = fu.xie_intersaction(pc, pc, eps=0.1, knn_mask=-1, C=3)[:64]
    # the BFS-route vote t5_d in float64 (harness as in gx2)
    seen = {}
    orig_in_order, orig_miqp = fu.xie_propagation_points_in_order, fu.MIQP

    def rec_in_order(pts, eps, order, *a, **k):
        seen["orders"] = np.array(order)
        res = orig_in_order(pts, eps, order, *a, **k)
        seen["flips"] = res.clone()
        return res

    def exhaustive(A, B):
        n = len(A)
        best, best_cost = None, None
        for tail in itertools.product((0, 1), repeat=n - 1):
            x = np.array((0,) + tail[::-1])
            cost = fu.cal_loss(x, A, B)
            if best_cost is None or cost < best_cost:
                best, best_cost = x, cost
        seen["status"] = best.astype(bool)
        return best.astype(float)

    fu.xie_propagation_points_in_order, fu.MIQP = rec_in_order, exhaustive
    try:
        pts = pc.clone()
        res = fu.xie_propagation_points_onbfstree(pts, 0.1, diffuse=True, starting_point=0, k=10, treshold=0.1, times=5,
                                                  knn_mask=-1, C=3)
        out["bfs_orders_t5_d"] = seen["orders"]
        out["bfs_flips_t5_d"] = seen["flips"]
        out["bfs_status_t5_d"] = seen["status"]
        out["bfs_result_t5_d"] = res
        out["bfs_normals_t5_d"] = pts[:, 3:]
        assert pts.dtype == torch.float64
        print(f"  G21 bfs t5_d f64: flipped {int(res.sum())}, status {seen['status'].astype(int)}")
    finally:
        fu.xie_propagation_points_in_order, fu.MIQP = orig_in_order, orig_miqp
    save("G21_f64_drivers", **out)


def make_inference_shim():
    """inference_utils.py imports models/ (torch_geometric, absent offline) at module level.
    Only its pure-torch fix_n_filter is on the path; load that one function's source object by
    executing the module with `models` stubbed - nothing is copied to disk."""
    for name in ("models", "models.pointcnn"):
        m = types.ModuleType(name)
        m.PointCNN = object
        sys.modules.setdefault(name, m)
    import importlib.util
    spec = importlib.util.spec_from_file_location("inference_utils_shim", f"{REF}/inference_utils.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["inference_utils_shim"] = mod
    spec.loader.exec_module(mod)


ALL = dict(G1=g1, G2=g2, G3=g3, G4=g4, G5=g5, G6=g6, G7=g7, G8=g8, G9=g9, G10=g10, G11=g11, G12=g12, GH=gh, GX=gx, G13=g13, G14=g14, G15=g15, G16=g16, GW=gw, GX2=gx2, G17=g17, G18=g18, G19=g19, G20=g20, G21=g21)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    sys.path.insert(0, REF)
    for name in ("open3d", "pymeshlab", "gurobipy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    make_inference_shim()
    fu, util = import_reference()
    todo = [s for s in args.only.split(",") if s] or list(ALL)
    for k in todo:
        print(f"[{k}]")
        t0 = time.time()
        with torch.no_grad():
            ALL[k](fu, util)
        print(f"  {time.time() - t0:.1f}s")
