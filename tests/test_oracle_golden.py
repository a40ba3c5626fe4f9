"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference itself
(tools/gen_golden.py).  CPU only.  Tolerances: the oracle's fp32 dense path is the same
op-sequence class as the reference but not the same instruction stream, so fp32 results are
compared at 2e-6 relative to the row norm (reference fp32 vs its own fp64 is 5.5e-7);
fp64 results at 1e-12; sign vectors and visit orders exactly."""
import numpy as np
import pytest
import torch

from conftest import csr_to_list, load_golden, rel_rowwise
from oracle import c_oracle
from oracle import dipole_oracle as O

F32_TOL = 2e-6
F64_TOL = 1e-12
t = torch.from_numpy


def test_G1_small_field_and_potential():
    g = load_golden("G1_field_grad_small")
    src, tgt = t(g["src"]), t(g["tgt"])
    for tag, eps in (("e5", 1e-5), ("e6", 1e-6)):
        assert rel_rowwise(O.field_grad(src, tgt, eps=eps), g[f"E6_{tag}"]) < F32_TOL
        assert rel_rowwise(O.field_grad(src, tgt[:, :3].contiguous(), eps=eps), g[f"E3_{tag}"]) < F32_TOL
        assert rel_rowwise(O.field_grad(src.double(), tgt.double(), eps=eps), g[f"E64_{tag}"]) < F64_TOL
        assert rel_rowwise(c_oracle.field_grad_f64(g["src"], g["tgt"], eps=eps), g[f"E64_{tag}"]) < F64_TOL
        assert np.allclose(O.potential(src, tgt).numpy(), g[f"phi_{tag}"], rtol=1e-5, atol=1e-5)
        assert np.allclose(c_oracle.potential_f64(g["src"], g["tgt"]), g[f"phi64_{tag}"], rtol=1e-12)
    assert rel_rowwise(O.field_grad(src[:1], tgt), g["E_S1"]) < F32_TOL
    assert rel_rowwise(O.field_grad(src, tgt[:1]), g["E_T1"]) < F32_TOL
    assert O.field_grad(src[:0], tgt).abs().max() == 0 and g["E_S0"].shape == (48, 3)


def test_G2_coincident_points_contribute_zero():
    g = load_golden("G2_zero_distance")
    s, tg = t(g["src"]), t(g["tgt"])
    assert rel_rowwise(O.field_grad(s, s), g["E_self"]) < F32_TOL
    assert rel_rowwise(O.field_grad(s, tg), g["E_part"]) < F32_TOL
    assert rel_rowwise(c_oracle.field_grad_f64(g["src"], g["src"]), g["E_self64"]) < F64_TOL
    assert rel_rowwise(c_oracle.field_grad_f64(g["src"], g["tgt"]), g["E_part64"]) < F64_TOL


def test_G3_analytic_single_dipole():
    g = load_golden("G3_analytic")
    src, tgt = t(g["src"]), t(g["tgt"])
    E = O.field_grad(src, tgt).numpy()
    assert np.allclose(E, g["E_e5"], rtol=1e-6, atol=1e-6)
    # closed forms: on axis -2 z^/(z^3+eps), equatorial +z^/(x^3+eps), coincident 0
    assert np.allclose(E[0], [0, 0, -2 / (0.5 ** 3 + 1e-5)], rtol=1e-5)
    assert np.allclose(E[2], [0, 0, 1 / (0.7 ** 3 + 1e-5)], rtol=1e-5)
    assert np.all(E[5] == 0)
    # eps = 0 with a coincident target: 0/0 = NaN, zeroed (and the reference prints a warning)
    assert np.allclose(O.field_grad(src, tgt, eps=0.0).numpy(), g["E_e0"], rtol=1e-6, atol=1e-6)
    assert np.all(g["E_e0"][5] == 0)
    assert np.array_equal(g["E_neg"], -g["E_e5"])            # linearity is bit exact in the reference
    assert np.allclose(O.potential(src, tgt).numpy(), g["phi_e5"], rtol=1e-6, atol=1e-6)
    assert g["phi_e5"][5] == 0                               # NaN -> 0


def test_G4_potential_lattice_and_mean():
    g = load_golden("G4_potential")
    assert np.array_equal(O.gen_grid().numpy(), g["grid"])
    pc = t(load_golden("G5_fandisk_allpairs")["pc"])
    phi = O.potential(pc, O.gen_grid()).numpy()
    scale = np.abs(g["phi_fandisk"]).max()
    assert np.abs(phi - g["phi_fandisk"]).max() / scale < 1e-5
    assert abs(float(O.measure_mean_potential(pc)) - float(g["mean_fandisk"])) / scale < 1e-5
    assert np.abs(c_oracle.potential_f64(pc.numpy(), g["grid"]) - g["phi64_fandisk"]).max() / scale < 1e-12
    node = t(g["node_src"])
    nphi = O.potential(node, O.gen_grid()).numpy()
    assert nphi[345] == 0 and g["node_phi"][345] == 0        # source on a lattice node: NaN -> 0
    assert np.abs(nphi - g["node_phi"]).max() / np.abs(g["node_phi"]).max() < 1e-5
    for name in ("ok", "fandisk", "hand"):
        assert float(g[f"mean_{name}"]) != 0


def test_G5_fandisk_rows():
    g = load_golden("G5_fandisk_allpairs")
    pc, rows = t(g["pc"]), g["rows"]
    E = O.field_grad(pc, pc[rows])
    assert rel_rowwise(E, g["E_rows"]) < F32_TOL
    E64 = c_oracle.field_grad_f64(g["pc"], g["pc"][rows])
    assert rel_rowwise(E64, g["E64_rows"]) < F64_TOL
    assert rel_rowwise(g["E_rows"], E64) < 2e-6               # the reference's own fp32 error
    assert np.array_equal((E64 * g["pc"][rows, 3:]).sum(-1) > 0, g["sign_all"][rows])


def test_G9_reference_field():
    g = load_golden("G9_reference_field")
    src = t(g["src"])
    out3 = O.reference_field(src, t(g["tgt3"])).numpy()
    assert np.array_equal(out3[:, :3], g["out3"][:, :3])
    assert np.abs(out3[:, 3:] - g["out3"][:, 3:]).max() < 5e-6
    assert np.allclose(np.linalg.norm(g["out3"][:, 3:], axis=1), 1, atol=1e-5)
    out6 = O.reference_field(src, t(g["tgt6"])).numpy()
    assert np.array_equal(out6, g["out6"])                   # sign decisions identical


def test_G18_reference_field_100k_sampled_rows():
    """Config 5 at S = T = 100 000 (G18): the oracle on the first 256 sampled target rows of the golden - field within
    1e-5, the sign decisions of those rows identical wherever the reference's E.n is above noise."""
    g15, g = load_golden("G15_boxunion_config3"), load_golden("G18_reference_field_100k")
    src = t(g15["pc"]).clone()
    N = src.shape[0]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    tgt3 = (src[:, :3] + 1e-3 * torch.randn(N, 3, generator=gen)).contiguous()
    assert np.array_equal(tgt3[:8].numpy(), g["tgt3_head"])
    flip = torch.rand(N, generator=torch.Generator().manual_seed(int(g["seed"]))) < 0.5
    rows = t(g["rows"][:256].astype(np.int64))
    E = O.field_grad(src, tgt3[rows]).numpy()
    ref = g["E_rows"][:256]
    assert (np.abs(E - ref) / np.linalg.norm(ref, axis=1, keepdims=True)).max() < 1e-5
    n = src[rows, 3:] * torch.where(flip[rows], -1.0, 1.0)[:, None]
    keep = (E * n.numpy()).sum(-1) >= 0
    ref_keep = np.unpackbits(g["keep"])[:N].astype(bool)[rows.numpy()]
    clear = np.abs(g["e_dot_n"][rows.numpy()]) > 1e-5 * np.linalg.norm(ref, axis=1)
    assert np.array_equal(keep[clear], ref_keep[clear]) and clear.mean() > 0.99


def test_G10_edge_weight():
    g = load_golden("G10_edge")
    w, invw = O.field_edge_calculator(t(g["a"]), t(g["b"]))
    assert abs(float(w) - float(g["w"])) <= 2e-5 * abs(float(g["w"]))
    assert float(invw) == -float(w) and float(g["invw"]) == -float(g["w"])
    n = g["a"].shape[0] * g["b"].shape[0]
    assert list(g["wcount"]) == [n, -n]
    assert list(g["wbool"]) == ([1, -1] if g["w"] > 0 else [-1, 1])


def _g11_cloud():
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(16000, 3, generator=gen)
    x = x / x.norm(dim=-1, keepdim=True) * 0.5
    n = torch.randn(16000, 3, generator=gen)
    n = n / n.norm(dim=-1, keepdim=True)
    return torch.cat([x, n], dim=1)


def test_G11_recursion_leaves():
    g = load_golden("G11_recursion")
    pc = _g11_cloud()
    assert np.array_equal(pc[:8].numpy(), g["pc_head"]) and np.array_equal(pc[-8:].numpy(), g["pc_tail"])
    assert O.source_leaves(16000, 15000) == [(0, 8000), (8000, 16000)]
    assert O.source_leaves(30001, 15000) == [(0, 15000), (15000, 22500), (22500, 30001)]
    assert O.source_leaves(100000, 15000)[0] == (0, 12500) and len(O.source_leaves(100000, 15000)) == 8
    rows = g["rows"]
    E64 = c_oracle.field_grad_f64(pc.numpy(), pc.numpy()[rows])
    assert rel_rowwise(E64, g["E64_rows"]) < F64_TOL
    assert rel_rowwise(g["E_rows"], E64) < 2e-6
    E = O.field_grad(pc, pc[rows[:16]])
    assert rel_rowwise(E, g["E_rows"][:16]) < F32_TOL


def sphere100k():
    """The headline cloud - ONE definition, shared with bench.py and the golden generator (tools/workloads.py)."""
    from tools.workloads import sphere_cloud
    return sphere_cloud()


def test_G12_sphere_generator_and_rows():
    g = load_golden("G12_sphere100k")
    pc = sphere100k()
    assert np.array_equal(pc[:16].numpy(), g["head"])
    rows = g["rows"][:8]
    E64 = c_oracle.field_grad_f64(pc.numpy(), pc.numpy()[rows])
    assert rel_rowwise(E64, g["E64_rows"][:8]) < F64_TOL
    assert rel_rowwise(g["E_rows"][:8], E64) < 2e-6
    assert float(g["mean_potential"]) > 0                     # outward normals -> positive mean potential


def test_G19_headline_workload_and_oracle_slab_rows():
    """G19 pins the bench workload against the reference: the inputs rebuilt from the seeds are the generator's (head
    rows, column sums, patch sizes, scramble), and the oracles reproduce the reference's own
    field_grad(pts[patch_k], pts[~patch_k]) - the dense-broadcast port on a row sample within fp32 rounding, the fp64 C
    arbiter on every row of one slab within the reference's own fp32 error."""
    from tools.workloads import headline_workload
    g = load_golden("G19_headline_sphere_patch_propagation")
    pc, patches, scramble = headline_workload()
    assert np.array_equal(pc[:16].numpy(), g["pc_head"]) and np.allclose(pc.double().sum(0).numpy(), g["pc_sum"], rtol=0, atol=1e-9)
    assert np.array_equal(np.array([len(p) for p in patches]), g["sizes"]) and np.array_equal(scramble, g["scramble"])
    assert sorted(g["order"].tolist()) == list(range(256)) and g["chosen"].shape == (255,)
    k = int(g["slab_patches"][1])
    mask = torch.ones(pc.shape[0], dtype=torch.bool)
    mask[patches[k]] = False
    others = pc[mask]
    ref = g[f"dE_{k}"]
    assert ref.shape == (others.shape[0], 3)
    rows = np.arange(0, others.shape[0], 97)
    E_port = O.field_grad(pc[patches[k]], others[rows])
    assert rel_rowwise(E_port, ref[rows]) < 2e-6
    E64 = c_oracle.field_grad_f64(pc[patches[k]].numpy(), others.numpy())
    assert rel_rowwise(ref, E64) < 2e-6
    # the start-patch rule on the reference's own curvatures: a clear minimum (no tie class on this cloud)
    c = np.abs(g["curv"])
    assert int(np.argmin(c)) == int(g["order"][0]) and np.sort(c)[1] / np.sort(c)[0] > 1.03


# ---- greedy drivers ---------------------------------------------------------------------------------
def _patch_case(g, tag):
    cname, dflag, wflag = tag.split("_")
    cloud = t(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
    w = t(g["weights"]) if wflag == "w" else None
    return cloud, patches, allp, dflag == "d", w


@pytest.mark.parametrize("tag", ["pf_n_nw", "pf_d_w", "sc_d_nw"])
def test_G6_patch_propagation(tag):
    g = load_golden("G6_patch_propagation")
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    start = int(g[f"order_{tag}"][0])
    out, trace = O.strongest_field_propagation(cloud, patches, allp, diffuse=diffuse, weights=w, start_patch=start)
    assert np.array_equal(trace["order"], g[f"order_{tag}"])
    assert np.array_equal(trace["flipped"], g[f"flipped_{tag}"])
    assert np.allclose(trace["chosen"], g[f"chosen_{tag}"], rtol=1e-4)
    sign = ((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy()
    assert np.array_equal(sign, g[f"sign_{tag}"])


def test_G6_start_patch_is_the_flattest():
    g = load_golden("G6_patch_propagation")
    curv = np.abs(g["curv"])
    assert int(np.argmin(curv)) == int(g["order_pf_n_nw"][0])


@pytest.mark.parametrize("tag", ["500_d", "50_n", "50_d"])
def test_G7_reps_propagation(tag):
    g = load_golden("G7_reps_propagation")
    cap, dflag = tag.split("_")
    cloud = t(g["pc_patchflip"])
    reps = list(zip(csr_to_list(g[f"rep_off_{cap}"], g[f"rep_idx_{cap}"]),
                    csr_to_list(g[f"rest_off_{cap}"], g[f"rest_idx_{cap}"])))
    start = int(g[f"order_{tag}"][0])
    out, trace = O.strongest_field_propagation_reps(cloud, reps, diffuse=(dflag == "d"), start_patch=start)
    assert np.array_equal(trace["order"], g[f"order_{tag}"])
    assert np.array_equal(trace["flipped"], g[f"flipped_{tag}"])
    sign = ((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy()
    assert (sign != g[f"sign_{tag}"]).sum() == 0


@pytest.mark.parametrize("tag", ["sub1000_n", "sub1000_d"])
def test_G8_point_propagation(tag):
    g = load_golden("G8_point_propagation")
    cloud = t(g["pc_sub1000"])
    out, order = O.strongest_field_propagation_points(cloud, diffuse=tag.endswith("_d"), starting_point=0)
    assert np.array_equal(order, g[f"order_{tag}"])
    sign = ((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy()
    assert np.array_equal(sign, g[f"sign_{tag}"])


# ---- the fork's "xie" pair functions (GX) --------------------------------------------------------------
def test_GX_xie_pair_functions():
    g = load_golden("GX_xie")
    src, tgt = t(g["src"]), t(g["tgt"])
    for C in (3, 2):
        f = O.xie_field(src, tgt, C=C).numpy()
        scale = np.abs(g[f"field_C{C}"]).max()
        assert f.shape == (40, 50, 3) and np.abs(f - g[f"field_C{C}"]).max() / scale < 1e-6
        m = O.xie_intersaction(src, tgt, C=C).numpy()
        assert np.abs(m - g[f"inter_C{C}"]).max() / np.abs(g[f"inter_C{C}"]).max() < 1e-6
    # coincident pairs are left undivided: ref = n_s
    assert np.allclose(g["field_C3"][0, 10], g["src"][10, 3:]) and np.allclose(g["field_C3"][4, 14], g["src"][14, 3:])
    mk = O.xie_intersaction(src, tgt, knn_mask=5, C=3).numpy()
    assert np.array_equal(mk != 0, g["inter_knn5"] != 0)
    assert np.abs(mk - g["inter_knn5"]).max() / np.abs(g["inter_knn5"]).max() < 1e-6
    assert np.allclose(O.xie_intersaction(src.double(), tgt.double()).numpy(), g["inter64"], rtol=1e-12, atol=1e-12)
    assert np.allclose(O.xie_distance(src, tgt).numpy(), g["distance"], rtol=1e-5)


@pytest.mark.parametrize("tag,diffuse,knn", [("n_k0", False, -1), ("d_k0", True, -1), ("n_k20", False, 20)])
def test_GX_ordered_propagation(tag, diffuse, knn):
    g = load_golden("GX_xie")
    res = O.xie_propagation_points_in_order(t(g["pc"]), g["orders"], diffuse=diffuse, knn_mask=knn).numpy()
    assert res.shape == (3, 1000)
    assert (res != g[f"flip_{tag}"]).sum() <= 2        # row sums within fp32 noise of zero may differ


def test_G13_hand_patch_propagation():
    g = load_golden("G13_hand")
    cloud = t(g["pc_patchflip"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    out, trace = O.strongest_field_propagation(cloud, list(enumerate(allp)), allp, diffuse=True,
                                               start_patch=int(g["order_patch"][0]))
    assert np.array_equal(trace["order"], g["order_patch"]) and np.array_equal(trace["flipped"], g["flipped_patch"])
    assert np.array_equal(((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g["sign_patch"])
