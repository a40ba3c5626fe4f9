"""Parity of the HIP path (through the C ABI of libdnp.so) with the reference: golden vectors
captured from the reference (tests/golden/, tools/gen_golden.py), the CPU oracle on seeded
inputs, and size-independent properties at the BASELINE sizes.  Run with `-m gpu` on an MI355X.

Tolerances: field values within 1e-5 relative to the per-target vector norm in
fp32 (1e-12 in fp64); sign decisions, visit orders and flip vectors identical."""
import numpy as np
import pytest
import torch

from conftest import check_chosen, csr_to_list, load_golden, rel_rowwise
from dipole_normal_prop_amd import _lib
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import patch_drivers as pd, point_driver as ptd  # noqa: E402
from dipole_normal_prop_amd import util
from oracle import c_oracle
from oracle import dipole_oracle as O
from test_oracle_golden import _g11_cloud, sphere100k

pytestmark = pytest.mark.gpu
TOL = 1e-5
t = torch.from_numpy


def test_native_library_is_loaded(dev):
    lib = _lib.require_device()
    assert lib.dnp_device_count() >= 1
    props = torch.cuda.get_device_properties(0)
    assert "gfx950" in props.gcnArchName


# ---- K1 / K2 against the goldens ---------------------------------------------------------------------
def test_G1_field_grad_small(dev):
    g = load_golden("G1_field_grad_small")
    src, tgt = t(g["src"]).to(dev), t(g["tgt"]).to(dev)
    for tag, eps in (("e5", 1e-5), ("e6", 1e-6)):
        E = fu.field_grad(src, tgt, eps=eps)
        assert E.shape == (48, 3) and E.dtype == torch.float32 and E.device == src.device
        assert rel_rowwise(E.cpu(), g[f"E6_{tag}"]) < TOL
        assert rel_rowwise(fu.field_grad(src, tgt[:, :3].contiguous(), eps=eps).cpu(), g[f"E3_{tag}"]) < TOL
        assert rel_rowwise(fu.field_grad(src, tgt[:, :3], eps=eps).cpu(), g[f"E3_{tag}"]) < TOL  # strided view
        assert rel_rowwise(fu.field_grad(src.double(), tgt.double(), eps=eps).cpu(), g[f"E64_{tag}"]) < 1e-12
        phi = fu.potential(src, tgt, eps=eps)
        assert phi.shape == (48,)
        assert np.allclose(phi.cpu().numpy(), g[f"phi_{tag}"], rtol=1e-5, atol=1e-5)
        assert np.allclose(fu.potential(src.double(), tgt.double()).cpu().numpy(), g[f"phi64_{tag}"], rtol=1e-11)
    assert rel_rowwise(fu.field_grad(src[:1], tgt).cpu(), g["E_S1"]) < TOL
    assert rel_rowwise(fu.field_grad(src, tgt[:1]).cpu(), g["E_T1"]) < TOL
    E0 = fu.field_grad(src[:0], tgt)
    assert E0.shape == (48, 3) and float(E0.abs().max()) == 0
    assert fu.field_grad(src, tgt[:0]).shape == (0, 3)
    assert fu.potential(src, tgt[:0]).shape == (0,)


def test_inputs_untouched_and_cpu_tensors_round_trip(dev):
    g = load_golden("G1_field_grad_small")
    src, tgt = t(g["src"]), t(g["tgt"])
    s0, t0 = src.clone(), tgt.clone()
    E = fu.field_grad(src, tgt)                       # CPU in -> staged to the device -> CPU out
    assert E.device.type == "cpu" and rel_rowwise(E, g["E6_e5"]) < TOL
    assert torch.equal(src, s0) and torch.equal(tgt, t0)
    sd = src.to(dev)
    sc = sd.clone()
    fu.field_grad(sd, tgt.to(dev))
    assert torch.equal(sd, sc)


def test_G2_zero_distance(dev):
    g = load_golden("G2_zero_distance")
    s, tg = t(g["src"]).to(dev), t(g["tgt"]).to(dev)
    assert rel_rowwise(fu.field_grad(s, s).cpu(), g["E_self"]) < TOL
    assert rel_rowwise(fu.field_grad(s, tg).cpu(), g["E_part"]) < TOL
    assert rel_rowwise(fu.field_grad(s.double(), s.double()).cpu(), g["E_self64"]) < 1e-12


def test_G3_analytic_and_eps0_and_linearity(dev):
    g = load_golden("G3_analytic")
    src, tgt = t(g["src"]).to(dev), t(g["tgt"]).to(dev)
    E = fu.field_grad(src, tgt).cpu().numpy()
    assert np.allclose(E, g["E_e5"], rtol=1e-6, atol=1e-5)
    assert np.allclose(E[0], [0, 0, -2 / (0.5 ** 3 + 1e-5)], rtol=1e-5)
    assert np.allclose(E[2], [0, 0, 1 / (0.7 ** 3 + 1e-5)], rtol=1e-5)
    assert np.all(E[5] == 0)                                        # coincident pair contributes exactly 0
    E0 = fu.field_grad(src, tgt, eps=0.0).cpu().numpy()              # eps = 0: 0/0 row -> NaN -> zeroed
    assert np.allclose(E0, g["E_e0"], rtol=1e-6, atol=1e-5) and np.all(E0[5] == 0)
    neg = src.clone()
    neg[:, 3:] *= -1
    assert np.array_equal(fu.field_grad(neg, tgt).cpu().numpy(), -E)  # bit-exact, as in the reference
    phi = fu.potential(src, tgt).cpu().numpy()
    assert np.allclose(phi, g["phi_e5"], rtol=1e-6, atol=1e-5) and phi[5] == 0


def test_G4_potential_lattice(dev):
    g = load_golden("G4_potential")
    for name, pc in (("fandisk", load_golden("G5_fandisk_allpairs")["pc"]), ("ok", load_golden("G9_reference_field")["src"])):
        pcd = t(pc).to(dev)
        phi = fu.potential(pcd, util.gen_grid().to(dev)).cpu().numpy()
        scale = np.abs(g[f"phi_{name}"]).max()
        assert np.abs(phi - g[f"phi_{name}"]).max() / scale < TOL
        m = fu.measure_mean_potential(pcd)
        assert m.dim() == 0 and abs(float(m) - float(g[f"mean_{name}"])) / scale < TOL
        assert (float(m) < 0) == (float(g[f"mean_{name}"]) < 0)
    node = t(g["node_src"]).to(dev)
    nphi = fu.potential(node, util.gen_grid().to(dev)).cpu().numpy()
    assert nphi[345] == 0                                            # source on a lattice node: NaN -> 0
    assert np.abs(nphi - g["node_phi"]).max() / np.abs(g["node_phi"]).max() < TOL
    assert abs(float(fu.measure_mean_potential(node)) - float(g["node_mean"])) < TOL * np.abs(g["node_phi"]).max()


def test_G5_fandisk_all_pairs(dev):
    """BASELINE config 2: fandisk.xyz full all-pairs field on one GPU, fp32."""
    g = load_golden("G5_fandisk_allpairs")
    pc = t(g["pc"]).to(dev)
    E = fu.field_grad(pc, pc).cpu().numpy()
    assert rel_rowwise(E[g["rows"]], g["E_rows"]) < TOL
    assert rel_rowwise(E[g["rows"]], g["E64_rows"]) < TOL
    nrm = np.linalg.norm(E.astype(np.float64), axis=-1)
    assert np.abs(nrm - g["norm_all"]).max() / g["norm_all"].max() < TOL      # every row, via its norm
    assert abs(nrm.sum() - float(g["sum_norm"])) / float(g["sum_norm"]) < 1e-6
    assert np.array_equal((E * g["pc"][:, 3:]).sum(-1) > 0, g["sign_all"])     # all 11 031 sign decisions


def test_G9_reference_field(dev):
    g = load_golden("G9_reference_field")
    src = t(g["src"]).to(dev)
    out3 = fu.reference_field(src, t(g["tgt3"]).to(dev)).cpu().numpy()
    assert out3.shape == (10000, 6) and np.array_equal(out3[:, :3], g["out3"][:, :3])
    assert np.abs(out3[:, 3:] - g["out3"][:, 3:]).max() < 2e-5
    tgt6 = t(g["tgt6"]).to(dev)
    out6 = fu.reference_field(src, tgt6)
    assert out6.data_ptr() == tgt6.data_ptr()                        # 6-column input is updated in place
    assert np.array_equal(out6.cpu().numpy(), g["out6"])             # sign decisions identical
    E = fu.field_grad(src, t(g["tgt3"]).to(dev)).cpu().numpy()
    assert rel_rowwise(E, g["E64"]) < TOL


def test_reference_field_fused_call_equals_the_two_step_form(dev, monkeypatch):
    """dnp_reference_field_* (field + tail in one call, round 3) against field_grad followed by the reference's own
    torch expressions: identical sign decisions and in-place behaviour, normals within rounding; fp32 and fp64, a
    3-column VIEW of a 6-column tensor (row stride 6), CPU tensors (two-step form), empty source / target sets, and
    the raw C ABI's argument checks."""
    lib = _lib.require_device()
    g = load_golden("G9_reference_field")
    src, tgt3, tgt6 = t(g["src"]).to(dev), t(g["tgt3"]).to(dev), t(g["tgt6"]).to(dev)

    def two_step(a, b):
        monkeypatch.setattr(fu, "_reference_field_fused", lambda p, q: None)
        try:
            return fu.reference_field(a, b)
        finally:
            monkeypatch.undo()

    for dt, tol in ((torch.float32, 2e-6), (torch.float64, 1e-13)):
        a = src.to(dt)
        one = fu.reference_field(a, tgt3.to(dt))
        two = two_step(a, tgt3.to(dt))
        assert one.shape == (10000, 6) and one.dtype == dt and torch.equal(one[:, :3], two[:, :3])
        assert float((one[:, 3:] - two[:, 3:]).abs().max()) < tol
        assert float((one[:, 3:].norm(dim=1) - 1).abs().max()) < 1e-6
        w1, w2 = tgt6.to(dt).clone(), tgt6.to(dt).clone()
        r1 = fu.reference_field(a, w1)
        two_step(a, w2)
        assert r1.data_ptr() == w1.data_ptr() and torch.equal(w1, w2)
        view = tgt6.to(dt).clone()
        got = fu.reference_field(a, view[:, :3])                      # 3-column view, row stride 6
        assert torch.equal(got, one)
    assert np.array_equal(fu.reference_field(src, tgt6.clone()).cpu().numpy(), g["out6"])
    # CPU tensors: staged, the same library call, results back (6 columns in place, also into a strided view, fp64 too)
    host6 = tgt6.cpu().clone()
    back = fu.reference_field(src.cpu(), host6)
    assert back.data_ptr() == host6.data_ptr() and np.array_equal(host6.numpy(), g["out6"])
    wide = torch.zeros(10000, 8)
    wide[:, 1:7] = tgt6.cpu()
    fu.reference_field(src.cpu(), wide[:, 1:7])
    assert np.array_equal(wide[:, 1:7].numpy(), g["out6"]) and float(wide[:, 0].abs().max()) == 0 and float(wide[:, 7].abs().max()) == 0
    h3 = fu.reference_field(src.cpu(), tgt3.cpu())
    assert h3.device.type == "cpu" and torch.equal(h3, fu.reference_field(src, tgt3).cpu())
    h64, d64 = tgt6.cpu().double().clone(), tgt6.double().clone()
    fu.reference_field(src.cpu().double(), h64)
    fu.reference_field(src.double(), d64)
    assert torch.equal(h64, d64.cpu())
    # empty sets
    e = fu.reference_field(src[:0], tgt3[:5].clone())
    assert e.shape == (5, 6) and float(e[:, 3:].abs().max()) == 0
    keep = tgt6[:5].clone()
    assert torch.equal(fu.reference_field(src[:0], keep), tgt6[:5])   # zero field: E.n = 0 >= 0 -> nothing flips
    assert fu.reference_field(src, tgt3[:0].clone()).shape == (0, 6)
    # C ABI: bad form / missing out are argument errors, not launches
    ws = torch.empty(lib.dnp_field_grad_workspace_bytes(100, 50, 15000), dtype=torch.uint8, device=dev)
    args = (_lib.ptr(src), 100, 6, _lib.ptr(tgt6), 50, 6)
    assert lib.dnp_reference_field_f32(*args, 3, 1e-5, 15000, None, 6, None, _lib.ptr(ws), ws.numel(), _lib.current_stream()) == -1
    assert lib.dnp_reference_field_f32(*args, 1, 1e-5, 15000, None, 6, None, _lib.ptr(ws), ws.numel(), _lib.current_stream()) == -1
    assert lib.dnp_reference_field_f32(_lib.ptr(src), 100, 6, _lib.ptr(tgt3), 50, 3, 2, 1e-5, 15000, None, 6, None,
                                       _lib.ptr(ws), ws.numel(), _lib.current_stream()) == -1
    assert b"form" in lib.dnp_last_error()


def test_G10_edge_weight(dev):
    g = load_golden("G10_edge")
    a, b = t(g["a"]).to(dev), t(g["b"]).to(dev)
    w, invw = fu.field_edge_calculator(a, b)
    assert isinstance(w, np.ndarray) and w.dtype == np.float32
    assert abs(float(w) - float(g["w"])) <= 2e-5 * abs(float(g["w"])) and float(invw) == -float(w)
    assert list(fu.field_edge_calculator_bool(a, b)) == list(g["wbool"])
    assert list(fu.field_edge_calculator_count(a, b)) == list(g["wcount"])
    assert abs(float(fu.self_interaction_all(a)) - float(g["wself"])) <= 2e-5 * abs(float(g["wself"]))


def test_G11_recursion_leaf_semantics(dev):
    g = load_golden("G11_recursion")
    pc = _g11_cloud().to(dev)
    E = fu.field_grad(pc, pc)                                        # 16 000 > max_pts: two source leaves
    assert rel_rowwise(E[t(g["rows"]).to(dev)].cpu(), g["E_rows"]) < TOL
    # other leaf structures change only the summation tree: all agree with fp64 within tolerance
    # (random normals: |E| is a random-walk residue, the hardest case for a relative bound)
    E64 = fu.field_grad(pc.double(), pc.double()).cpu().numpy()
    n64 = np.linalg.norm(E64, axis=1)
    for other in (E, fu.field_grad(pc, pc, recursive=False), fu.field_grad(pc, pc, max_pts=1000)):
        err = np.linalg.norm(other.cpu().numpy() - E64, axis=1)
        assert np.all(err <= TOL * n64)                              # plain 1e-5 |E| on every row (worst measured: 4.5e-6)
        assert np.quantile(err / n64, 0.99) < 3e-6 and np.median(err / n64) < 5e-7


def test_recursion_leaf_nan_filter_is_per_leaf(dev):
    """An Inf/NaN leaf sum is zeroed per leaf (field_utils.py:110-115): with eps = 0 a coincident
    pair poisons only the leaf that holds it; the other leaf's contribution survives."""
    pc = _g11_cloud()[:4000].clone()
    tgt = pc[10:11, :3].clone()                                      # coincides with source row 10 (leaf 0)
    ref = O.field_grad(pc, tgt, eps=0.0, max_pts=2000)
    leaf1 = O.field_grad(pc[2000:], tgt, eps=0.0, recursive=False)
    assert torch.equal(ref, leaf1)                                   # oracle: leaf 0 zeroed, leaf 1 kept
    E = fu.field_grad(pc.to(dev), tgt.to(dev), eps=0.0, max_pts=2000).cpu()
    assert rel_rowwise(E, ref) < TOL
    assert float(fu.field_grad(pc.to(dev), tgt.to(dev), eps=0.0, recursive=False).abs().max()) == 0


def test_G12_sphere100k_rows_and_orientation_sign(dev):
    g = load_golden("G12_sphere100k")
    pc = sphere100k().to(dev)
    E = fu.field_grad(pc, pc[t(g["rows"]).to(dev)])
    assert rel_rowwise(E.cpu(), g["E_rows"]) < TOL and rel_rowwise(E.cpu(), g["E64_rows"]) < TOL
    m = float(fu.measure_mean_potential(pc))
    assert m > 0 and abs(m - float(g["mean_potential"])) < 1e-4 * abs(float(g["mean_potential"]))


# ---- seeded inputs against the oracle: ragged sizes, gathers, accumulate ---------------------------------
def term_magnitude(src, tgt, eps=1e-5):
    """sum_s |term_s| per target: |3 (p.r^) r^ - p| / (|r|^3 + eps) <= 2 |p| / (|r|^3 + eps)."""
    out = np.zeros(tgt.shape[0])
    for i in range(0, tgt.shape[0], 256):
        r = src[None, :, :3].astype(np.float64) - tgt[i:i + 256, None, :3].astype(np.float64)
        d = np.linalg.norm(r, axis=-1)
        out[i:i + 256] = (2 * np.linalg.norm(src[:, 3:6], axis=-1)[None, :] / (d ** 3 + eps) * (d > 0)).sum(axis=1)
    return out


@pytest.mark.parametrize("S,T", [(1, 1), (3, 700), (255, 257), (256, 512), (513, 1025), (2049, 33), (5000, 1)])
def test_ragged_sizes_against_oracle(dev, S, T):
    gen = torch.Generator().manual_seed(S * 7919 + T)
    src = torch.rand(S, 6, generator=gen) - 0.5
    tgt = torch.rand(T, 3, generator=gen) - 0.5
    tgt[: min(S, T) // 3] = src[: min(S, T) // 3, :3]               # some coincident pairs
    ref = c_oracle.field_grad_f64(src.numpy(), tgt.numpy())
    E = fu.field_grad(src.to(dev), tgt.to(dev)).cpu().numpy()
    # random dipoles: some rows are cancellation residues (|E| << sum of |terms|), where no fp32
    # evaluation - the reference's included - can hold a bound relative to |E| alone; the bound that any
    # fp32 evaluation can hold is relative to the magnitude of what is summed, so the row tolerance is
    # 1e-5 |E| + 16 u * sum_s |term_s|  (u = 2^-24)
    mag = term_magnitude(src.numpy(), tgt.numpy())[:, None]
    err = np.linalg.norm(E - ref, axis=1)
    assert np.all(err <= TOL * np.linalg.norm(ref, axis=1) + 16 * 2.0 ** -24 * mag[:, 0])
    assert np.median(err / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)) < 1e-6
    refp = c_oracle.potential_f64(src.numpy(), tgt.numpy())
    phi = fu.potential(src.to(dev), tgt.to(dev)).cpu().numpy()
    scale = np.abs(refp).max() if np.abs(refp).max() > 0 else 1.0
    assert np.abs(phi - refp).max() / scale < TOL


def test_raw_c_abi_gather_scatter_accumulate(dev):
    """Call dnp_field_grad_f32 directly: row gathers on both operands, scattered accumulate into a
    full-size E - the `E[mask] = E[mask] + field_grad(pts[patch], pts[mask])` of the drivers."""
    lib = _lib.require_device()
    gen = torch.Generator().manual_seed(42)
    pts = (torch.rand(3000, 6, generator=gen) - 0.5)
    src_idx = torch.randperm(3000, generator=gen)[:700]
    tgt_idx = torch.randperm(3000, generator=gen)[:1900]
    E0 = torch.randn(3000, 3, generator=gen)
    ref = E0.clone()
    ref[tgt_idx] += t(c_oracle.field_grad_f64(pts[src_idx].numpy(), pts[tgt_idx].numpy())).float()
    d_pts, d_E = pts.to(dev), E0.to(dev)
    d_si, d_ti = src_idx.to(dev), tgt_idx.to(dev)
    nbytes = lib.dnp_field_grad_workspace_bytes(700, 1900, 15000)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = lib.dnp_field_grad_f32(_lib.ptr(d_pts), 700, 6, _lib.ptr(d_si), _lib.ptr(d_pts), 1900, 6, _lib.ptr(d_ti),
                                1e-5, 15000, _lib.ptr(d_E), 3, 1, 1, None, None, _lib.ptr(ws), nbytes, _lib.current_stream())
    assert rc == 0, lib.dnp_last_error()
    torch.cuda.synchronize()
    out = d_E.cpu()
    assert rel_rowwise(out[tgt_idx], ref[tgt_idx]) < TOL
    untouched = torch.ones(3000, dtype=torch.bool)
    untouched[tgt_idx] = False
    assert torch.equal(out[untouched], E0[untouched])
    # too-small workspace is refused, nothing is launched
    rc = lib.dnp_field_grad_f32(_lib.ptr(d_pts), 700, 6, _lib.ptr(d_si), _lib.ptr(d_pts), 1900, 6, _lib.ptr(d_ti),
                                1e-5, 15000, _lib.ptr(d_E), 3, 1, 1, None, None, _lib.ptr(ws), 16, _lib.current_stream())
    assert rc == -3


def test_threads_call_concurrently(dev):
    """The reference calls field_grad from several Python threads (util.py:187-196)."""
    import threading
    g = load_golden("G2_zero_distance")
    s = t(g["src"]).to(dev)
    outs = [None] * 6

    def work(i):
        outs[i] = fu.field_grad(s, s).cpu()

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [x.start() for x in th]
    [x.join() for x in th]
    for o in outs:
        assert torch.equal(o, outs[0]) and rel_rowwise(o, g["E_self"]) < TOL


# ---- batched per-patch fields, interaction matrix, combination ----------------------------------------
def test_patch_boxes_and_boxed_fields(dev):
    """dnp_patch_boxes_f32 against per-patch min / max, with and without the row gather; the slabs of the scalar-unit
    kernel are bit-identical whether it is handed the boxes or finds them itself (boxunion, 100 000 points, the
    reference's 369 patches, cloud sorted by patch)."""
    g = load_golden("G15_boxunion_config3")
    cloud = t(g["pc"]).to(dev)
    off_np, idx_np = g["patch_off"].astype(np.int64), g["patch_idx"].astype(np.int64)
    off, idx = t(off_np).to(dev), t(idx_np).to(dev)
    P = len(off_np) - 1
    boxes_gather = fu._patch_boxes(cloud, off, idx)
    swork = cloud[idx].contiguous()
    boxes = fu._patch_boxes(swork, off, None)
    assert boxes.shape == (P, 6) and torch.equal(boxes, boxes_gather)
    ref = torch.stack([torch.cat([swork[off_np[k]:off_np[k + 1], :3].min(0).values, swork[off_np[k]:off_np[k + 1], :3].max(0).values])
                       for k in range(P)])
    assert torch.equal(boxes, ref)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    plain = fu._patch_slabs(swork, off, None, point_patch, 100, 140, 1e-5)
    boxed = fu._patch_slabs(swork, off, None, point_patch, 100, 140, 1e-5, boxes)
    assert torch.equal(plain, boxed)
    k = 117 - 100                                        # one slab against the fp64 oracle
    others = (point_patch != 117).cpu()
    src = swork[off_np[117]:off_np[118]].cpu().numpy()
    rows = torch.nonzero(others).flatten()[::97]
    ref64 = c_oracle.field_grad_f64(src, swork.cpu()[rows].numpy())
    assert rel_rowwise(boxed[k].cpu()[rows], ref64) < TOL


def test_tile_tables_and_fused_interaction_rows(dev):
    """Round 3: the target-tile box table and the interaction partials of the pair kernel's epilogue.
    dnp_tile_boxes_f32 equals per-tile min / max; the slabs are bit-identical with and without the tile table and with
    and without the partials; W assembled from the partials (dnp_interactions_from_tiles) equals the K3 pass over the
    slabs up to fp64 reassociation - on boxunion (369 patches of >= 100 points: some 128-row tiles straddle three
    patches; rounds 3-4 refused the fused form there, round 5 gives the partials a third group slot) and on a cut with
    patches of >= 128 points plus rows that are in no patch (two slots)."""
    lib = _lib.require_device()
    g = load_golden("G15_boxunion_config3")
    cloud = t(g["pc"]).to(dev)
    off_np, idx_np = g["patch_off"].astype(np.int64), g["patch_idx"].astype(np.int64)
    swork = cloud[t(idx_np).to(dev)].contiguous()
    N = swork.shape[0]
    R = int(lib.dnp_patch_tile_rows())
    assert R == 128
    tiles = fu._TileTables(swork, np.diff(off_np))
    ref = torch.stack([torch.cat([swork[i:i + R, :3].min(0).values, swork[i:i + R, :3].max(0).values]) for i in range(0, N, R)])
    assert tiles.n_tiles == ref.shape[0] and torch.equal(tiles.boxes, ref)
    assert tiles.fused and tiles.slots == 3              # min patch 100 < 128 rows: a tile can hold three patches
    off = t(off_np).to(dev)
    P = len(off_np) - 1
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes = fu._patch_boxes(swork, off, None)
    boxed = fu._patch_slabs(swork, off, None, point_patch, 100, 140, 1e-5, boxes)
    tiled = fu._patch_slabs(swork, off, None, point_patch, 100, 140, 1e-5, boxes, tiles.boxes)
    assert torch.equal(boxed, tiled)
    dE, W = fu._slabs_and_rows(swork, off, point_patch, 100, 140, 1e-5, boxes, tiles)     # three-slot partials
    W3 = fu._interaction_rows(boxed, swork, off, None)
    assert torch.equal(dE, boxed) and float((W - W3).abs().max()) <= 1e-12 * float(W3.abs().max())
    dE, W = fu._slabs_and_rows(swork, off, point_patch, 0, P, 1e-5, boxes, tiles, np.diff(off_np))   # all 369, split tail rule
    W3 = fu._interaction_rows(dE, swork, off, None)
    assert float((W - W3).abs().max()) <= 1e-12 * float(W3.abs().max())
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.dnp_check_tile_groups(_lib.ptr(point_patch), N, 3, _lib.ptr(bad), _lib.current_stream()) == 0 and int(bad.item()) == 0
    assert lib.dnp_check_tile_groups(_lib.ptr(point_patch), N, 2, _lib.ptr(bad), _lib.current_stream()) == 0 and int(bad.item()) > 0

    # a cut that allows the fused form: patches of 128..700 rows over the first 90 000 rows, the rest in no patch
    rng = np.random.default_rng(3)
    sizes = []
    while sum(sizes) < 90000:
        sizes.append(int(rng.integers(128, 700)))
    sizes[-1] -= sum(sizes) - 90000
    if sizes[-1] < 128:
        sizes[-2] += sizes.pop()
    sizes = np.array(sizes + [0, 300], dtype=np.int64)    # an empty patch and a last one
    off2_np = np.concatenate([[0], np.cumsum(sizes)])
    off2 = t(off2_np).to(dev)
    P2 = len(sizes)
    covered = int(off2_np[-1])
    pp2 = torch.cat([torch.repeat_interleave(torch.arange(P2, device=dev), off2[1:] - off2[:-1]),
                     torch.full((N - covered,), -1, dtype=torch.int64, device=dev)])
    tiles2 = fu._TileTables(swork, sizes)
    assert not tiles2.fused                               # an empty patch makes a tile's group ids jump: refused
    sizes3 = np.array(sizes[:-2].tolist() + [300], dtype=np.int64)
    off3_np = np.concatenate([[0], np.cumsum(sizes3)])
    off3 = t(off3_np).to(dev)
    P3 = len(sizes3)
    pp3 = torch.cat([torch.repeat_interleave(torch.arange(P3, device=dev), off3[1:] - off3[:-1]),
                     torch.full((N - int(off3_np[-1]),), -1, dtype=torch.int64, device=dev)])
    tiles3 = fu._TileTables(swork, sizes3)
    assert tiles3.fused and tiles3.slots == 2
    boxes3 = fu._patch_boxes(swork, off3, None)
    for b0, b1 in ((0, P3), (7, 31)):
        dE, W = fu._slabs_and_rows(swork, off3, pp3, b0, b1, 1e-5, boxes3, tiles3)
        plain = fu._patch_slabs(swork, off3, None, pp3, b0, b1, 1e-5, boxes3)
        assert torch.equal(dE, plain)
        W3 = fu._interaction_rows(plain, swork, off3, None)
        assert W.shape == W3.shape == (b1 - b0, P3)
        assert float((W - W3).abs().max()) <= 1e-12 * float(W3.abs().max())
        assert float(W[:, :].abs().max()) > 0
    # every (slab, patch) entry is the plain definition: fp32 dot per point, fp64 sum over the patch
    k = 11
    d32 = (dE[k - 7][:, 0] * swork[:, 3] + dE[k - 7][:, 1] * swork[:, 4] + dE[k - 7][:, 2] * swork[:, 5]).double()
    want = torch.stack([d32[off3_np[j]:off3_np[j + 1]].sum() for j in range(P3)])
    assert float((W[k - 7] - want).abs().max()) <= 1e-6 * float(want.abs().max())


def test_tile_group_precondition_is_checkable_on_the_device(dev):
    """dnp_check_tile_groups: the precondition of the fused interaction partials (every 128-row tile inside at most two
    groups) counted on the device, for callers of the raw C ABI that have no patch sizes on the host (round-3 verdict:
    dnp_patch_fields_tiled_f32 trusted the caller - a wrong W, silently).  A cut that satisfies it: 0, and W from the tiles
    equals W from the slabs; a cut with a 20-row patch inside a tile: counted, and the host rule agrees."""
    lib = _lib.require_device()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1000, 3, generator=g)
    pc = torch.cat([0.4 * x / x.norm(dim=1, keepdim=True), x / x.norm(dim=1, keepdim=True)], 1).to(dev)
    for sizes, want_ok, slots in ((np.array([300, 200, 380, 120]), True, 2), (np.array([300, 100, 20, 300, 280]), False, 3),
                                  (np.array([300, 60, 20, 30, 310, 280]), False, 0)):
        off = t(np.concatenate([[0], np.cumsum(sizes)])).to(dev)
        pp = torch.repeat_interleave(torch.arange(len(sizes), device=dev), off[1:] - off[:-1])
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.dnp_check_tile_groups(_lib.ptr(pp), 1000, 2, _lib.ptr(bad), _lib.current_stream()) == 0
        assert (int(bad.item()) == 0) == want_ok
        bad3 = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.dnp_check_tile_groups(_lib.ptr(pp), 1000, 3, _lib.ptr(bad3), _lib.current_stream()) == 0
        assert (int(bad3.item()) == 0) == (slots != 0)
        assert fu._tiles_within_two_groups(sizes, 1000, 128) == want_ok
        boxes, tiles = fu._patch_boxes(pc, off, None), fu._TileTables(pc, sizes)
        assert tiles.slots == slots and tiles.fused == (slots != 0)
        if slots:
            dE, W = fu._slabs_and_rows(pc, off, pp, 0, len(sizes), 1e-5, boxes, tiles, sizes)
            W3 = fu._interaction_rows(dE, pc, off, None)
            assert float((W - W3).abs().max()) <= 1e-12 * float(W3.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_launch_order_does_not_change_a_bit(dev, dtype):
    """dnp_patch_fields_ordered_* (round 5): the launch rows evaluate the patches in a given ORDER - the drivers' longest-patch-
    first plan, a random permutation, the reverse - and slab k stays patch p_begin + k: slabs and interaction partials bit-identical
    to the patch-order launch, with and without a split tail behind it (fp32: the tail's items are then the LAST ROWS, i.e. other
    patches than in patch order - still the same bits), on an uneven cut (41..900 points, rows in no patch) and on a range inside
    the headline cloud.  Outputs poisoned first; the plan itself: longest first, stable, no tail when the shortest patch is short."""
    from tools.workloads import headline_workload
    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr([p.to(dev) for p in patches], dev)
    swork = pc.to(dev)[idx].contiguous().to(dtype)
    N, P = swork.shape[0], len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(swork, off, None), fu._TileTables(swork, sizes)
    rng = np.random.default_rng(3)

    def check(sw, of, pp, bx, tl, p0, p1, slots):
        K = p1 - p0
        wp0 = torch.full((K, tl.n_tiles, slots), float("nan"), dtype=torch.float64, device=dev) if slots else None
        ref = fu._patch_slabs(sw, of, None, pp, p0, p1, 1e-5, bx, tl.boxes, wp0, 1)
        plan = np.argsort(-np.diff(of.cpu().numpy())[p0:p1], kind="stable")
        for perm in (plan, rng.permutation(K), np.arange(K)[::-1].copy()):
            order = t(perm.astype(np.int32)).to(dev)
            for ss in ((1, -1, -3, -K) if dtype == torch.float32 else (1,)):
                wp = torch.full((K, tl.n_tiles, slots), float("nan"), dtype=torch.float64, device=dev) if slots else None
                torch.full((K, sw.shape[0], 3), float("nan"), dtype=dtype, device=dev)
                got = fu._patch_slabs(sw, of, None, pp, p0, p1, 1e-5, bx, tl.boxes, wp, ss, order)
                assert torch.equal(got, ref), (perm[:4], ss)
                assert slots == 0 or torch.equal(wp, wp0), (perm[:4], ss)

    check(swork, off, point_patch, boxes, tiles, 40, 64, tiles.slots)
    sizes2 = np.array([64, 128, 129, 300, 512, 513, 900, 41, 390, 2000], dtype=np.int64)
    off2 = t(np.concatenate([[0], np.cumsum(sizes2)])).to(dev)
    n2 = int(sizes2.sum()) + 77
    sw2 = swork[:n2].contiguous()
    pp2 = torch.cat([torch.repeat_interleave(torch.arange(len(sizes2), device=dev), off2[1:] - off2[:-1]),
                     torch.full((77,), -1, dtype=torch.int64, device=dev)])
    boxes2, tiles2 = fu._patch_boxes(sw2, off2, None), fu._TileTables(sw2, sizes2)
    check(sw2, off2, pp2, boxes2, tiles2, 0, len(sizes2), 0)
    check(sw2, off2, pp2, boxes2, tiles2, 2, 9, 0)
    # the plan: longest first (stable); a tail only when the shortest patch has more than 256 points
    order, split = fu._launch_plan(sizes2, 100000, dev)
    assert order.cpu().tolist() == [9, 6, 5, 4, 8, 3, 2, 1, 0, 7] and split == 1
    order, split = fu._launch_plan(np.array([300, 420, 390, 410, 350]), 100000, dev)
    assert order.cpu().tolist() == [1, 3, 2, 4, 0] and split == -3                 # 300 + 350 + 390 >= 1000
    assert fu._launch_plan(np.array([500, 400, 300]), 100000, dev) == (None, -3)   # already longest first: no table
    assert fu._launch_plan(np.array([500, 400, 300]), 100000, dev, tail=False) == (None, 1)
    # the raw C ABI refuses an order on the gathered layout
    lib = _lib.load()
    dE = torch.empty((2, N, 3), dtype=dtype, device=dev)
    fn = lib.dnp_patch_fields_ordered_f64 if dtype == torch.float64 else lib.dnp_patch_fields_ordered_f32
    args = [_lib.ptr(swork), N, 6, _lib.ptr(off), P, _lib.ptr(point_patch), None, None, 0, 2, _lib.ptr(t(np.array([1, 0], dtype=np.int32)).to(dev))]
    tail_args = [_lib.ptr(dE), None, 2] + ([] if dtype == torch.float64 else [1, None, 0]) + [_lib.current_stream()]
    assert fn(*args, -1.0, *tail_args) != 0 and b"patch_order" in lib.dnp_last_error()
    assert fn(*args, 1e-5, *tail_args) == 0
    assert torch.equal(dE, fu._patch_slabs(swork, off, None, point_patch, 0, 2, 1e-5, None, None, None, 1))


def test_source_split_does_not_change_a_bit(dev):
    """dnp_patch_fields_tiled_f32's source_split = -k (ONE launch whose last k patches are split items: four wavefronts
    on one target tile, one 128-source run of the patch each, the run terms through the exchange buffer, added in run
    order by whichever wavefront arrives last) against 1: slabs and interaction partials bit-identical - on the headline
    cloud's patches (all of 129..512 points) and on a cut with patches outside that window (<= 128 points: a single run;
    > 512: one wavefront evaluates them whatever the split) - and the all-split slabs agree with the fp64 oracle.  Every
    launch reuses ONE exchange buffer (its counters re-arm themselves, a record's place does not depend on the launch's
    size), and the outputs are poisoned first: a launch that never finished an item cannot pass on stale rows."""
    from tools.workloads import headline_workload
    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr([p.to(dev) for p in patches], dev)
    swork = pc.to(dev)[idx].contiguous()
    N, P = swork.shape[0], len(sizes)
    assert sizes.min() > 128 and sizes.max() <= 512
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(swork, off, None), fu._TileTables(swork, sizes)
    assert fu._pick_source_split(sizes[:16], N) == -3 and fu._pick_source_split(sizes[:32], N) == -3
    assert fu._pick_source_split(sizes[:128], N) == -3 and fu._pick_source_split(sizes, N) == 1
    res = {}

    def poison(shape):       # _patch_slabs returns torch.empty memory: make sure a launch that wrote nothing cannot pass on stale rows
        torch.full(shape, float("nan"), dtype=torch.float32, device=dev)

    for ss in (1, -1, -3, -7, -23, -24, -100, -2, -3):    # k >= 24: every patch of the range split
        wp = torch.full((24, tiles.n_tiles, 2), float("nan"), dtype=torch.float64, device=dev)
        poison((24, N, 3))
        res[ss] = (fu._patch_slabs(swork, off, None, point_patch, 40, 64, 1e-5, boxes, tiles.boxes, wp, ss), wp)
        assert torch.equal(res[1][0], res[ss][0]) and torch.equal(res[1][1], res[ss][1]), ss
        poison((24, N, 3))
        assert torch.equal(fu._patch_slabs(swork, off, None, point_patch, 40, 64, 1e-5, boxes, tiles.boxes, None, ss), res[1][0]), ss
    # the exchange buffer is left re-armed: every arrival counter zero again
    xch = fu._exchange(1, dev)
    item = int(_lib.load().dnp_patch_exchange_bytes(N, 1)) // tiles.n_tiles
    assert int(xch[: 24 * tiles.n_tiles * item].view(-1, item)[:, :128].sum().item()) == 0
    # the raw C ABI with a caller's own buffer: dirty memory, zeroed by dnp_exchange_init, then two launches of different sizes
    lib = _lib.load()
    nb = int(lib.dnp_patch_exchange_bytes(N, 5))
    own = torch.full((nb,), 0xa5, dtype=torch.uint8, device=dev)
    assert lib.dnp_exchange_init(_lib.ptr(own), nb, _lib.current_stream()) == 0
    for k in (5, 2):
        dEr = torch.full((24, N, 3), float("nan"), dtype=torch.float32, device=dev)
        rc = lib.dnp_patch_fields_tiled_f32(_lib.ptr(swork), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch), _lib.ptr(boxes),
                                            _lib.ptr(tiles.boxes), 40, 64, 1e-5, _lib.ptr(dEr), None, 2, -k, _lib.ptr(own), nb,
                                            _lib.current_stream())
        assert rc == 0 and torch.equal(dEr, res[1][0]), k
    # a split launch without the buffer is refused, not run wrong
    dE = torch.empty((24, N, 3), dtype=torch.float32, device=dev)
    rc = _lib.load().dnp_patch_fields_tiled_f32(_lib.ptr(swork), N, 6, _lib.ptr(off), None, P, _lib.ptr(point_patch), _lib.ptr(boxes),
                                                _lib.ptr(tiles.boxes), 40, 64, 1e-5, _lib.ptr(dE), None, 2, -3, None, 0,
                                                _lib.current_stream())
    assert rc == -3 and b"exchange buffer" in _lib.load().dnp_last_error()
    k = 51
    others = (point_patch != k).cpu()
    rows = torch.nonzero(others).flatten()[::53]
    lo, hi = int(off[k]), int(off[k + 1])
    ref = c_oracle.field_grad_f64(swork[lo:hi].cpu().numpy(), swork.cpu()[rows].numpy())
    assert rel_rowwise(res[-100][0][k - 40].cpu()[rows], ref) < TOL
    # patches outside the 2..4-run window, and a ragged last tile
    sizes2 = np.array([64, 128, 129, 300, 512, 513, 900, 41, 390, 2000], dtype=np.int64)
    off2 = t(np.concatenate([[0], np.cumsum(sizes2)])).to(dev)
    n2 = int(sizes2.sum()) + 77                           # 77 rows in no patch at the end
    sw2 = swork[:n2].contiguous()
    pp2 = torch.cat([torch.repeat_interleave(torch.arange(len(sizes2), device=dev), off2[1:] - off2[:-1]),
                     torch.full((77,), -1, dtype=torch.int64, device=dev)])
    boxes2, tiles2 = fu._patch_boxes(sw2, off2, None), fu._TileTables(sw2, sizes2)
    a = fu._patch_slabs(sw2, off2, None, pp2, 0, len(sizes2), 1e-5, boxes2, tiles2.boxes, None, 1)
    for tail in (-1, -2, -4, -9, -10, -50, -3):
        poison(tuple(a.shape))
        b = fu._patch_slabs(sw2, off2, None, pp2, 0, len(sizes2), 1e-5, boxes2, tiles2.boxes, None, tail)
        assert torch.equal(a, b), tail
    # ... with the interaction partials in their three-slot form (round 5): a cut whose 50- / 30-point patches put three groups
    # into one 128-row tile - slabs and partials of the split launches equal the plain launch's, W from them equals K3's
    sizes3 = np.array([200, 50, 30, 300, 512, 513, 100, 100, 390], dtype=np.int64)
    off3 = t(np.concatenate([[0], np.cumsum(sizes3)])).to(dev)
    n3 = int(sizes3.sum())
    sw3 = swork[:n3].contiguous()
    pp3 = torch.repeat_interleave(torch.arange(len(sizes3), device=dev), off3[1:] - off3[:-1])
    boxes3, tiles3 = fu._patch_boxes(sw3, off3, None), fu._TileTables(sw3, sizes3)
    assert tiles3.slots == 3
    wa = torch.full((len(sizes3), tiles3.n_tiles, 3), float("nan"), dtype=torch.float64, device=dev)
    a3 = fu._patch_slabs(sw3, off3, None, pp3, 0, len(sizes3), 1e-5, boxes3, tiles3.boxes, wa, 1)
    for tail in (-4, -50, -3):
        wb = torch.full_like(wa, float("nan"))
        b3 = fu._patch_slabs(sw3, off3, None, pp3, 0, len(sizes3), 1e-5, boxes3, tiles3.boxes, wb, tail)
        assert torch.equal(a3, b3) and torch.equal(wa, wb), tail
    W3 = fu._interaction_rows(a3, sw3, off3, None)
    _, Wt = fu._slabs_and_rows(sw3, off3, pp3, 0, len(sizes3), 1e-5, boxes3, tiles3, sizes3)
    assert float((Wt - W3).abs().max()) <= 1e-12 * float(W3.abs().max())
    for k in (1, 2, 5, 9):
        lo, hi = int(off2[k]), int(off2[k + 1])
        others = (pp2 != k).cpu()
        ref = c_oracle.field_grad_f64(sw2[lo:hi].cpu().numpy(), sw2.cpu()[others].numpy())
        assert rel_rowwise(b[k].cpu()[others], ref) < TOL


def test_split_tail_on_two_streams_and_two_threads_at_once(dev):
    """The exchange buffer serves ONE stream at a time; field_utils keeps one per (thread, device, stream).  Two Python threads,
    each on a stream of its own, run split-tail launches of different sizes at the same time (the reference calls the drivers
    from threads, util.py:187-196): every result equals the plain launch's, bit for bit."""
    import threading
    from tools.workloads import headline_workload
    pc, patches, _ = headline_workload()
    off, idx, sizes = util.patch_csr([p.to(dev) for p in patches], dev)
    swork = pc.to(dev)[idx].contiguous()
    P = len(sizes)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes, tiles = fu._patch_boxes(swork, off, None), fu._TileTables(swork, sizes)
    want = {(lo, hi): fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, None, 1)
            for lo, hi in ((0, 12), (100, 124))}
    torch.cuda.synchronize()
    errors = []

    def worker(lo, hi, k):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(6):
                    got = fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes, tiles.boxes, None, -k)
                    if not torch.equal(got, want[(lo, hi)]):
                        errors.append((lo, hi, k))
            st.synchronize()
        except Exception as exc:                     # pragma: no cover - reported below
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=a) for a in ((0, 12, 3), (100, 124, 5))]
    [t_.start() for t_ in threads]
    [t_.join() for t_ in threads]
    assert not errors, errors


@pytest.mark.parametrize("eps", [1e-40, 1e-33, 1e-30, 3e-12])
def test_far_chain_is_safe_for_tiny_eps(dev, eps):
    """Round-2 advisor finding: with a denormal / tiny eps the far-field threshold (eps / 4e-3)^(2/3) admitted pairs whose
    1/|r|^3 overflows fp32 and the one-transcendental chain returned NaN -> a silently zeroed slab row, where the exact
    chain (and the reference) return a finite field.  Such eps now keep the exact chain (far_threshold_d2); checked on
    the two launchers that enable the far path - the boxed patch slabs and a >= 10^9-pair field_grad - against fp64.
    The cloud has near-coincident neighbours (spacing 1e-6..1e-3) next to ordinary ones."""
    gen = torch.Generator().manual_seed(11)
    base = torch.rand(4096, 3, generator=gen) - 0.5
    near = base[:2048] + 10.0 ** (-6 + 3 * torch.rand(2048, 1, generator=gen)) * torch.randn(2048, 3, generator=gen)
    xyz = torch.cat([base, near])
    order = torch.argsort(torch.floor((xyz[:, 0] + 0.5) * 8) * 64 + torch.floor((xyz[:, 1] + 0.5) * 8) * 8 +
                          torch.floor((xyz[:, 2] + 0.5) * 8))                       # 512 spatial cells = patches
    xyz = xyz[order]
    nrm = torch.randn(xyz.shape[0], 3, generator=gen)
    cloud = torch.cat([xyz, nrm / nrm.norm(dim=1, keepdim=True)], 1).contiguous()
    N = cloud.shape[0]
    P = 48
    off_np = np.linspace(0, N, P + 1).astype(np.int64)
    off = t(off_np).to(dev)
    swork = cloud.to(dev)
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes = fu._patch_boxes(swork, off, None)
    dE = fu._patch_slabs(swork, off, None, point_patch, 0, P, eps, boxes)
    assert bool(torch.isfinite(dE).all())
    for k in (0, 23, P - 1):
        others = (point_patch != k).cpu()
        ref = c_oracle.field_grad_f64(cloud[off_np[k]:off_np[k + 1]].numpy(), cloud[others].numpy(), eps=eps)
        got = dE[k].cpu()[others]
        assert int((got.abs().sum(-1) == 0).sum()) == 0, "a finite field was zeroed"
        assert rel_rowwise(got, ref) < 1e-4          # near-coincident pairs: 1/|r|^3 ~ 1e21, fp32 cancellation rows
    # the generic entry point from 10^9 pairs on (32 768 x 32 768 replicas of the cloud, jittered)
    big = torch.cat([cloud] * 6)[:32768].clone()
    big[:, :3] += 1e-4 * torch.randn(32768, 3, generator=gen)
    E = fu.field_grad(big.to(dev), big.to(dev), eps=eps).cpu()
    rows = torch.arange(0, 32768, 257)
    ref = c_oracle.field_grad_f64(big.numpy(), big[rows].numpy(), eps=eps)
    assert bool(torch.isfinite(E).all()) and int((E[rows].abs().sum(-1) == 0).sum()) == 0
    assert rel_rowwise(E[rows], ref) < 1e-4


def test_patch_fields_interactions_and_combine(dev):
    g = load_golden("G6_patch_propagation")
    pts = t(g["pc_patchflip"]).to(dev)
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    N, P = pts.shape[0], len(allp)
    off, idx, sizes = util.patch_csr(allp, dev)
    point_patch = fu._point_patch_ids(idx, sizes, N)
    dE = fu._patch_slabs(pts, off, idx, point_patch, 0, P, 1e-5)
    assert dE.shape == (P, N, 3)
    cpu = pts.cpu()
    for k in (0, 17, P - 1):
        others = torch.ones(N, dtype=torch.bool)
        others[allp[k]] = False
        ref = c_oracle.field_grad_f64(cpu[allp[k]].numpy(), cpu[others].numpy())
        assert rel_rowwise(dE[k].cpu()[others], ref) < TOL
        assert float(dE[k][allp[k].to(dev)].abs().max()) == 0       # rows of the source patch itself are 0
    # a sub-range of patches gives the same slabs (the multi-GPU shard unit)
    part = fu._patch_slabs(pts, off, idx, point_patch, 10, 20, 1e-5)
    assert torch.equal(part, dE[10:20])
    # sum of all slabs + own-patch fields == all-pairs field (size-independent property)
    total = dE.double().sum(dim=0)
    own = torch.zeros(N, 3, dtype=torch.float64, device=dev)
    for p in allp:
        pd = p.to(dev)
        own[pd] = fu.field_grad(pts[pd], pts[pd]).double()
    covered = point_patch >= 0
    full = fu.field_grad(pts[covered], pts).double()
    assert rel_rowwise((total + own).cpu(), full.cpu()) < TOL
    # W[k][j] = sum_{t in j} dE[k][t] . n_t
    W = fu._interaction_rows(dE, pts, off, idx).cpu().numpy()
    dots = (dE.double() * pts[None, :, 3:].double()).sum(-1).cpu().numpy()
    Wref = np.stack([[dots[k, allp[j].numpy()].sum() for j in range(P)] for k in range(P)])
    assert np.abs(W - Wref).max() <= 1e-6 * np.abs(Wref).max()
    # ordered +-1 combination equals the sequential fp32 sum
    order = torch.randperm(P, generator=torch.Generator().manual_seed(1))
    coef = (torch.randint(0, 2, (P,), generator=torch.Generator().manual_seed(2)) * 2 - 1).float()
    E = torch.zeros(N, 3, device=dev)
    fu._combine(dE, coef.to(dev), order.to(dev), E, False)
    ref = torch.zeros(N, 3, device=dev)
    for i in range(P):
        ref = ref + coef[i].item() * dE[order[i]]
    assert torch.equal(E, ref)
    # signed fp64 combination (what the batched drivers use): any split of the slabs over "ranks" adds up
    sig = coef.double().to(dev)
    E64 = torch.empty(N, 3, dtype=torch.float64, device=dev)
    fu._combine_signed(dE, sig, 0, E64, False)
    want = (dE.double() * sig[:, None, None]).sum(dim=0)
    assert float((E64 - want).abs().max()) <= 1e-12 * float(want.abs().max())
    parts = torch.zeros(N, 3, dtype=torch.float64, device=dev)
    for lo, hi in ((0, 20), (20, 21), (21, P)):
        fu._combine_signed(dE[lo:hi], sig, lo, parts, True)
    assert torch.equal(parts.float(), E64.float())


# ---- greedy drivers against the reference's traces -----------------------------------------------------
def _patch_case(g, tag):
    cname, dflag, wflag = tag.split("_")
    cloud = t(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
    w = t(g["weights"]) if wflag == "w" else None
    return cloud, patches, allp, dflag == "d", w


ALL_G6 = [f"{c}_{d}_{w}" for c in ("pf", "sc") for d in ("n", "d") for w in ("nw", "w")]


@pytest.mark.parametrize("mode", ["batched", "sequential"])
@pytest.mark.parametrize("tag", ALL_G6)
def test_G6_patch_propagation(dev, tag, mode, monkeypatch):
    """BASELINE config 2 driver: fandisk, 72 patches (n_part 30, min 100), every diffuse/weights
    combination: start patch (the driver's own choice - nothing is pinned), visit order, flip decisions,
    chosen interactions and final signs of the reference."""
    g = load_golden("G6_patch_propagation")
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    monkeypatch.setattr(pd, "PATCH_MODE", mode)
    pts = cloud.clone().to(dev)
    allp_dev = [p.to(dev) for p in allp]
    if tag.startswith("pf"):      # the filtered lists ARE the patch objects (what the callers pass): fused tail kernel
        filt = [(i, allp_dev[i]) for i, _ in patches]
    else:                         # separate index tensors: the general (torch) tail
        filt = [(i, p.to(dev)) for i, p in patches]
    ret = fu.strongest_field_propagation(pts, filt, allp_dev, diffuse=diffuse, weights=None if w is None else w.to(dev))
    assert ret is None                                               # in place, returns None
    tr = fu.last_trace("patches")
    assert tr["start"] == int(g[f"order_{tag}"][0])
    assert np.array_equal(tr["order"], g[f"order_{tag}"])
    assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g[f"flipped_{tag}"])
    check_chosen(tr["chosen"], g[f"chosen_{tag}"], f"G6 {tag} {mode}")
    out = pts.cpu()
    assert np.array_equal(((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])
    assert np.abs(out[:, 3:].numpy() - g[f"normals_{tag}"]).max() < 1e-6
    assert torch.equal(out[:, :3], cloud[:, :3])


@pytest.mark.parametrize("tag", ["pf_d_w", "sc_d_nw", "pf_n_nw"])
def test_G6_slabs_larger_than_the_memory_budget(dev, tag, monkeypatch):
    """The batched driver when the [P, N, 3] slabs do not fit the device budget: blocks of patches, as many blocks
    kept as fit, the others evaluated a second time for the diffuse combine - same trace and signs as the reference
    (here: 72 fandisk patches in blocks of 12 with room for two blocks at a time)."""
    g = load_golden("G6_patch_propagation")
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    per_slab = cloud.shape[0] * 12
    monkeypatch.setattr(pd, "PATCH_MODE", "batched")
    monkeypatch.setattr(pd, "SLAB_BUDGET_BYTES", 20 * per_slab)
    monkeypatch.setattr(pd, "SLAB_FREE_CHECK_BYTES", 0)
    monkeypatch.setattr(pd, "_free_device_bytes", lambda dev: int(24.5 * per_slab / 0.8))     # budget 24 slabs: blocks of 12
    calls = []
    real = fu._patch_slabs
    monkeypatch.setattr(pd, "_patch_slabs", lambda *a, **k: (calls.append((a[4], a[5])), real(*a, **k))[1])
    pts = cloud.clone().to(dev)
    allp_dev = [p.to(dev) for p in allp]
    filt = [(i, allp_dev[i]) for i, _ in patches]
    fu.strongest_field_propagation(pts, filt, allp_dev, diffuse=diffuse, weights=None if w is None else w.to(dev))
    first_pass = [(b, min(b + 12, 72)) for b in range(0, 72, 12)]
    if diffuse:                          # the first block and the last (no working block after it) were kept
        assert calls == first_pass + first_pass[1:-1]
    else:
        assert calls == first_pass
    tr = fu.last_trace("patches")
    assert np.array_equal(tr["order"], g[f"order_{tag}"])
    assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g[f"flipped_{tag}"])
    out = pts.cpu()
    assert np.array_equal(((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])
    assert np.abs(out[:, 3:].numpy() - g[f"normals_{tag}"]).max() < 1e-6


def test_G6_slabs_within_the_budget_but_not_within_free_memory(dev, monkeypatch):
    """A GPU shared by several ranks (or a smaller part): the slabs are inside SLAB_BUDGET_BYTES but the device has
    less than that free - the driver must take the blocked two-pass path instead of one allocation that cannot
    succeed (round-2 advisor finding)."""
    g = load_golden("G6_patch_propagation")
    tag = "sc_d_nw"
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    per_slab = cloud.shape[0] * 12
    monkeypatch.setattr(pd, "PATCH_MODE", "batched")
    monkeypatch.setattr(pd, "SLAB_FREE_CHECK_BYTES", 0)
    monkeypatch.setattr(pd, "_free_device_bytes", lambda dev: int(24.5 * per_slab / 0.8))     # 72 slabs wanted, 24 fit
    calls = []
    real = fu._patch_slabs
    monkeypatch.setattr(pd, "_patch_slabs", lambda *a, **k: (calls.append((a[4], a[5])), real(*a, **k))[1])
    pts = cloud.clone().to(dev)
    allp_dev = [p.to(dev) for p in allp]
    fu.strongest_field_propagation(pts, [(i, allp_dev[i]) for i, _ in patches], allp_dev, diffuse=diffuse)
    assert len(calls) > 1 and max(b - a for a, b in calls) <= 12
    tr = fu.last_trace("patches")
    assert np.array_equal(tr["order"], g[f"order_{tag}"])
    assert np.array_equal(((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])


def test_start_patch_out_of_range_is_an_error(dev):
    """A Python-int start patch outside [0, P) raises like the host loop would (round-2 advisor finding: the device
    greedy kernels clamp, which produced a valid-looking but different propagation)."""
    g = load_golden("G6_patch_propagation")
    cloud, patches, allp, diffuse, w = _patch_case(g, "pf_n_nw")
    pts = cloud.clone().to(dev)
    allp_dev = [p.to(dev) for p in allp]
    for bad in (-1, len(allp), 10 ** 6):
        with pytest.raises(IndexError):
            fu.strongest_field_propagation(pts, [(i, allp_dev[i]) for i, _ in patches], allp_dev, start_patch=bad)
    assert torch.equal(pts.cpu(), cloud)


@pytest.mark.parametrize("tag", ["pf_d_nw", "sc_n_w"])
def test_G6_cpu_tensor_input_default_start(dev, tag):
    """CPU tensors in (like the reference's CPU run): staged to the device, same start patch, trace and signs."""
    g = load_golden("G6_patch_propagation")
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    pts = cloud.clone()
    fu.strongest_field_propagation(pts, patches, allp, diffuse=diffuse, weights=w)
    tr = fu.last_trace("patches")
    assert tr["start"] == int(g[f"order_{tag}"][0])
    assert np.array_equal(tr["order"], g[f"order_{tag}"])
    assert pts.device.type == "cpu"
    assert np.array_equal(((pts[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])
    # the same normals as a device run, bit for bit; xyz untouched; an fp64 host tensor and a strided host view likewise
    ref = cloud.clone().to(dev)
    allp_d = [p.to(dev) for p in allp]
    fu.strongest_field_propagation(ref, [(i, allp_d[i]) for i, _ in patches], allp_d, diffuse=diffuse,
                                   weights=None if w is None else w.to(dev))
    assert torch.equal(pts, ref.cpu())
    # a float64 strided host view: propagated in float64 since round 5 (the reference computes in the tensor's dtype) - the same
    # trace and signs, normals +-1 times the input up to the weight round trip n w / w in fp64, the padding columns untouched
    wide = torch.zeros(cloud.shape[0], 8, dtype=torch.float64)
    wide[:, 1:7] = cloud.double()
    fu.strongest_field_propagation(wide[:, 1:7], patches, allp, diffuse=diffuse, weights=w)
    assert np.array_equal(fu.last_trace("patches")["order"], g[f"order_{tag}"])
    assert torch.equal(wide[:, 1:4], cloud[:, :3].double())
    assert torch.equal((wide[:, 4:7] * cloud[:, 3:].double()).sum(-1) > 0, (ref.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0)
    assert float((wide[:, 4:7].abs() - cloud[:, 3:].double().abs()).abs().max()) <= (0.0 if w is None else 4e-16)
    assert float(wide[:, 0].abs().max()) == 0 and float(wide[:, 7].abs().max()) == 0


def test_start_patch_rule_matches_the_reference_curvatures(dev):
    """The device PCA behind the start-patch rule against the reference's own per-patch |lambda_min| (G6 `curv`,
    util.pca_eigen_values in fp32): same values up to fp32 eigen noise, same argmin; device and CPU paths of
    util.patch_pca agree to fp64 rounding; the kernel is deterministic."""
    g = load_golden("G6_patch_propagation")
    cloud = t(g["pc_patchflip"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    m_d, e_d, v_d, _ = util.patch_pca(cloud.to(dev), [p.to(dev) for p in allp])
    m_c, e_c, v_c, _ = util.patch_pca(cloud, allp)
    assert e_d.dtype == torch.float64 and e_d.shape == (72, 3)
    scale = float(e_c.abs().max())
    assert float((e_d.cpu() - e_c).abs().max()) < 1e-13 * scale
    assert float((m_d.cpu() - m_c).abs().max()) < 1e-14
    assert float((v_d.cpu() - v_c).abs().max()) < 1e-6              # same sign convention on both paths
    assert np.abs(e_d[:, 0].cpu().numpy() - g["curv"]).max() < 2e-6 * scale
    assert int(torch.argmin(e_d[:, 0].abs())) == int(np.argmin(np.abs(g["curv"]))) == int(g["order_pf_n_nw"][0])
    again = util.patch_pca(cloud.to(dev), [p.to(dev) for p in allp])
    assert torch.equal(again[1], e_d) and torch.equal(again[2], v_d)
    # eigen-decomposition is one: cov v = lambda v, orthonormal columns
    for k in (0, 17, 71):
        x = cloud[allp[k], :3].double()
        cov = (x - x.mean(0)).T @ (x - x.mean(0)) / x.shape[0]
        V, L = v_d[k].cpu(), e_d[k].cpu()
        assert float((cov @ V - V * L[None, :]).abs().max()) < 1e-12 * scale
        assert float((V.T @ V - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-12


def test_patch_greedy_kernel_equals_the_host_loop(dev):
    """dnp_patch_greedy against greedy_order_from_interactions on the real W of G6 and on random matrices of
    every size class (<= 256, 512, 1024, 2048, 4096 patches: one wavefront; 4097..8192 and ..16384: the one-workgroup
    kernel), ties included."""
    rng = np.random.default_rng(3)
    mats = []
    g = load_golden("G6_patch_propagation")
    pts = t(g["pc_patchflip"]).to(dev)
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    off, idx, sizes = util.patch_csr(allp, dev)
    dE = fu._patch_slabs(pts, off, idx, fu._point_patch_ids(idx, sizes, pts.shape[0]), 0, 72, 1e-5)
    mats.append((fu._interaction_rows(dE, pts, off, idx).cpu().numpy(), 42))
    for P, start in ((1, 0), (2, 1), (65, 64), (300, 7), (700, 699), (1500, 3), (2500, 11), (4100, 4099), (9001, 17)):
        mats.append((rng.standard_normal((P, P)) * np.exp(rng.standard_normal((P, 1)) * 3), start))
    tie = np.round(rng.standard_normal((130, 130)) * 2)              # many exact ties: first maximum must win
    mats.append((tie, 5))
    mats.append((np.round(rng.standard_normal((4500, 4500)) * 2), 77))   # the same through the one-workgroup kernel
    assert _lib.require_device().dnp_patch_greedy_max_patches() == 16384
    for W, start in mats:
        o_ref, s_ref, c_ref = fu.greedy_order_from_interactions(W, start)
        o, sg, c = fu._greedy_on_device(torch.from_numpy(W).to(dev), torch.tensor([start], device=dev))
        assert np.array_equal(o.cpu().numpy(), o_ref), W.shape
        assert np.array_equal(sg.cpu().numpy(), s_ref)
        assert np.array_equal(c.cpu().numpy(), c_ref)                # same fp64 arithmetic, bit for bit


@pytest.mark.parametrize("tag", ["500_n", "500_d", "50_n", "50_d"])
def test_G7_reps_propagation(dev, tag):
    g = load_golden("G7_reps_propagation")
    cap, dflag = tag.split("_")
    cloud = t(g["pc_patchflip"])
    reps = list(zip(csr_to_list(g[f"rep_off_{cap}"], g[f"rep_idx_{cap}"]),
                    csr_to_list(g[f"rest_off_{cap}"], g[f"rest_idx_{cap}"])))
    pts = cloud.clone().to(dev)
    fu.strongest_field_propagation_reps(pts, [(a.to(dev), b.to(dev)) for a, b in reps], diffuse=(dflag == "d"))
    tr = fu.last_trace("reps")
    assert tr["start"] == int(g[f"order_{tag}"][0])                  # the driver's own choice, nothing pinned
    assert np.array_equal(tr["order"], g[f"order_{tag}"])
    assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g[f"flipped_{tag}"])
    check_chosen(tr["chosen"], g[f"chosen_{tag}"], f"G7 {tag}")
    out = pts.cpu()
    assert np.array_equal(((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])


@pytest.mark.parametrize("tag", ["sub1000_n", "sub1000_d", "full_n", "full_d"])
def test_G8_point_propagation(dev, tag, monkeypatch):
    """BASELINE config 1: ok.xyz per-point propagation (10 000 points) and a 1000-point subsample:
    the complete visit order and every final sign of the reference."""
    g = load_golden("G8_point_propagation")
    name, dflag = tag.split("_")
    if tag == "full_n":
        monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", 1)              # the single-workgroup form on the full cloud
    cloud = t(g[f"pc_{name}"])
    pts = cloud.clone().to(dev)
    ret = fu.strongest_field_propagation_points(pts, diffuse=(dflag == "d"), starting_point=0)
    assert ret.data_ptr() == pts.data_ptr()
    order = fu.last_trace("points")["order"]
    ref_order = g[f"order_{tag}"]
    assert sorted(order.tolist()) == list(range(cloud.shape[0]))
    first_diff = int(np.argmax(order != ref_order)) if (order != ref_order).any() else -1
    assert first_diff == -1, f"visit order diverges from the reference at step {first_diff}"
    out = pts.cpu()
    assert np.array_equal(((out[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])


@pytest.mark.parametrize("groups", [256, 37])
def test_G8_multi_workgroup_form_reproduces_the_reference_order(dev, groups, monkeypatch):
    """The multi-workgroup persistent form (chosen above 2048 points), pinned explicitly on ok.xyz: same 10 000-step
    visit order and signs as the reference, for two different workgroup counts."""
    monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", 2)
    monkeypatch.setattr(ptd, "POINT_GREEDY_GROUPS", groups)
    g = load_golden("G8_point_propagation")
    cloud = t(g["pc_full"])
    pts = cloud.clone().to(dev)
    fu.strongest_field_propagation_points(pts, diffuse=True, starting_point=0)
    order = fu.last_trace("points")["order"]
    assert np.array_equal(order, g["order_full_d"])
    assert np.array_equal(((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g["sign_full_d"])


def test_point_propagation_beyond_single_workgroup_capacity(dev):
    """20 000 points (> 12 288): the multi-workgroup form is selected automatically; a sphere with 30 % of
    its normals flipped comes back consistently oriented, and two runs agree bit for bit."""
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(20000, 3, generator=gen)
    n = x / x.norm(dim=-1, keepdim=True)
    pc = torch.cat([n * 0.5, n], dim=1)
    flip = torch.rand(20000, generator=gen) < 0.3
    scr = pc.clone()
    scr[flip, 3:] *= -1
    a = scr.clone().to(dev)
    fu.strongest_field_propagation_points(a, diffuse=True)
    oa = torch.from_numpy(fu.last_trace("points")["order"])
    assert sorted(oa.tolist()) == list(range(20000)) and int(oa[0]) == 0
    agree = ((a.cpu()[:, 3:] * pc[:, 3:]).sum(-1) > 0).float().mean().item()
    assert agree in (0.0, 1.0)
    b = scr.clone().to(dev)
    fu.strongest_field_propagation_points(b, diffuse=True)
    assert torch.equal(a, b) and torch.equal(oa, torch.from_numpy(fu.last_trace("points")["order"]))


def test_point_propagation_stepwise_fallback_matches_kernel(dev):
    g = load_golden("G8_point_propagation")
    cloud = t(g["pc_sub1000"])[:300].clone()
    a = cloud.clone().to(dev)
    fu.strongest_field_propagation_points(a, diffuse=True)
    oa = fu.last_trace("points")["order"]
    b = cloud.clone().to(dev)
    ob = fu._points_stepwise(b, True, 0).cpu().numpy()
    _, oc = O.strongest_field_propagation_points(cloud, diffuse=True)
    assert np.array_equal(oa, oc) and np.array_equal(ob, oc)
    assert torch.equal(a.cpu()[:, 3:], b.cpu()[:, 3:])


def test_capacity_cliff_fallbacks_reproduce_the_reference_traces(dev, monkeypatch):
    """The two capacity cliffs of the device-side drivers (field_utils.PATCH_GREEDY_MAX / POINT_GREEDY_MAX_PER_GROUP):
    beyond 16 384 patches the greedy loop runs on the host, beyond 2^20 points (or the per-CU register capacity) the
    per-point driver launches step by step.  Both ways around are forced here on golden cases and must reproduce the
    reference's complete traces: G6 (72 patches: order, flips, chosen, signs) and G8 (1000 points: the 1000-step order
    and signs)."""
    g = load_golden("G6_patch_propagation")
    tag = "sc_d_w"
    cloud, patches, allp, diffuse, w = _patch_case(g, tag)
    monkeypatch.setattr(pd, "PATCH_GREEDY_MAX", 0)                    # as if P exceeded the kernels' 16 384 patches
    pts = cloud.clone().to(dev)
    allp_dev = [p.to(dev) for p in allp]
    fu.strongest_field_propagation(pts, [(i, allp_dev[i]) for i, _ in patches], allp_dev, diffuse=diffuse, weights=w.to(dev))
    tr = fu.last_trace("patches")
    assert tr["start"] == int(g[f"order_{tag}"][0]) and np.array_equal(tr["order"], g[f"order_{tag}"])
    assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g[f"flipped_{tag}"])
    check_chosen(tr["chosen"], g[f"chosen_{tag}"], f"G6 {tag} host greedy loop")
    assert np.array_equal(((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])
    monkeypatch.undo()

    g8 = load_golden("G8_point_propagation")
    monkeypatch.setattr(ptd, "POINT_GREEDY_MAX_PER_GROUP", {torch.float32: 0, torch.float64: 0})   # as if N were too large
    for dflag in ("n", "d"):
        cloud = t(g8["pc_sub1000"])
        pts = cloud.clone().to(dev)
        fu.strongest_field_propagation_points(pts, diffuse=(dflag == "d"), starting_point=0)
        assert np.array_equal(fu.last_trace("points")["order"], g8[f"order_sub1000_{dflag}"])
        assert np.array_equal(((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g8[f"sign_sub1000_{dflag}"])


# ---- BASELINE sizes: size-independent properties -------------------------------------------------------
def test_sphere100k_all_pairs_properties(dev):
    """BASELINE headline size: linearity is bit exact, a split of the sources adds up, and sampled
    rows agree with the fp64 oracle."""
    pc = sphere100k().to(dev)
    E = fu.field_grad(pc, pc)
    neg = pc.clone()
    neg[:, 3:] *= -1
    assert torch.equal(fu.field_grad(neg, pc), -E)
    half = fu.field_grad(pc[:50000], pc) .double() + fu.field_grad(pc[50000:], pc).double()
    assert rel_rowwise(half.cpu(), E.cpu()) < 2e-6
    rows = torch.arange(0, 100000, 391)
    ref = c_oracle.field_grad_f64(pc.cpu().numpy(), pc.cpu().numpy()[rows.numpy()])
    assert rel_rowwise(E.cpu()[rows], ref) < TOL
    # outward normals: the field of everybody else is aligned with the normal at every point
    assert bool(((E * pc[:, 3:]).sum(-1) > 0).all())


def test_sphere100k_256_patches_propagation_recovers_orientation(dev):
    """BASELINE config 4 on one GPU: 256 patches, whole patches flipped at random; after the
    propagation + global potential fix every normal points outward again."""
    from tools.workloads import fibonacci_patches
    pc = sphere100k()
    patches = [p.to(dev) for p in fibonacci_patches(pc, 256)]
    pc = pc.to(dev)
    assert sum(len(p) for p in patches) == 100000 and min(len(p) for p in patches) > 100
    gen = torch.Generator().manual_seed(0)
    flip = torch.rand(256, generator=gen) < 0.5
    work = pc.clone()
    for k, p in enumerate(patches):
        if flip[k]:
            work[p, 3:] *= -1
    fu.strongest_field_propagation(work, list(enumerate(patches)), patches, diffuse=True)
    if fu.measure_mean_potential(work) < 0:
        work[:, 3:] *= -1
    assert torch.equal(work[:, 3:], pc[:, 3:])
    tr = fu.last_trace("patches")
    assert sorted(tr["order"].tolist()) == list(range(256))
    s = tr["sigma"][~flip.numpy()]
    assert np.all(tr["sigma"][flip.numpy()] == -s[0]) and np.all(s == s[0])


# ---- randomized shapes / strides / gathers (launch planning and indexing) --------------------------------
def test_fuzz_shapes_strides_gathers_against_oracle(dev):
    """Random S, T (including T >= 65 536, where the 4-targets-per-lane kernel is used, and sizes straddling tile
    and chunk boundaries), wide row strides, row gathers on either operand, scattered accumulation and leaf
    limits - every result against the fp64 C oracle on a row sample."""
    lib = _lib.require_device()
    rng = np.random.default_rng(2024)
    cases = [(37, 70001), (1, 65536), (1030, 66000), (5, 131075)] + \
            [(int(rng.integers(1, 3000)), int(rng.integers(1, 5000))) for _ in range(20)]
    for case, (S, T) in enumerate(cases):
        gen = torch.Generator().manual_seed(1000 + case)
        ld_s, ld_t = int(rng.integers(6, 10)), int(rng.integers(3, 9))
        nrow_s, nrow_t = S + int(rng.integers(0, 50)), T + int(rng.integers(0, 50))
        src_all = torch.rand(nrow_s, ld_s, generator=gen) - 0.5
        tgt_all = torch.rand(nrow_t, ld_t, generator=gen) - 0.5
        use_si, use_ti = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        si = torch.randperm(nrow_s, generator=gen)[:S] if use_si else None
        ti = torch.randperm(nrow_t, generator=gen)[:T] if use_ti else None
        scatter = use_ti and bool(rng.integers(0, 2))
        accumulate = bool(rng.integers(0, 2))
        max_pts = int(rng.choice([0, 15000, 700]))
        eps = float(rng.choice([1e-5, 1e-6, 1e-3]))
        src_rows = src_all[si] if use_si else src_all[:S]
        tgt_rows = tgt_all[ti] if use_ti else tgt_all[:T]
        sample = np.unique(np.r_[0, T - 1, rng.integers(0, T, size=min(T, 48))])
        ref = c_oracle.field_grad_f64(src_rows[:, :6].numpy(), tgt_rows[sample][:, :3].numpy(), eps=eps,
                                      recursive=max_pts > 0, max_pts=max_pts if max_pts > 0 else 15000)
        out_rows = nrow_t if scatter else T
        out0 = torch.randn(out_rows, 3, generator=gen) * 10
        d_src, d_tgt, d_out = src_all.to(dev), tgt_all.to(dev), out0.to(dev)
        d_si = si.to(dev) if use_si else None
        d_ti = ti.to(dev) if use_ti else None
        nbytes = lib.dnp_field_grad_workspace_bytes(S, T, max_pts)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.dnp_field_grad_f32(_lib.ptr(d_src), S, ld_s, _lib.ptr(d_si), _lib.ptr(d_tgt), T, ld_t, _lib.ptr(d_ti),
                                    eps, max_pts, _lib.ptr(d_out), 3, int(scatter), int(accumulate), None, None, _lib.ptr(ws),
                                    nbytes, _lib.current_stream())
        assert rc == 0, (case, lib.dnp_last_error())
        torch.cuda.synchronize()
        out = d_out.cpu()
        rows_out = ti[sample] if scatter else torch.from_numpy(sample)
        got = out[rows_out].double().numpy() - (out0[rows_out].double().numpy() if accumulate else 0.0)
        err = np.linalg.norm(got - ref, axis=1)
        mag = term_magnitude(src_rows[:, :6].numpy(), tgt_rows[sample][:, :3].numpy(), eps=eps)
        slack = 1e-6 * np.abs(out0[rows_out].numpy()).max() if accumulate else 0.0      # fp32 rounding of out0 + E
        assert np.all(err <= TOL * np.linalg.norm(ref, axis=1) + 16 * 2.0 ** -24 * mag + slack), \
            f"case {case}: S={S} T={T} ld=({ld_s},{ld_t}) idx=({use_si},{use_ti}) scatter={scatter} acc={accumulate}"
        if scatter:                                           # rows that are not targets stay untouched
            mask = torch.ones(nrow_t, dtype=torch.bool)
            mask[ti] = False
            assert torch.equal(out[mask], out0[mask])


# ---- a second cloud for the greedy drivers: hand.xyz (G13) --------------------------------------------------
def test_G13_hand_point_and_patch_propagation(dev):
    g = load_golden("G13_hand")
    cloud = t(g["pc_scrambled"])
    pts = cloud.clone().to(dev)
    fu.strongest_field_propagation_points(pts, diffuse=True, starting_point=17)
    order = fu.last_trace("points")["order"]
    assert np.array_equal(order, g["order_points"])                           # all 10 000 steps
    assert np.array_equal(((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g["sign_points"])
    pc_patch = t(g["pc_patchflip"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    work = pc_patch.clone().to(dev)
    fu.strongest_field_propagation(work, [(i, p.to(dev)) for i, p in enumerate(allp)], [p.to(dev) for p in allp],
                                   diffuse=True)
    tr = fu.last_trace("patches")
    assert tr["start"] == int(g["order_patch"][0])                   # default start patch = the reference's
    assert np.array_equal(tr["order"], g["order_patch"])
    assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g["flipped_patch"])
    check_chosen(tr["chosen"], g["chosen_patch"], "G13 hand patches")
    assert np.array_equal(((work.cpu()[:, 3:] * pc_patch[:, 3:]).sum(-1) > 0).numpy(), g["sign_patch"])


# ---- float64 clouds through the per-point driver (the socket path) -----------------------------------------
@pytest.mark.parametrize("tag", ["sub1000_n", "sub1000_d", "sub3000_n", "sub3000_d"])
def test_G14_float64_cloud_point_propagation(dev, tag):
    """The reference propagates a float64 cloud in fp64 (util.npxyz2tensor feeds float64) and so does
    dnp_point_greedy_f64: the reference's fp64 visit order and signs, float64 out, and normals only ever
    multiplied by +-1 (the float64 payload stays exact)."""
    g = load_golden("G14_point_propagation_f64")
    name, dflag = tag.split("_")
    cloud = t(g[f"pc_{name}"])
    assert cloud.dtype == torch.float64
    pts = cloud.clone().to(dev)
    out = fu.strongest_field_propagation_points(pts, diffuse=(dflag == "d"), starting_point=0)
    assert out.dtype == torch.float64 and out.data_ptr() == pts.data_ptr()
    order = fu.last_trace("points")["order"]
    assert np.array_equal(order, g[f"order_{tag}"])
    res = pts.cpu()
    assert np.array_equal(((res[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g[f"sign_{tag}"])
    assert torch.equal(res[:, 3:].abs(), cloud[:, 3:].abs()) and torch.equal(res[:, :3], cloud[:, :3])


def test_point_propagation_multi_workgroup_two_points_per_lane(dev):
    """140 000 points: 256 workgroups x 512 lanes x 2 points per lane in the persistent multi-workgroup form.
    A sphere with 30 % of the normals flipped comes back consistently oriented; the visit order is a
    permutation that starts at the requested point; a second run is bit-identical."""
    gen = torch.Generator().manual_seed(33)
    x = torch.randn(140000, 3, generator=gen)
    n = x / x.norm(dim=-1, keepdim=True)
    pc = torch.cat([n * 0.5, n], dim=1)
    scr = pc.clone()
    scr[torch.rand(140000, generator=gen) < 0.3, 3:] *= -1
    a = scr.clone().to(dev)
    fu.strongest_field_propagation_points(a, diffuse=True, starting_point=4242)
    oa = torch.from_numpy(fu.last_trace("points")["order"])
    assert int(oa[0]) == 4242 and torch.equal(torch.sort(oa).values, torch.arange(140000))
    agree = ((a.cpu()[:, 3:] * pc[:, 3:]).sum(-1) > 0).float().mean().item()
    assert agree in (0.0, 1.0)
    b = scr.clone().to(dev)
    fu.strongest_field_propagation_points(b, diffuse=True, starting_point=4242)
    assert torch.equal(a, b)


@pytest.mark.parametrize("form", [0, 1, 2])
def test_G16_float64_full_ok_cloud(dev, form, monkeypatch):
    """The FULL 10 000-point ok.xyz as a float64 cloud (what the socket path feeds, util.py:71-77): all 10 000
    steps of the reference's fp64 visit order and every final sign, with the form chosen by the library (one
    workgroup per CU), with 37 workgroups, and - on the first 4096 points, its capacity in fp64 - the
    single-workgroup form against the multi-workgroup one."""
    g8, g16 = load_golden("G8_point_propagation"), load_golden("G16_point_propagation_f64_full")
    cloud = t(g8["pc_full"]).double()
    if form == 1:
        sub = cloud[:4096].clone()
        monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", 1)
        a = sub.clone().to(dev)
        fu.strongest_field_propagation_points(a, diffuse=True)
        oa = fu.last_trace("points")["order"]
        monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", 2)
        b = sub.clone().to(dev)
        fu.strongest_field_propagation_points(b, diffuse=True)
        assert np.array_equal(oa, fu.last_trace("points")["order"]) and torch.equal(a, b)
        return
    monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", form)
    if form == 2:
        monkeypatch.setattr(ptd, "POINT_GREEDY_GROUPS", 37)
    pts = cloud.clone().to(dev)
    out = fu.strongest_field_propagation_points(pts, diffuse=True, starting_point=0)
    assert out.dtype == torch.float64 and out.data_ptr() == pts.data_ptr()
    order = fu.last_trace("points")["order"]
    ref = g16["order_full_d"].astype(np.int64)
    first_diff = int(np.argmax(order != ref)) if (order != ref).any() else -1
    assert first_diff == -1, f"fp64 visit order diverges from the reference at step {first_diff}"
    res = pts.cpu()
    assert np.array_equal(((res[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy(), g16["sign_full_d"])


def test_point_greedy_never_reads_a_rewritten_normal(dev):
    """Regression for the last-step race: the kernels leave pts untouched until a stream-ordered copy, so the last
    chosen point's field enters everybody's E with the sign decided for it.  E_out of the persistent kernel must
    equal the step-wise evaluation (which has no concurrency at all) bit for bit, in both forms."""
    lib = _lib.require_device()
    g = load_golden("G8_point_propagation")
    cloud = t(g["pc_sub1000"]).to(dev)
    ref = cloud.clone()
    fu._points_stepwise(ref, True, 0)
    for form, groups in ((1, 0), (2, 0), (2, 3)):
        work = cloud.clone()
        N = work.shape[0]
        order = torch.empty(N, dtype=torch.int64, device=dev)
        nbytes = lib.dnp_point_greedy_workspace_bytes(N, 4)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.dnp_point_greedy_f32(_lib.ptr(work), N, 6, 0, 1e-6, 1, _lib.ptr(order), None, form, groups,
                                      _lib.ptr(ws), nbytes, _lib.current_stream())
        assert rc == 0, lib.dnp_last_error()
        torch.cuda.synchronize()
        assert int(ws[:4].view(torch.int32).item()) == 0
        assert torch.equal(work, ref), (form, groups)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_point_greedy_timeout_leaves_pts_untouched_and_the_fallback_equals_the_kernel(dev, dtype, monkeypatch, capsys):
    """The abort path of the multi-workgroup form (a workgroup gave up waiting for its peers), forced with the ABI's
    test hook form = 3: the status word is set, pts stays bit for bit the caller's input (the copy kernel stores
    nothing - round 2 stored a partial propagation and could hand the fallback a globally inverted start), and the
    Python driver's step-wise fallback ends exactly where the persistent kernel does."""
    lib = _lib.require_device()
    g = load_golden("G8_point_propagation")
    cloud = t(g["pc_sub1000"]).to(dev).to(dtype)
    N = cloud.shape[0]
    work = cloud.clone()
    order = torch.full((N,), -1, dtype=torch.int64, device=dev)
    nbytes = lib.dnp_point_greedy_workspace_bytes(N, work.element_size())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fn = lib.dnp_point_greedy_f64 if dtype == torch.float64 else lib.dnp_point_greedy_f32
    rc = fn(_lib.ptr(work), N, 6, 0, 1e-6, 1, _lib.ptr(order), None, 3, 4, _lib.ptr(ws), nbytes, _lib.current_stream())
    assert rc == 0, lib.dnp_last_error()
    torch.cuda.synchronize()
    assert int(ws[:4].view(torch.int32).item()) != 0, "form 3 must end in the time-out state"
    assert torch.equal(work, cloud), "an aborted launch must leave pts untouched"
    # the driver: same hook, falls back to the step-wise launches and warns
    good = cloud.clone()
    fu.strongest_field_propagation_points(good, diffuse=True)
    good_order = fu.last_trace("points")["order"]
    monkeypatch.setattr(ptd, "POINT_GREEDY_FORM", 3)
    monkeypatch.setattr(ptd, "POINT_GREEDY_GROUPS", 4)
    fell = cloud.clone()
    fu.strongest_field_propagation_points(fell, diffuse=True)
    assert "timed out" in capsys.readouterr().out
    assert np.array_equal(fu.last_trace("points")["order"], good_order)
    assert torch.equal(fell, good)


def test_nonfinite_leaf_components_are_counted_and_zeroed(dev, capsys):
    """field_utils.py:110-115: Inf/NaN leaf components are reported ("warning: %d nan in field_grad") and zeroed.
    With eps = 0 a target that coincides with a source makes that LEAF's sum 0/0 = NaN for the target's three
    components (G3 pins the case); with 15000-row leaves the other leaf of a 16000-row source set still
    contributes.  One warning, three components, the row equal to the reference-class oracle's."""
    gen = torch.Generator().manual_seed(4)
    src = torch.rand(16000, 6, generator=gen) - 0.5
    tgt = torch.rand(40, 3, generator=gen) - 0.5
    tgt[5] = src[3, :3]                                # coincides with a source of the first leaf (rows 0..7999)
    E = fu.field_grad(src.to(dev), tgt.to(dev), eps=0.0)
    ref = O.field_grad(src, tgt, eps=0.0)
    fu.flush_warnings()
    printed = capsys.readouterr().out
    assert "warning: 3 nan in field_grad" in printed and "inf in field_grad" not in printed
    assert bool(torch.isfinite(E).all())
    assert rel_rowwise(E.cpu(), ref) < 1e-4            # row 5 = the surviving leaf alone, as in the reference
    second_leaf = O.field_grad(src[8000:], tgt[5:6], eps=0.0)
    assert rel_rowwise(E.cpu()[5:6], second_leaf) < 1e-4
    fu.field_grad(src.to(dev), tgt.to(dev))            # eps > 0: the coincident pair contributes 0, no warning
    fu.flush_warnings()
    assert "field_grad" not in capsys.readouterr().out
    # the counters travel to the host in blocks of 16 calls, printed by a later call without any explicit flush;
    # slots are recycled correctly over several turns of the 64-slot ring
    small_s, small_t = src[:300].to(dev), tgt[10:18].to(dev)      # rows without a coincident source
    bad_t = small_t.clone()
    bad_t[2] = small_s[7, :3]
    expected = 0
    for k in range(200):
        if k % 37 == 5:
            fu.field_grad(small_s, bad_t, eps=0.0)
            expected += 1
        else:
            fu.field_grad(small_s, small_t, eps=0.0)
    torch.cuda.synchronize()
    for _ in range(17):                                # one more block of calls: finds every earlier block landed
        fu.field_grad(small_s, small_t, eps=0.0)
    lazily = capsys.readouterr().out.count("warning: 3 nan in field_grad")
    assert lazily == expected, (lazily, expected)
    fu.flush_warnings()
    assert "field_grad" not in capsys.readouterr().out


def test_randomised_cross_check_of_the_round3_entry_points(dev):
    """tools/gpu_fuzz.py for a dozen seconds inside the suite (the tool ran 4253 cases in 200 s without a failure when it
    was written): random clustered clouds with coincident points, random patch cuts (1 .. 700 points, empty patches, rows in
    no patch), random eps - slabs bit-identical across tile table / interaction partials / source split, W from the
    partials against K3, the fused reference_field against the two-step form, field_grad against the fp64 oracle."""
    from tools import gpu_fuzz
    cases, fails = gpu_fuzz.run(budget=12.0, seed=20251004)
    assert cases > 50 and not fails, fails[:5]
