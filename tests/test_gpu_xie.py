"""The fork's "xie" pair functions through the C ABI (dnp_xie_pairs_*, dnp_xie_order_f32) against the golden
vectors captured from the reference (GX) and the oracle.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dipole_normal_prop_amd import field_utils as fu
from oracle import dipole_oracle as O

pytestmark = pytest.mark.gpu
t = torch.from_numpy


def test_xie_field_and_interaction_small(dev):
    g = load_golden("GX_xie")
    src, tgt = t(g["src"]).to(dev), t(g["tgt"]).to(dev)
    for C in (3, 2):
        f = fu.xie_field(src, tgt, eps=0.1, C=C)
        assert f.shape == (40, 50, 3) and f.device == src.device
        assert np.abs(f.cpu().numpy() - g[f"field_C{C}"]).max() / np.abs(g[f"field_C{C}"]).max() < 1e-6
        m = fu.xie_intersaction(src, tgt, eps=0.1, knn_mask=-1, C=C)
        assert m.shape == (40, 50)
        assert np.abs(m.cpu().numpy() - g[f"inter_C{C}"]).max() / np.abs(g[f"inter_C{C}"]).max() < 1e-6
    f = fu.xie_field(src, tgt, eps=0.1).cpu().numpy()
    assert np.array_equal(f[0, 10], g["src"][10, 3:]) and np.array_equal(f[4, 14], g["src"][14, 3:])   # coincident: n_s
    mk = fu.xie_intersaction(src, tgt, eps=0.1, knn_mask=5, C=3).cpu().numpy()
    assert np.array_equal(mk != 0, g["inter_knn5"] != 0)
    assert np.abs(mk - g["inter_knn5"]).max() / np.abs(g["inter_knn5"]).max() < 1e-6
    m64 = fu.xie_intersaction(src.double(), tgt.double(), eps=0.1, knn_mask=-1, C=3).cpu().numpy()
    assert np.allclose(m64, g["inter64"], rtol=1e-12, atol=1e-12)
    assert np.allclose(fu.xie_distance(src, tgt, 0.1).cpu().numpy(), g["distance"], rtol=1e-5)
    # CPU tensors are staged and come back on the CPU
    assert fu.xie_intersaction(t(g["src"]), t(g["tgt"]), 0.1, -1, 3).device.type == "cpu"


def test_xie_interaction_matrix_rows_on_ok_subsample(dev):
    g = load_golden("GX_xie")
    pc = t(g["pc"]).to(dev)
    M = fu.xie_intersaction(pc, pc, eps=0.1, knn_mask=-1, C=3)
    assert M.shape == (1000, 1000)
    ref = g["inter_pc"]
    assert np.abs(M[:64].cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-6
    assert np.array_equal(np.diag(M.cpu().numpy()), (g["pc"][:, 3:] ** 2).sum(-1).astype(np.float32))   # self pair: n.n


def _noise_level_decisions(M, order, ref_flip, got_flip):
    """Every decision where `got` differs from the reference must be one the reference itself took on rounding
    noise: with the weights of the points visited before it (the reference's), the row sum in fp64 is below
    2e-6 of the sum of the magnitudes (fp32 summation error of ~1000 terms)."""
    w = np.zeros(M.shape[0])
    for i in order:
        if got_flip[i] != ref_flip[i]:
            terms = M[i].astype(np.float64) * w
            assert abs(terms.sum()) <= 2e-6 * np.abs(terms).sum(), (i, terms.sum(), np.abs(terms).sum())
        w[i] = -1.0 if ref_flip[i] else 1.0


def _consistent_with_the_reference(M, orders, ref_n, got_n, ref_d=None, got_d=None):
    """The complete noise argument for the ordered propagation, both phases, per visiting order r:
    ordered phase - every decision that differs from the reference's is one the reference took on rounding noise
    (_noise_level_decisions, with the reference's weights);
    diffuse phase (interactions = M @ weights with the FINAL weights, field_utils.py:597-603) - every decision of the
    kernel path equals the sign of the exact (fp64) row sum with the kernel path's own weights unless that sum is itself
    noise, and wherever it differs from the reference's either the reference's own sum is noise or the exact sums with
    the two weight vectors have different signs, i.e. the difference is the consequence of an ordered-phase decision
    already shown to be noise-level.  Together: the result is the reference algorithm's, in exact arithmetic, with at
    most a few noise-level decisions fallen the other way."""
    M64 = M.astype(np.float64)
    mag = np.abs(M64).sum(axis=1)
    for r in range(len(orders)):
        _noise_level_decisions(M, orders[r], ref_n[r], got_n[r])
        if got_d is None:
            continue
        s_got = M64 @ np.where(got_n[r], -1.0, 1.0)
        s_ref = M64 @ np.where(ref_n[r], -1.0, 1.0)
        assert np.all((got_d[r] == (s_got < 0)) | (np.abs(s_got) <= 2e-6 * mag)), r
        for i in np.nonzero(got_d[r] != ref_d[r])[0]:
            assert abs(s_ref[i]) <= 2e-6 * mag[i] or (s_got[i] < 0) != (s_ref[i] < 0), (r, i, s_ref[i], s_got[i])


@pytest.mark.parametrize("tag,diffuse,knn", [("n_k0", False, -1), ("d_k0", True, -1), ("n_k20", False, 20),
                                             ("d_k20", True, 20)])
def test_xie_ordered_propagation(dev, tag, diffuse, knn):
    g = load_golden("GX_xie")
    pc = t(g["pc"]).to(dev)
    res = fu.xie_propagation_points_in_order(pc, 0.1, g["orders"], diffuse=diffuse, knn_mask=knn, C=3)
    assert res.dtype == torch.bool and res.shape == (3, 1000)
    got, ref = res.cpu().numpy(), g[f"flip_{tag}"]
    assert int((got != ref).sum()) <= 2
    # sign decisions are bit-exact except where the reference's own fp32 row sum is rounding noise around zero (the
    # kernel sums the row in fp64): every differing decision - of the ordered phase and, for the diffuse variants, of
    # the final matrix product too - is shown to be exactly such a case or the consequence of one
    M = fu.xie_intersaction(pc, pc, 0.1, knn, 3).cpu().numpy()
    ntag = tag.replace("d_", "n_")
    got_n = got if not diffuse else fu.xie_propagation_points_in_order(pc, 0.1, g["orders"], diffuse=False, knn_mask=knn,
                                                                       C=3).cpu().numpy()
    _consistent_with_the_reference(M, g["orders"], g[f"flip_{ntag}"], got_n, ref if diffuse else None,
                                   got if diffuse else None)
    assert torch.equal(pc, t(g["pc"]).to(dev))                          # input untouched


@pytest.mark.parametrize("tag,times,diffuse", [("t1_n", 1, False), ("t5_n", 5, False), ("t5_d", 5, True)])
def test_xie_bfstree_propagation_with_vote(dev, tag, times, diffuse):
    """field_utils.xie_propagation_points_onbfstree (field_utils.py:657-710): routes, per-route flips, the vote
    (the reference's MIQP solved by enumeration) and the final flips against GX2.  Per-route decisions may differ from
    the reference's only as the noise argument above allows (checked for every route, ordered and diffuse phase)."""
    g = load_golden("GX2_xie_bfstree")
    pts = t(g["pc"]).clone().to(dev)
    res = fu.xie_propagation_points_onbfstree(pts, 0.1, diffuse=diffuse, starting_point=0, k=10, treshold=0.1,
                                              times=times, knn_mask=-1, C=3)
    tr = fu.last_trace("bfstree")
    assert np.array_equal(tr["orders"], g[f"orders_{tag}"])
    assert int((tr["flips"] != g[f"flips_{tag}"]).sum()) <= 2 * times
    pc0 = t(g["pc"]).to(dev)
    M = fu.xie_intersaction(pc0, pc0, 0.1, -1, 3).cpu().numpy()
    ntag = tag.replace("_d", "_n")
    assert np.array_equal(g[f"orders_{ntag}"], g[f"orders_{tag}"])
    got_n = tr["flips"] if not diffuse else fu.xie_propagation_points_in_order(pc0, 0.1, tr["orders"], diffuse=False,
                                                                               knn_mask=-1, C=3).cpu().numpy()
    _consistent_with_the_reference(M, tr["orders"], g[f"flips_{ntag}"], got_n, g[f"flips_{tag}"] if diffuse else None,
                                   tr["flips"] if diffuse else None)
    assert np.array_equal(tr["status"], g[f"status_{tag}"])
    got = res.cpu().numpy()
    assert int((got != g[f"result_{tag}"]).sum()) <= 2
    same = got == g[f"result_{tag}"]
    assert np.array_equal(pts.cpu().numpy()[same, 3:], g[f"normals_{tag}"][same])
    assert np.array_equal(pts.cpu().numpy()[:, :3], g["pc"][:, :3])


def test_xie_pairs_ragged_against_oracle(dev):
    gen = torch.Generator().manual_seed(77)
    src = torch.randn(333, 6, generator=gen)
    tgt = torch.randn(257, 6, generator=gen)
    ref = O.xie_intersaction(src.double(), tgt.double(), C=2.5).numpy()
    m = fu.xie_intersaction(src.to(dev), tgt.to(dev), eps=0.0, knn_mask=-1, C=2.5).cpu().numpy()
    assert np.abs(m - ref).max() / np.abs(ref).max() < 1e-5


def _torch_op_by_op(src, tgt, C):
    """The pair body as separate torch kernels on the device - every operation IEEE-rounded on its own, in the kernel's
    (= the reference's, field_utils.py xie_field) order; no contraction can happen across torch kernels."""
    r = src[None, :, :3] - tgt[:, None, :3]
    rx, ry, rz = r[..., 0], r[..., 1], r[..., 2]
    nrm = torch.sqrt((rx * rx + ry * ry) + rz * rz)
    ux, uy, uz = rx / nrm, ry / nrm, rz / nrm
    nx, ny, nz = src[None, :, 3], src[None, :, 4], src[None, :, 5]
    d = C * ((nx * ux + ny * uy) + nz * uz)
    n3 = (nrm * nrm) * nrm
    zero = nrm == 0
    f = [torch.where(zero, n.expand_as(nrm), (n - d * u) / n3) for n, u in ((nx, ux), (ny, uy), (nz, uz))]
    m = (f[0] * tgt[:, None, 3] + f[1] * tgt[:, None, 4]) + f[2] * tgt[:, None, 5]
    return torch.stack(f, -1), torch.where(torch.isfinite(m), m, torch.zeros_like(m))


@pytest.mark.parametrize("scale", [1.0, 1e-3, 37.0, 1e-12, 3e11])
def test_xie_pairs_are_the_ieee_op_order_bit_for_bit(dev, scale):
    """The kernel divides by shared, refined reciprocals and takes the fp32 root directly (round 3, csrc/dnp_xie.hip) -
    claimed to be the bits of IEEE division / sqrt applied operation by operation.  Checked against exactly that: the
    same expression as separate torch kernels, on random clouds at three length scales, with coincident pairs."""
    gen = torch.Generator().manual_seed(int(scale * 1000) % 100000 + 3)         # 1e-12 / 3e11: |R|^3 denormal / overflowing
    src = torch.randn(1300, 6, generator=gen)
    tgt = torch.randn(900, 6, generator=gen)
    src[:, :3] *= scale
    tgt[:, :3] *= scale
    tgt[:50, :3] = src[100:150, :3]                                   # coincident pairs: |R| == 0
    src, tgt = src.to(dev), tgt.to(dev)
    for C in (3.0, 2.5):
        want_f, want_m = _torch_op_by_op(src, tgt, C)
        got_f = fu.xie_field(src, tgt, eps=0.1, C=C)
        got_m = fu.xie_intersaction(src, tgt, eps=0.1, knn_mask=-1, C=C)
        assert torch.equal(got_f, want_f), float((got_f - want_f).abs().max())
        assert torch.equal(got_m, want_m), float((got_m - want_m).abs().max())


def _order_kernel_spec(M, order):
    """dnp_xie_order_f32 as specified (csrc/dnp_xie.hip): fp32 products, fp64 sums - thread t of 1024 adds its columns
    t, t + 1024, ... in ascending order, each wavefront folds its 64 sums by halving (lane l += lane l + off, off = 32 ... 1),
    the 16 wavefront sums are added in wavefront order - and the sign of the rounded fp32 total becomes the weight."""
    N = M.shape[0]
    pad = -(-N // 1024) * 1024
    w = np.zeros(N, dtype=np.float32)
    inter = np.zeros(N, dtype=np.float32)
    for idx in order:
        p = np.zeros(pad, dtype=np.float64)
        p[:N] = (M[idx] * w).astype(np.float32)
        s = np.zeros(1024)
        for k in range(pad // 1024):
            s = s + p[k * 1024:(k + 1) * 1024]
        v = s.reshape(16, 64).copy()
        for off in (32, 16, 8, 4, 2, 1):
            v[:, :off] = v[:, :off] + v[:, off:2 * off]
        tot = 0.0
        for k in range(16):
            tot = tot + v[k, 0]
        inter[idx] = np.float32(tot)
        w[idx] = -1.0 if inter[idx] < 0 else 1.0
    return inter, w


@pytest.mark.parametrize("n", [700, 1500, 5000])
def test_xie_order_kernel_is_its_specification_bit_for_bit(dev, n):
    """The register-resident, row-prefetching forms of the ordered propagation (round 3: 4 and 16 columns per thread)
    against the summation order they are specified to keep - inter and weights bit for bit, two orders at once."""
    from dipole_normal_prop_amd import _lib
    lib = _lib.require_device()
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, 6, generator=gen)
    pc = torch.cat([x[:, :3], torch.nn.functional.normalize(x[:, 3:], dim=1)], 1).to(dev)
    M = fu.xie_intersaction(pc, pc, 0.1, -1, 3).contiguous()
    orders = np.stack([np.random.default_rng(s).permutation(n) for s in (1, 2)]).astype(np.int64)
    order_t = t(orders).to(dev)
    weights = torch.full((2, n), 7.0, dtype=torch.float32, device=dev)
    inter = torch.full((2, n), 7.0, dtype=torch.float32, device=dev)
    assert lib.dnp_xie_order_f32(_lib.ptr(M), n, _lib.ptr(order_t), 2, _lib.ptr(weights), _lib.ptr(inter), _lib.current_stream()) == 0
    Mh = M.cpu().numpy()
    for r in range(2):
        want_i, want_w = _order_kernel_spec(Mh, orders[r])
        assert np.array_equal(inter[r].cpu().numpy(), want_i)
        assert np.array_equal(weights[r].cpu().numpy(), want_w)
