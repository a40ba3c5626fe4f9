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


@pytest.mark.parametrize("tag,diffuse,knn", [("n_k0", False, -1), ("d_k0", True, -1), ("n_k20", False, 20),
                                             ("d_k20", True, 20)])
def test_xie_ordered_propagation(dev, tag, diffuse, knn):
    g = load_golden("GX_xie")
    pc = t(g["pc"]).to(dev)
    res = fu.xie_propagation_points_in_order(pc, 0.1, g["orders"], diffuse=diffuse, knn_mask=knn, C=3)
    assert res.dtype == torch.bool and res.shape == (3, 1000)
    got, ref = res.cpu().numpy(), g[f"flip_{tag}"]
    assert int((got != ref).sum()) <= 2
    if not diffuse and (got != ref).any():
        # sign decisions are bit-exact except where the reference's own fp32 row sum is rounding noise around zero
        # (the kernel sums the row in fp64): each differing decision is checked to be exactly such a case
        M = fu.xie_intersaction(pc, pc, 0.1, knn, 3).cpu().numpy()
        for r in range(3):
            _noise_level_decisions(M, g["orders"][r], ref[r], got[r])
    assert torch.equal(pc, t(g["pc"]).to(dev))                          # input untouched


@pytest.mark.parametrize("tag,times,diffuse", [("t1_n", 1, False), ("t5_n", 5, False), ("t5_d", 5, True)])
def test_xie_bfstree_propagation_with_vote(dev, tag, times, diffuse):
    """field_utils.xie_propagation_points_onbfstree (field_utils.py:657-710): routes, per-route flips, the vote
    (the reference's MIQP solved by enumeration) and the final flips against GX2."""
    g = load_golden("GX2_xie_bfstree")
    pts = t(g["pc"]).clone().to(dev)
    res = fu.xie_propagation_points_onbfstree(pts, 0.1, diffuse=diffuse, starting_point=0, k=10, treshold=0.1,
                                              times=times, knn_mask=-1, C=3)
    tr = fu.last_trace("bfstree")
    assert np.array_equal(tr["orders"], g[f"orders_{tag}"])
    assert int((tr["flips"] != g[f"flips_{tag}"]).sum()) <= 2 * times    # noise-level decisions, see above
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
