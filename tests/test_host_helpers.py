"""Host-side logic of the drop-in layer against the reference's golden vectors (CPU only):
the util helpers on the path (GH) and the P x P greedy loop that replaces the per-step field
evaluations of the patch drivers."""
import numpy as np
import torch

from conftest import csr_to_list, load_golden
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import util
from oracle import dipole_oracle as O

t = torch.from_numpy


def test_gen_grid_matches_reference():
    assert np.array_equal(util.gen_grid().numpy(), load_golden("G4_potential")["grid"])


def test_xyz_parse():
    g = load_golden("GH_xyz_parse")
    txt = "1 2 3\n4 5 6 0 0 1\nnan 1 2\n\n7.5 8 9e-1"
    assert np.array_equal(util.xyz2tensor(txt).numpy(), g["parsed"])
    assert np.array_equal(util.xyz2tensor("1 2 3\n4 5 6", append_normals=False).numpy(), g["parsed_noappend"])


def test_export_roundtrip(tmp_path):
    pc = t(load_golden("GH_host_helpers")["pc_head"])
    util.export_pc(pc.transpose(0, 1), tmp_path / "a.xyz")
    text = open(tmp_path / "a.xyz").read()
    assert not text.endswith("\n") and len(text.split("\n")) == pc.shape[0]
    assert np.array_equal(util.xyz2tensor(text).numpy(), pc.numpy())


def test_transform_matches_reference():
    g = load_golden("GH_host_helpers")
    raw = t(load_golden("G5_fandisk_allpairs")["raw"])
    pc, tr = util.Transform.trans(raw)
    assert np.array_equal(tr.center.numpy(), g["center"]) and float(tr.scale) == float(g["scale"])
    assert np.array_equal(pc[:32].numpy(), g["pc_head"])
    assert np.array_equal(tr.inverse(pc)[:32].numpy(), g["inv_head"])


def test_divide_pc_matches_reference_partition_and_merge():
    g = load_golden("GH_host_helpers")
    pc = t(load_golden("G5_fandisk_allpairs")["pc"])
    # plain voxel partition of a small cloud: cells, order and membership
    ind, ijk = util._divide_pc(t(g["small"])[:, :3], 6)
    assert np.array_equal(np.cumsum([0] + [len(i) for i in ind]), g["small_off"])
    assert np.array_equal(torch.cat(ind).numpy(), g["small_idx"])
    assert np.array_equal(np.array(ijk), g["small_ijk"])
    # partition + merge at the fandisk.sh settings: 72 patches, identical index lists
    patches = util.divide_pc(pc[:, :3], 30, min_patch=100)
    assert len(patches) == len(g["patch_off"]) - 1 == 72
    assert np.array_equal(np.cumsum([0] + [len(p) for p in patches]), g["patch_off"])
    assert np.array_equal(torch.cat(patches).numpy(), g["patch_idx"])


def test_pca_orient_center_and_filter():
    g = load_golden("GH_host_helpers")
    pc = t(load_golden("G5_fandisk_allpairs")["pc"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])
    ev = np.array([float(util.pca_eigen_values(pc[p])[0][0]) for p in allp])
    tr = np.array([float(pc[p][:, :3].var(dim=0).sum()) for p in allp])
    assert np.abs(ev - g["eig_min"]).max() < 1e-6 * tr.max()          # fp32 eigen noise, see DESIGN.md
    assert np.array_equal(util.orient_center(t(g["oc_in"]).clone()).numpy(), g["oc_out"])
    pcf = t(g["filt_in"]).clone()
    kept = util.fix_n_filter(pcf, [p.clone() for p in allp], 0.01)
    assert np.array_equal(np.array([i for i, _ in kept]), g["filt_kept"])
    same = (np.sign((pcf.numpy()[:, 3:] * g["filt_out"][:, 3:]).sum(-1)) > 0).mean()
    assert same > 0.999                                                  # PCA normal sign is arbitrary per patch


def test_greedy_loop_on_interaction_matrix_equals_stepwise_driver():
    """I_j = sum_{k visited} sigma_k W[k,j] reproduces the oracle's step-by-step driver: same
    visit order, same flips, same chosen interactions."""
    g = load_golden("G6_patch_propagation")
    cloud = t(g["pc_patchflip"])
    allp = csr_to_list(g["patch_off"], g["patch_idx"])[:12]
    sub = torch.cat(allp)
    remap = -torch.ones(cloud.shape[0], dtype=torch.long)
    remap[sub] = torch.arange(sub.shape[0])
    pts = cloud[sub].clone()
    patches = [remap[p] for p in allp]
    P, N = len(patches), pts.shape[0]
    W = np.zeros((P, P))
    for k in range(P):
        others = torch.ones(N, dtype=torch.bool)
        others[patches[k]] = False
        dE = torch.zeros(N, 3)
        dE[others] = O.field_grad(pts[patches[k]], pts[others])
        for j in range(P):
            W[k, j] = float((dE[patches[j]].double() * pts[patches[j], 3:].double()).sum())
    for start in (0, 5):
        order, sigma, chosen = fu.greedy_order_from_interactions(W, start)
        _, trace = O.strongest_field_propagation(pts, [], patches, diffuse=True, start_patch=start)
        assert np.array_equal(order, trace["order"])
        assert np.array_equal(sigma < 0, trace["flipped"][np.argsort(trace["order"])])
        assert np.allclose(chosen, trace["chosen"], rtol=1e-4)


def test_balanced_blocks():
    sizes = np.array([100, 400, 50, 50, 300, 100])
    b = fu._balanced_blocks(sizes, 2)
    assert b[0] == 0 and b[-1] == 6 and len(b) == 3
    assert abs(sizes[b[0]:b[1]].sum() - sizes[b[1]:b[2]].sum()) <= 400
    for w in (1, 3, 8):
        bb = fu._balanced_blocks(sizes, w)
        assert len(bb) == w + 1 and np.all(np.diff(bb) >= 0) and bb[-1] == 6
