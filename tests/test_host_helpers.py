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
    # dropped patches are aligned with their PCA normal, whose sign is arbitrary (LAPACK's in the reference, the
    # largest component positive here): every patch agrees with the reference as a whole or is its mirror image
    agree = (pcf.numpy()[:, 3:] * g["filt_out"][:, 3:]).sum(-1) > 0
    for i, p in enumerate(allp):
        a = agree[p.numpy()]
        assert a.all() or (not a.any()), f"patch {i}"
        if i in set(g["filt_kept"].tolist()):
            assert a.all()                                                   # kept patches are left untouched


def test_native_merge_equals_the_literal_double_loop():
    """dnp_merge_cells (host C++: voxel-owner map, O(27) per small cell) against the literal O(cells^2) restatement
    of util.merge_nodes (oracle/host_oracle.py) on random cell tables: same groups, same order, same sweeps."""
    from oracle import host_oracle
    rng = np.random.default_rng(11)
    for case in range(30):
        side = int(rng.integers(2, 9))
        n_cells = int(rng.integers(1, side ** 3 + 1))
        flat = np.sort(rng.choice(side ** 3, size=n_cells, replace=False))
        ijk = np.stack([flat // (side * side), (flat // side) % side, flat % side], axis=1).astype(np.int32)
        sizes = rng.integers(1, 60, size=n_cells)
        min_patch = int(rng.choice([0, 5, 30, 80, 400]))
        seq, seq_off = util.merge_cells(ijk, sizes, min_patch)
        groups, _ = host_oracle.merge_nodes(sizes.tolist(), [tuple(r) for r in ijk.tolist()], min_patch)
        got = [seq[seq_off[g]:seq_off[g + 1]].tolist() for g in range(len(seq_off) - 1)]
        assert got == groups, f"case {case}: side {side}, {n_cells} cells, min_patch {min_patch}"


def test_divide_pc_on_boxunion_matches_the_reference_partition():
    """BASELINE config 3 stand-in: data/boxunion.xyz, number_parts 41, minimum_points_per_patch 100
    (demos/lion.sh): the reference's _divide_pc + merge_nodes gave 369 patches; identical index lists."""
    g = load_golden("G15_boxunion_config3")
    patches = util.divide_pc(t(g["pc"])[:, :3], 41, min_patch=100)
    assert isinstance(patches, util.PatchList) and len(patches) == 369
    assert np.array_equal(np.cumsum([0] + patches.sizes), g["patch_off"])
    assert np.array_equal(patches.flat.numpy(), g["patch_idx"].astype(np.int64))
    assert all(torch.equal(p, patches.flat[o:o + n]) for p, o, n in
               zip(patches, np.cumsum([0] + patches.sizes), patches.sizes))


def test_patch_pca_start_rule_on_every_golden():
    """util.patch_pca (the one shared covariance code; CPU path here) picks the reference's start patch on the
    fandisk and hand goldens; on boxunion (axis-aligned boxes) dozens of patches are EXACTLY planar and the
    reference's fp32 choice among them is decided by the rounding of its fp32 mean - the rule here takes the first
    exactly flat patch and the reference's choice lies in the same tie class (|lambda_min| below 1e-12 of the trace)."""
    g6, g7, g13, g15 = (load_golden(n) for n in ("G6_patch_propagation", "G7_reps_propagation", "G13_hand",
                                                  "G15_boxunion_config3"))
    cases = [(t(g6["pc_patchflip"]), csr_to_list(g6["patch_off"], g6["patch_idx"]), int(g6["order_pf_d_nw"][0])),
             (t(g7["pc_patchflip"]), csr_to_list(g7["rep_off_500"], g7["rep_idx_500"]), int(g7["order_500_d"][0])),
             (t(g7["pc_patchflip"]), csr_to_list(g7["rep_off_50"], g7["rep_idx_50"]), int(g7["order_50_d"][0])),
             (t(g13["pc_patchflip"]), csr_to_list(g13["patch_off"], g13["patch_idx"]), int(g13["order_patch"][0]))]
    for cloud, lists, want in cases:
        _, ev, _, _ = util.patch_pca(cloud, lists)
        assert int(torch.argmin(ev[:, 0].abs())) == want
    reps = csr_to_list(g15["rep_off"], g15["rep_idx"])
    _, ev, _, _ = util.patch_pca(t(g15["pc"]), reps)
    lam = ev[:, 0].abs().numpy()
    trace = ev.sum(dim=1).numpy()
    ref_start = int(g15["order"][0])
    assert float(g15["curv"][ref_start]) == 0.0 and (g15["curv"] == 0).sum() > 30      # the reference's tie class
    assert lam[ref_start] <= 1e-12 * trace[ref_start]
    assert lam[int(np.argmin(lam))] <= 1e-12 * trace[int(np.argmin(lam))]


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


def test_knn_graph_bfs_routes_and_vote_match_the_reference():
    """graph.getEMSTfromPC + LinkedListGraph.get_bfs_route (routes), the np.random.seed(0) starting points and the
    vote alignment of field_utils.xie_propagation_points_onbfstree (field_utils.py:657-710) against GX2, captured
    from the reference with its gurobi MIQP replaced by exhaustive search (tools/gen_golden.py gx2)."""
    g = load_golden("GX2_xie_bfstree")
    pc = g["pc"]
    adj, mean_k = util.knn_graph(pc[:, :3], 10, 0.1)
    assert len(adj) == 1000 and mean_k.shape == (1000,)
    for tag in ("t1_n", "t5_n", "t5_d"):
        orders = g[f"orders_{tag}"]
        for row in orders:
            route = util.bfs_route(adj, int(row[0]))
            assert np.array_equal(np.array(route), row)
        status = fu.align_votes(t(g[f"flips_{tag}"]))
        assert np.array_equal(status.numpy(), g[f"status_{tag}"])
        aligned = g[f"flips_{tag}"] ^ g[f"status_{tag}"][:, None]
        assert np.array_equal(aligned.sum(axis=0) > len(orders) / 2, g[f"result_{tag}"])
    # a vote where two routes are mirror images of the third: they must end up with opposite x
    a = torch.zeros(3, 50, dtype=torch.bool)
    a[1] = True
    a[2, :3] = True
    assert fu.align_votes(a).tolist() == [False, True, False]


def test_native_xyz_text_is_python_str_float_byte_for_byte(tmp_path):
    """dnp_xyz_format_f32 / dnp_xyz_parse_f32 against the literal Python of util.export_pc / util.xyz2tensor
    (util.py:46-69): random float32 bit patterns, the notation switch at 1e-4 / 1e16, denormals, signed zero, inf."""
    rng = np.random.default_rng(7)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 100.0, 1e-4, 9.9999e-5, 1e-5, 1e16, 9.999e15, 1e22, 123456.0, 0.5,
                     1e-45, 3.4e38, np.inf, -np.inf, 16777216.0, 0.1, 1e15, 1.5e16, 2.5e-7, 7.0e20, 65504.0],
                    dtype=np.float32)
    bits = rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals = np.concatenate([edge, bits[~np.isnan(bits)], rng.standard_normal(30000).astype(np.float32)])
    vals = vals[: len(vals) // 6 * 6].reshape(-1, 6)
    pc = torch.from_numpy(vals)
    util.export_pc(pc.transpose(0, 1), tmp_path / "a.xyz")
    want = "\n".join(" ".join(str(v) for v in row) for row in vals.tolist())
    assert open(tmp_path / "a.xyz").read() == want
    back = util.xyz2tensor(want)
    assert np.array_equal(back.numpy().view(np.uint32), vals.view(np.uint32))      # bit-exact round trip
    assert util._xyz_native(want, True) is not None                                  # ... through the native parser
    three = util.xyz2tensor("\n".join(" ".join(str(v) for v in row[:3]) for row in vals[:50].tolist()))
    assert three.shape == (50, 6) and float(three[:, 3:].abs().max()) == 0
    # texts that are not regular take the line-by-line path and give the reference's answer
    irregular = "1 2 3\n4 5 6 0 0 1\nnan 1 2\n\n7.5 8 9e-1"
    assert util._xyz_native(irregular, True) is None
    for txt in ("1  2 3\n4 5 6", "1\t2 3\n4 5 6", "1_0 2 3", "1 2 3 4\n1 2 3", "1 2 NaN"):
        assert util._xyz_native(txt, True) is None, txt
    assert np.array_equal(util.xyz2tensor(" +1.5 2 3 \r\n4 5 6\n").numpy()[:, :3], [[1.5, 2, 3], [4, 5, 6]])


def test_fused_interaction_rule_and_source_split_policy():
    """Host-side decisions of the batched drivers (no device involved): when may the pair kernel's epilogue leave the
    interaction partials (every 128-row tile inside two groups), and which launches split their work items."""
    from dipole_normal_prop_amd import field_utils as fu
    assert fu._tiles_within_two_groups([128, 128, 128], 384, 128)
    assert fu._tiles_within_two_groups([200, 300, 129], 629, 128)
    assert fu._tiles_within_two_groups([200, 300], 700, 128)             # loose rows behind the last patch: one more group
    assert not fu._tiles_within_two_groups([100, 20, 300], 420, 128)      # rows 0..127 hold three patches
    assert not fu._tiles_within_two_groups([130, 0, 130], 260, 128)       # an empty patch: refused (conservative)
    assert fu._tiles_within_two_groups([], 0, 128) and fu._tiles_within_two_groups([5], 5, 128)
    # brute force on random cuts
    rng = np.random.default_rng(0)
    for _ in range(200):
        sizes = rng.integers(1, 400, rng.integers(1, 12))
        n = int(sizes.sum()) + int(rng.integers(0, 200))
        grp = np.concatenate([np.repeat(np.arange(len(sizes)), sizes), np.full(n - int(sizes.sum()), len(sizes))])
        want = all(len(np.unique(grp[i:i + 128])) <= 2 for i in range(0, n, 128))
        assert fu._tiles_within_two_groups(sizes, n, 128) == want
    # three slots (round 5): patches of 100..127 points put three CONSECUTIVE groups into a tile
    assert fu._tile_group_slots([100, 20, 300], 420, 128) == 3 and fu._tile_group_slots([300, 100, 20, 300, 280], 1000, 128) == 3
    assert fu._tile_group_slots([300, 60, 20, 30, 310], 720, 128) == 0    # four groups in one tile
    assert fu._tile_group_slots([100, 0, 20, 300], 420, 128) == 0         # an empty patch between: refused
    for _ in range(200):
        sizes = rng.integers(1, 400, rng.integers(1, 12))
        n = int(sizes.sum()) + int(rng.integers(0, 200))
        grp = np.concatenate([np.repeat(np.arange(len(sizes)), sizes), np.full(n - int(sizes.sum()), len(sizes))])
        worst = max(len(np.unique(grp[i:i + 128])) for i in range(0, n, 128))
        assert fu._tile_group_slots(sizes, n, 128) == (2 if worst <= 2 else 3 if worst == 3 else 0)
    big = np.full(256, 390)
    assert fu._pick_source_split(big, 100000) == 1 and fu._pick_source_split(big[:16], 100000) == -3
    assert fu._pick_source_split(big[:32], 100000) == -3                  # a rank's share of 8: 1.25e9 pairs
    assert fu._pick_source_split(big[:128], 100000) == -3                 # a rank's share of 2: 5e9 pairs, still below the threshold
    assert fu._pick_source_split(big[:210], 100000) == 1                  # 8.2e9 pairs: above it
    # the tail is sized by its sources (round 5): the fewest trailing patches holding 1000 points, at most 8, and it ends in front
    # of the last patch of more than 512 points (such a patch would stay one wavefront per tile inside a four-wavefront item:
    # profiles/r05_tail_sweep.txt); no tail when every tail patch is a single run
    assert fu._pick_source_split(np.array([128, 300]), 1000) == -2 and fu._pick_source_split(np.array([129, 512]), 1000) == -2
    assert fu._pick_source_split(np.array([300, 600]), 1000) == 1 and fu._pick_source_split(np.array([400, 677, 677]), 1000) == 1
    assert fu._pick_source_split(np.array([677, 400, 300]), 1000) == -2 and fu._pick_source_split(np.array([600, 100, 120]), 1000) == 1
    assert fu._pick_source_split(np.array([403, 361, 554, 521, 518, 549, 389, 357]), 100000) == -2     # rank 3 of 8, the reference's grid partition
    assert fu._pick_source_split(np.array([500, 100, 100, 100, 400, 100, 100, 100, 100, 100]), 1000) == -7
    assert fu._pick_source_split(np.full(20, 100), 1000) == 1              # single-run patches only: nothing to split
    assert fu._pick_source_split(np.array([300, 1025]), 1000) == 1 and fu._pick_source_split(np.array([1025, 300]), 1000) == -1
    assert fu._pick_source_split(np.array([], dtype=np.int64), 1000) == 1


def test_store_normals_writes_any_host_tensor_through_one_copy():
    """field_utils._store_normals (the write-back of the drivers for HOST tensors): fp32 / fp64 tensors, a strided view of a
    wider tensor, a dtype numpy cannot view (fp16: the torch path), xyz columns and neighbouring columns untouched."""
    from dipole_normal_prop_amd import field_utils as fu
    n = torch.randn(500, 3)
    for dt in (torch.float32, torch.float64, torch.float16):
        pts = torch.randn(500, 6).to(dt)
        xyz = pts[:, :3].clone()
        fu._store_normals(pts, n)
        assert torch.equal(pts[:, 3:], n.to(dt)) and torch.equal(pts[:, :3], xyz)
    wide = torch.zeros(500, 9, dtype=torch.float64)
    fu._store_normals(wide[:, 2:8], n)
    assert torch.equal(wide[:, 5:8], n.double()) and float(wide[:, :5].abs().max()) == 0 and float(wide[:, 8].abs().max()) == 0
    grad = torch.randn(10, 6, requires_grad=True)
    with torch.no_grad():
        fu._store_normals(grad, torch.ones(10, 3))               # a leaf that requires grad: the torch path, under no_grad
    assert bool((grad[:, 3:] == 1).all())


def test_native_xyz_parser_row_bound_is_tight_for_minimal_rows():
    """util._xyz_native sizes its output from the text length (a row is at least "1 2 3" + newline): shortest possible rows,
    with and without the final newline, long enough for the threaded path."""
    for n in (1, 2, 7, 30000):
        for tail in ("", "\n"):
            txt = "\n".join(["1 2 3"] * n) + tail
            got = util._xyz_native(txt, False)
            assert got is not None and got.shape == (n, 3) and float((got - torch.tensor([1.0, 2.0, 3.0])).abs().max()) == 0
