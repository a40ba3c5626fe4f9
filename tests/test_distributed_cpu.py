"""The N>1 path on CPU: two gloo ranks shard the per-patch field evaluations, all-gather the rows
of the interaction matrix and all-reduce the partial fields (dipole_normal_prop_amd/parallel.py).
The device entry points are replaced by an oracle-backed stand-in (test infrastructure) so that
the partition / gather / reduce plumbing is exercised without a GPU; the result must equal the
reference's golden trace."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import csr_to_list, load_golden
from dipole_normal_prop_amd import field_utils, parallel
from oracle import dipole_oracle as O


class OracleStandIn:
    """CPU stand-ins for the device entry wrappers (dnp_patch_fields_f32 / dnp_interactions_f32 /
    dnp_patch_greedy / dnp_combine_signed_f32 / dnp_patch_finish_f32).  The worker processes of this test patch them into patch_drivers
    so that the partition / gather / reduce plumbing of the N>1 path can run without a GPU; the product code
    itself has no such switch."""

    @staticmethod
    def slabs(work, off, idx, point_patch, b0, b1, eps, boxes=None, tile_boxes=None, w_part=None, source_split=1, order=None):
        N = work.shape[0]
        idx = torch.arange(N) if idx is None else idx        # None: cloud sorted by patch
        dE = torch.zeros(b1 - b0, N, 3)
        for k in range(b0, b1):
            src = idx[off[k]:off[k + 1]]
            others = point_patch != k
            dE[k - b0][others] = O.field_grad(work[src], work[others], eps=eps)
        return dE

    @staticmethod
    def interactions(dE, work, off, idx):
        P = off.shape[0] - 1
        idx = torch.arange(work.shape[0]) if idx is None else idx
        dots = (dE.double() * work[None, :, 3:].double()).sum(-1)
        return torch.stack([torch.stack([dots[k, idx[off[j]:off[j + 1]]].sum() for j in range(P)])
                            for k in range(dE.shape[0])])

    @staticmethod
    def greedy(W, start_t):
        from dipole_normal_prop_amd import field_utils as fu
        order, sigma, chosen = fu.greedy_order_from_interactions(W.numpy(), int(start_t[0]))
        return torch.from_numpy(order), torch.from_numpy(sigma), torch.from_numpy(chosen)

    @staticmethod
    def finish(pts, st, diffuse, listed, w):
        n = st.swork[:, 3:].clone()
        in_patch = st.sorted_patch >= 0
        n[in_patch] = n[in_patch] * st.sigma[st.sorted_patch[in_patch]].float()[:, None]
        if diffuse:
            dot = (st.Es.float() * n).sum(dim=-1)
            flip = in_patch & (dot <= 0)
            if listed is not None:
                flip &= listed.bool()[st.sorted_patch.clamp(min=0)]
            n[flip] = -n[flip]
        pts[st.perm, 3:] = n.to(pts.dtype)

    @staticmethod
    def combine_signed(dE, sigma, p_lo, E64, accumulate):
        if not accumulate:
            E64.zero_()
        for k in range(dE.shape[0]):
            E64 += float(sigma[p_lo + k]) * dE[k].double()


def _case():
    g = load_golden("G6_patch_propagation")
    allp = csr_to_list(g["patch_off"], g["patch_idx"])[:10]
    sub = torch.cat(allp)
    remap = -torch.ones(g["pc_patchflip"].shape[0], dtype=torch.long)
    remap[sub] = torch.arange(sub.shape[0])
    cloud = torch.from_numpy(g["pc_patchflip"])[sub].clone()
    patches = [remap[p] for p in allp]
    return cloud, patches


def _worker(rank, world, port, start, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cloud, patches = _case()
        pts = cloud.clone()
        from dipole_normal_prop_amd import patch_drivers as fu          # where the drivers look their helpers up
        fu._patch_slabs, fu._interaction_rows = OracleStandIn.slabs, OracleStandIn.interactions
        fu._patch_boxes = lambda work, off, idx: None                            # only the device kernel reads them
        fu._TileTables = lambda swork, sizes: None
        fu._greedy_on_device, fu._combine_signed = OracleStandIn.greedy, OracleStandIn.combine_signed
        fu._finish_batched = OracleStandIn.finish
        fu._prepare_work = lambda p, w: (p.detach().clone().float(), None)       # CPU working copy (no weights here)
        parallel.sharded_patch_propagation(pts, list(enumerate(patches)), patches, diffuse=True, start_patch=start)
        tr = field_utils.last_trace("sharded")
        q.put((rank, pts[:, 3:].numpy().copy(), tr["order"].copy(), tr["sigma"].copy(), tr["start"]))
    finally:
        dist.destroy_process_group()


def _worker_many(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cloud, patches = _case()
        from dipole_normal_prop_amd import patch_drivers as fu          # where the drivers look their helpers up
        fu._patch_slabs, fu._interaction_rows = OracleStandIn.slabs, OracleStandIn.interactions
        fu._patch_boxes = lambda work, off, idx: None
        fu._TileTables = lambda swork, sizes: None
        fu._greedy_on_device, fu._combine_signed = OracleStandIn.greedy, OracleStandIn.combine_signed
        fu._finish_batched = OracleStandIn.finish
        fu._prepare_work = lambda p, w: (p.detach().clone().float(), None)
        flipped = cloud.clone()
        flipped[patches[2], 3:] *= -1                      # a second, different cloud: one more patch turned over
        a, b, c = cloud.clone(), flipped.clone(), cloud.clone()
        jobs = [(a, list(enumerate(patches)), patches, None, 3), (b, list(enumerate(patches)), patches),
                (c, list(enumerate(patches)), patches)]
        traces = parallel.sharded_patch_propagation_many(jobs, diffuse=True)
        q.put((rank, [t[:, 3:].numpy().copy() for t in (a, b, c)], [(tr["order"].copy(), tr["start"]) for tr in traces]))
    finally:
        dist.destroy_process_group()


def test_many_clouds_pipelined_equal_the_one_by_one_results():
    """parallel.sharded_patch_propagation_many on two gloo ranks (the gather falls back to the in-order form there: what is
    exercised is the two-stage control flow - begin of job i + 1 before the end of job i, ONE start broadcast for all jobs):
    three jobs, pinned and default starts, two different clouds; every job must equal the oracle's single-process run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    cloud, patches = _case()
    flipped = cloud.clone()
    flipped[patches[2], 3:] *= -1
    want = [O.strongest_field_propagation(c, list(enumerate(patches)), patches, diffuse=True, start_patch=s)
            for c, s in ((cloud, 3), (flipped, None), (cloud, None))]
    for rank, normals, traces in res:
        assert len(traces) == 3
        for (ref_pts, ref_tr), got_n, (order, start) in zip(want, normals, traces):
            assert start == int(ref_tr["order"][0])
            assert np.array_equal(order, ref_tr["order"])
            assert np.array_equal(got_n, ref_pts[:, 3:].numpy())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest  # noqa: E402


@pytest.mark.parametrize("world,start", [(2, 3), (2, None), (3, None)])
def test_ranks_reproduce_the_single_process_result(world, start):
    """start = None: every rank finds the flattest patch itself (deterministic fp64 PCA) and rank 0's choice is
    broadcast - the ranks must agree with each other and with the oracle's own default start.  Three ranks over ten
    patches: uneven blocks, i.e. the padded all-gather of the interaction rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, start, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    cloud, patches = _case()
    ref_pts, ref_tr = O.strongest_field_propagation(cloud, list(enumerate(patches)), patches, diffuse=True,
                                                    start_patch=start)
    for rank, normals, order, sigma, used_start in res:
        assert used_start == int(ref_tr["order"][0])
        assert np.array_equal(order, ref_tr["order"])
        assert np.array_equal((sigma < 0)[order], ref_tr["flipped"])
        assert np.array_equal(normals, ref_pts[:, 3:].numpy())
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1])


def test_gather_rows_single_process_is_identity():
    W = torch.arange(12, dtype=torch.float64).view(3, 4)
    assert torch.equal(parallel.gather_rows(W, np.array([0, 3])), W)
    assert parallel.world() == (0, 1)


def _gather_worker(rank, world, port, sizes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dipole_normal_prop_amd import field_utils as fu
        P = len(sizes)
        W = torch.from_numpy(np.random.default_rng(5).standard_normal((P, P)))          # the same matrix on every rank
        bounds = fu._balanced_blocks(np.asarray(sizes), world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        full = parallel.gather_rows(W[lo:hi].clone(), bounds)
        again, work = parallel.gather_rows_async(W[lo:hi].clone(), bounds)     # host tensors over gloo: the synchronous form, no handle
        assert work is None and torch.equal(again, full)
        E = parallel.reduce_field(torch.full((7, 3), float(rank + 1), dtype=torch.float64))
        start = parallel.agree_on_start(torch.tensor([rank + 40]))
        q.put((rank, bool(torch.equal(full, W)), float(E[0, 0]), int(start[0]), bounds.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["headline_256_even", "ragged_61"])
def test_world8_gather_rows_reduce_and_start_agreement(kind):
    """north_star's shape: 8 ranks.  The collectives of parallel.py on 8 gloo ranks - the all-gather of the W rows with
    the headline workload's patch sizes (256 patches -> equal blocks of 32) and with ragged sizes (61 patches -> uneven
    blocks, the padded all-gather) must rebuild the single-process matrix on every rank; the all-reduce sums all eight
    partial fields; every rank ends with rank 0's start patch."""
    if kind == "headline_256_even":
        sizes = load_golden("G19_headline_sphere_patch_propagation")["sizes"].tolist()
    else:
        sizes = (np.random.default_rng(2).integers(1, 900, 61)).tolist()
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    bounds = np.array(res[0][4])
    assert bounds[0] == 0 and bounds[-1] == len(sizes) and np.all(np.diff(bounds) >= 0)
    if kind == "headline_256_even":
        assert np.all(np.diff(bounds) == 32)
    else:
        assert len(set(np.diff(bounds).tolist())) > 1
    for rank, same, esum, start, b in res:
        assert same, f"rank {rank}: gathered matrix differs from the single-process W"
        assert esum == sum(range(1, world + 1)) and start == 40 and b == bounds.tolist()
