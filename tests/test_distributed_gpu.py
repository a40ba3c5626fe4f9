"""The N>1 path with the real HIP kernels: 2, 4 and 5 ranks (gloo for the collectives - a single-GPU box cannot
host several RCCL ranks - all computing on cuda:0) shard the per-patch fields of the fandisk golden cases, gather
the interaction rows, all-reduce the fp64 partial fields, and must reproduce the reference's trace on ALL eight
G6 variants (cloud x diffuse x weights), choosing the start patch themselves.  GPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import csr_to_list, load_golden

pytestmark = pytest.mark.gpu
ALL_G6 = [f"{c}_{d}_{w}" for c in ("pf", "sc") for d in ("n", "d") for w in ("nw", "w")]


def _worker(rank, world, port, q):
    torch.set_num_threads(2)      # several ranks on one box's CPU share: torch's default (every core it sees) times the ranks thrashes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dipole_normal_prop_amd import field_utils as fu
        from dipole_normal_prop_amd import parallel
        g = load_golden("G6_patch_propagation")
        dev = torch.device("cuda:0")
        out = {}
        for tag in ALL_G6:
            cname, dflag, wflag = tag.split("_")
            cloud = torch.from_numpy(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
            allp = csr_to_list(g["patch_off"], g["patch_idx"])
            patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
            w = torch.from_numpy(g["weights"]) if wflag == "w" else None
            pts = cloud.clone().to(dev)
            parallel.sharded_patch_propagation(pts, [(i, p.to(dev)) for i, p in patches], [p.to(dev) for p in allp],
                                               diffuse=(dflag == "d"), weights=None if w is None else w.to(dev))
            tr = fu.last_trace("sharded")
            sign = ((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy()
            out[tag] = (tr["start"], tr["order"].copy(), (tr["sigma"] < 0)[tr["order"]].copy(), sign,
                        pts.cpu()[:, 3:].numpy().copy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 5])        # 5: uneven patch blocks (72 patches), the padded all-gather
def test_hip_ranks_reproduce_the_reference_traces(dev, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda x: x[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = load_golden("G6_patch_propagation")
    # the one-GPU result of the same product code, for the rank-count independence of the diffuse field
    from dipole_normal_prop_amd import field_utils as fu
    for tag in ALL_G6:
        for rank, out in res:
            start, order, flipped, sign, normals = out[tag]
            assert start == int(g[f"order_{tag}"][0]), (tag, rank)
            assert np.array_equal(order, g[f"order_{tag}"]), (tag, rank)
            assert np.array_equal(flipped, g[f"flipped_{tag}"]), (tag, rank)
            assert np.array_equal(sign, g[f"sign_{tag}"]), (tag, rank)
            assert np.array_equal(normals, res[0][1][tag][4]), (tag, rank)     # every rank ends with identical normals
        cname, dflag, wflag = tag.split("_")
        cloud = torch.from_numpy(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
        allp = csr_to_list(g["patch_off"], g["patch_idx"])
        patches = [(int(i), allp[int(i)].to(dev)) for i in g["filtered"]]
        w = torch.from_numpy(g["weights"]).to(dev) if wflag == "w" else None
        one = cloud.clone().to(dev)
        fu.strongest_field_propagation(one, patches, [p.to(dev) for p in allp], diffuse=(dflag == "d"), weights=w)
        assert np.array_equal(one.cpu()[:, 3:].numpy(), res[0][1][tag][4]), tag   # N ranks == one GPU, bit for bit


# ---- the headline workload (G19: 100 000-point sphere, 256 patches) ---------------------------------------------
def _g19_worker(rank, world, port, q):
    torch.set_num_threads(2)      # several ranks on one box's CPU share: torch's default (every core it sees) times the ranks thrashes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dipole_normal_prop_amd import field_utils as fu
        from dipole_normal_prop_amd import parallel
        from tools.workloads import headline_workload
        dev = torch.device("cuda:0")
        pc, patches, _ = headline_workload()
        allp = [p.to(dev) for p in patches]
        pts = pc.clone().to(dev)
        parallel.sharded_patch_propagation(pts, list(enumerate(allp)), allp, diffuse=True)
        tr = fu.last_trace("sharded")
        q.put((rank, tr["start"], tr["order"].copy(), (tr["sigma"] < 0)[tr["order"]].copy(), tr["chosen"].copy(),
               pts.cpu()[:, 3:].numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 5])     # 4: equal blocks of 64 patches; 5: 51/52-patch blocks, padded all-gather
def test_hip_ranks_reproduce_the_reference_on_the_headline_workload(dev, world):
    """BASELINE config 4 as north_star shards it, with the real kernels: the ranks split the 256 per-patch field
    evaluations of the 100 000-point sphere, all-gather the W rows, all-reduce the fp64 partial fields - and every
    rank ends with the REFERENCE's start patch, visit order, flips and 100 000 signs (G19), and with normals bit
    identical to the one-GPU run.  (The box's process guard allows 6 processes on the card: 5 ranks + this one; the
    8-rank split is covered by test_eight_way_split_of_the_headline_workload_is_bit_identical below and by the
    8-rank gloo test of the collectives in test_distributed_cpu.py.)"""
    from conftest import check_chosen
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_g19_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda x: x[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = load_golden("G19_headline_sphere_patch_propagation")
    from dipole_normal_prop_amd import field_utils as fu
    from tools.workloads import headline_workload
    pc, patches, _ = headline_workload()
    allp = [p.to(dev) for p in patches]
    one = pc.clone().to(dev)
    fu.strongest_field_propagation(one, list(enumerate(allp)), allp, diffuse=True)
    one = one.cpu()[:, 3:].numpy()
    ref_sign = np.unpackbits(g["sign"])[:pc.shape[0]].astype(bool)
    for rank, start, order, flipped, chosen, normals in res:
        assert start == int(g["order"][0]), rank
        assert np.array_equal(order, g["order"]) and np.array_equal(flipped, g["flipped"]), rank
        check_chosen(chosen, g["chosen"], f"G19 sharded x{world} rank {rank}")
        assert np.array_equal((normals * pc[:, 3:].numpy()).sum(-1) > 0, ref_sign), rank
        assert np.array_equal(normals, one), rank                      # N ranks == one GPU, bit for bit


def test_eight_way_split_of_the_headline_workload_is_bit_identical(dev):
    """north_star's shape - 256 patches over 8 ranks - at the level the ranks differ: each of the 8 blocks of 32
    patches evaluated on its own (dnp_patch_fields_boxed_f32 with that p_begin/p_end, dnp_interactions_f32,
    dnp_combine_signed_f32 on its own slabs) gives slabs and W rows bit identical to the one-launch evaluation, the
    stacked rows reproduce the reference's G19 trace through the device greedy kernel, and the sum of the eight fp64
    partial fields rounds to the same fp32 field as one GPU's."""
    from dipole_normal_prop_amd import field_utils as fu
    from dipole_normal_prop_amd import util
    from tools.workloads import headline_workload
    g = load_golden("G19_headline_sphere_patch_propagation")
    pc, patches, _ = headline_workload()
    P, N = len(patches), pc.shape[0]
    allp = [p.to(dev) for p in patches]
    off, idx, sizes = util.patch_csr(allp, dev)
    swork = pc.to(dev)[idx].contiguous()
    point_patch = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
    boxes = fu._patch_boxes(swork, off, None)
    bounds = fu._balanced_blocks(sizes, 8)
    assert np.array_equal(np.diff(bounds), np.full(8, 32))
    dE_all = fu._patch_slabs(swork, off, None, point_patch, 0, P, 1e-5, boxes)
    W_all = fu._interaction_rows(dE_all, swork, off, None)
    start = torch.tensor([int(g["order"][0])], device=dev)
    order, sigma, _ = fu._greedy_on_device(W_all, start)
    assert np.array_equal(order.cpu().numpy(), g["order"])
    E_one = torch.empty((N, 3), dtype=torch.float64, device=dev)
    fu._combine_signed(dE_all, sigma, 0, E_one, False)
    rows, E_sum = [], torch.zeros((N, 3), dtype=torch.float64, device=dev)
    for r in range(8):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        dE = fu._patch_slabs(swork, off, None, point_patch, lo, hi, 1e-5, boxes)
        assert torch.equal(dE, dE_all[lo:hi]), r
        rows.append(fu._interaction_rows(dE, swork, off, None))
        part = torch.empty((N, 3), dtype=torch.float64, device=dev)
        fu._combine_signed(dE, sigma, lo, part, False)
        E_sum += part
    assert torch.equal(torch.cat(rows), W_all)
    assert torch.equal(E_sum.float(), E_one.float())


def test_many_clouds_pipelined_on_the_device_equal_the_single_calls(dev):
    """parallel.sharded_patch_propagation_many with the real kernels (one process: the two-stage order of the launches -
    begin of cloud i + 1 before the end of cloud i - on one stream): all eight G6 variants' clouds as ONE queue per diffuse
    flag; every cloud ends with the normals of its own strongest_field_propagation call, bit for bit, and its trace is
    the reference's."""
    from dipole_normal_prop_amd import field_utils as fu
    from dipole_normal_prop_amd import parallel
    g = load_golden("G6_patch_propagation")
    allp = [p.to(dev) for p in csr_to_list(g["patch_off"], g["patch_idx"])]
    patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
    for dflag in ("n", "d"):
        tags = [t for t in ALL_G6 if t.split("_")[1] == dflag]
        jobs, singles = [], []
        for tag in tags:
            cname, _, wflag = tag.split("_")
            cloud = torch.from_numpy(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
            w = torch.from_numpy(g["weights"]).to(dev) if wflag == "w" else None
            jobs.append((cloud.clone().to(dev), patches, allp, w))
            one = cloud.clone().to(dev)
            fu.strongest_field_propagation(one, patches, allp, diffuse=(dflag == "d"), weights=w)
            singles.append(one)
        traces = parallel.sharded_patch_propagation_many(jobs, diffuse=(dflag == "d"))
        assert len(traces) == len(tags)
        for tag, job, one, tr in zip(tags, jobs, singles, traces):
            assert np.array_equal(tr["order"], g[f"order_{tag}"]), tag
            assert np.array_equal((tr["sigma"] < 0)[tr["order"]], g[f"flipped_{tag}"]), tag
            assert torch.equal(job[0], one), tag


def _one_rank_rccl_worker(port, q):
    torch.set_num_threads(2)      # several ranks on one box's CPU share: torch's default (every core it sees) times the ranks thrashes
    """A process of its own with a ONE-rank nccl group (all a one-GPU box allows): the asynchronous RCCL path of
    sharded_patch_propagation_many - all_gather_into_tensor(async_op=True) on RCCL's stream under the next cloud's pair kernel,
    interleaved with the diffuse form's all-reduce - against the single calls."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from dipole_normal_prop_amd import field_utils as fu
        from dipole_normal_prop_amd import parallel
        g = load_golden("G6_patch_propagation")
        allp = [p.to(dev) for p in csr_to_list(g["patch_off"], g["patch_idx"])]
        patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
        seen = {"async": 0}
        real = parallel.gather_rows_async

        def spy(*a, **k):
            W, work = real(*a, **k)
            seen["async"] += work is not None
            return W, work
        parallel.gather_rows_async = spy
        ok = True
        for dflag in ("n", "d"):
            tags = [t for t in ALL_G6 if t.split("_")[1] == dflag]
            jobs, singles = [], []
            for tag in tags:
                cname, _, wflag = tag.split("_")
                cloud = torch.from_numpy(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
                w = torch.from_numpy(g["weights"]).to(dev) if wflag == "w" else None
                jobs.append((cloud.clone().to(dev), patches, allp, w))
                one = cloud.clone().to(dev)
                fu.strongest_field_propagation(one, patches, allp, diffuse=(dflag == "d"), weights=w)
                singles.append(one)
            jobs.insert(2, (torch.zeros(4, 6, device=dev), [], []))          # a job without patches: None in its place
            traces = parallel.sharded_patch_propagation_many(jobs, diffuse=(dflag == "d"), force_async=True)
            ok &= len(traces) == len(jobs) and traces[2] is None
            del jobs[2], traces[2]
            for tag, job, one, tr in zip(tags, jobs, singles, traces):
                ok &= bool(np.array_equal(tr["order"], g[f"order_{tag}"])) and bool(torch.equal(job[0], one))
        torch.cuda.synchronize()
        q.put((ok, seen["async"]))
    finally:
        dist.destroy_process_group()


def test_many_clouds_through_the_asynchronous_rccl_gather_with_one_rank(dev):
    """Round-4 advisor: the world > 1 asynchronous path of sharded_patch_propagation_many had no test (gloo and one process both
    take the synchronous fallback).  A one-rank nccl group with force_async runs it through RCCL itself: eight asynchronous
    gathers seen, every cloud equal to its single call, a job without patches answered with None in place."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl_worker, args=(port, q))
    p.start()
    ok, n_async = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0 and ok and n_async == 8
