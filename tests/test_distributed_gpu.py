"""The N>1 path with the real HIP kernels: 2 and 4 ranks (gloo for the collectives - a single-GPU box cannot
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
