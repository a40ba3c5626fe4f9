"""The N>1 path with the real HIP kernels: two ranks (gloo for the collectives - a single-GPU box cannot host
two RCCL ranks - both computing on cuda:0) shard the per-patch fields of the fandisk golden case, gather the
interaction rows, all-reduce the partial fields, and must reproduce the reference's trace.  GPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import csr_to_list, load_golden

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tag, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dipole_normal_prop_amd import parallel
        g = load_golden("G6_patch_propagation")
        cname, dflag, wflag = tag.split("_")
        cloud = torch.from_numpy(g["pc_patchflip"] if cname == "pf" else g["pc_scrambled"])
        allp = csr_to_list(g["patch_off"], g["patch_idx"])
        patches = [(int(i), allp[int(i)]) for i in g["filtered"]]
        w = torch.from_numpy(g["weights"]) if wflag == "w" else None
        dev = torch.device("cuda:0")
        pts = cloud.clone().to(dev)
        parallel.sharded_patch_propagation(pts, [(i, p.to(dev)) for i, p in patches], [p.to(dev) for p in allp],
                                           diffuse=(dflag == "d"), weights=None if w is None else w.to(dev),
                                           start_patch=int(g[f"order_{tag}"][0]))
        tr = parallel.sharded_patch_propagation.last_trace
        sign = ((pts.cpu()[:, 3:] * cloud[:, 3:]).sum(-1) > 0).numpy()
        q.put((rank, tr["order"].copy(), (tr["sigma"] < 0)[tr["order"]].copy(), sign, pts.cpu()[:, 3:].numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["pf_d_w", "sc_n_nw"])
def test_two_hip_ranks_reproduce_the_reference_trace(dev, tag):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = load_golden("G6_patch_propagation")
    for rank, order, flipped, sign, normals in res:
        assert np.array_equal(order, g[f"order_{tag}"])
        assert np.array_equal(flipped, g[f"flipped_{tag}"])
        assert np.array_equal(sign, g[f"sign_{tag}"])
    assert np.array_equal(res[0][4], res[1][4])          # both ranks end with identical normals
