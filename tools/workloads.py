"""Synthetic workloads shared by bench.py, the tests and the golden generator (ONE definition each).

BASELINE.json config 4 / SURVEY.md 8d: the 100 000-point sphere (torch.randn seed 1234, normalised, outward
normals, the reference's unit-box Transform of util.py:577-609) cut into 256 patches around Fibonacci-lattice
centres, whole patches sign-scrambled with seed 0.  Pure torch on the CPU; nothing here touches the reference
or the device library.
"""
import numpy as np
import torch

N_POINTS = 100_000
N_PATCHES = 256


def sphere_cloud(n=N_POINTS, seed=1234):
    """randn normalised, outward normals, unit-box Transform."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    nrm = x / x.norm(dim=-1, keepdim=True)
    pc = torch.cat([nrm, nrm], dim=1)
    pc[:, :3] -= pc[:, :3].mean(dim=0)[None, :]
    pc[:, :3] = pc[:, :3] / (pc[:, :3].max(dim=0)[0] - pc[:, :3].min(dim=0)[0]).max()
    return pc


def fibonacci_patches(pc, P=N_PATCHES):
    """Patch j = the points whose (outward) normal is nearest to the j-th of P Fibonacci-lattice directions."""
    k = torch.arange(P, dtype=torch.float64) + 0.5
    phi = torch.acos(1 - 2 * k / P)
    theta = np.pi * (1 + 5 ** 0.5) * k
    c = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], 1).float()
    lab = (pc[:, 3:6] @ c.T).argmax(dim=1)
    # one stable sort instead of P passes over the labels: the same lists (ascending indices per patch), empty patches included
    order = torch.sort(lab, stable=True).indices
    return list(torch.split(order, torch.bincount(lab, minlength=P).tolist()))


def headline_workload(n=N_POINTS, P=N_PATCHES, cloud_seed=1234, scramble_seed=0):
    """The bench workload: (cloud with whole patches flipped, patch index lists, bool[P] which patches were flipped).
    Patches are cut on the outward normals BEFORE the scramble."""
    pc = sphere_cloud(n, cloud_seed)
    patches = fibonacci_patches(pc, P)
    scramble = (torch.rand(P, generator=torch.Generator().manual_seed(scramble_seed)) < 0.5).numpy()
    for k, p in enumerate(patches):
        if scramble[k]:
            pc[p, 3:] *= -1
    return pc, patches, scramble
