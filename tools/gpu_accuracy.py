#!/usr/bin/env python3
"""Error distribution of the fp32 HIP path against fp64 (developer tool): per-row |dE|/|E| on the
G11 random-normal cloud, fandisk and the 100k sphere, with the reference-class fp32 CPU error on a
row sample beside it."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dipole_normal_prop_amd import field_utils as fu
from oracle import c_oracle, dipole_oracle as O
from test_oracle_golden import _g11_cloud, sphere100k
from conftest import load_golden

dev = torch.device("cuda:0")

def stats(name, pc, rows_cpu=256):
    d = pc.to(dev)
    E32 = fu.field_grad(d, d).double().cpu().numpy()
    E64 = fu.field_grad(d.double(), d.double()).cpu().numpy()
    n = pc.shape[0]
    rows = np.arange(0, n, max(1, n // rows_cpu))
    ref = c_oracle.field_grad_f64(pc.numpy(), pc.numpy()[rows])
    e64 = np.linalg.norm(E64[rows] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    err = np.linalg.norm(E32 - E64, axis=1) / np.linalg.norm(E64, axis=1)
    cpu32 = O.field_grad(pc, pc[rows]).double().numpy()
    ecpu = np.linalg.norm(cpu32 - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print(f"{name:10s} N={n}: GPU f64 vs C oracle max {e64.max():.1e} | GPU f32 vs f64: median {np.median(err):.2e} "
          f"p99 {np.quantile(err, .99):.2e} max {err.max():.2e} (rows>1e-6: {(err > 1e-6).sum()}, >5e-6: {(err > 5e-6).sum()}) | "
          f"on sample rows: GPU max {err[rows].max():.2e}, reference-class CPU fp32 max {ecpu.max():.2e} median {np.median(ecpu):.2e}",
          flush=True)
    w = np.argmax(err)
    print(f"           worst row {w}: |E|={np.linalg.norm(E64[w]):.3e}, median |E|={np.median(np.linalg.norm(E64, axis=1)):.3e}")

stats("G11", _g11_cloud())
stats("fandisk", torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]))
stats("sphere100k", sphere100k(), rows_cpu=64)

# the bench kernel's slabs against the REFERENCE's own fp32 slabs (G19) and against fp64, every row of three patches
from dipole_normal_prop_amd import util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402
g19 = load_golden("G19_headline_sphere_patch_propagation")
pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr([p.to(dev) for p in patches], dev)
sw = pc.to(dev)[idx].contiguous()
N, P = sw.shape[0], len(sizes)
pp = torch.repeat_interleave(torch.arange(P, device=dev), off[1:] - off[:-1])
boxes, tiles = fu._patch_boxes(sw, off, None), fu._TileTables(sw, sizes)
for k in (int(x) for x in g19["slab_patches"]):
    dE = fu._patch_slabs(sw, off, None, pp, k, k + 1, 1e-5, boxes, tiles.boxes)[0]
    full = torch.empty_like(dE)
    full[idx] = dE
    others = torch.ones(N, dtype=torch.bool)
    others[patches[k]] = False
    got = full.cpu()[others].double().numpy()
    ref32 = g19[f"dE_{k}"].astype(np.float64)
    ref64 = c_oracle.field_grad_f64(pc[patches[k]].numpy(), pc[others].numpy())
    nrm = np.linalg.norm(ref64, axis=1)
    e_ref = np.linalg.norm(got - ref32, axis=1) / nrm
    e_64 = np.linalg.norm(got - ref64, axis=1) / nrm
    r_64 = np.linalg.norm(ref32 - ref64, axis=1) / nrm
    print(f"G19 slab of patch {k:3d} ({len(patches[k])} sources x {int(others.sum())} rows): kernel vs reference fp32 max {e_ref.max():.2e} "
          f"median {np.median(e_ref):.2e} | kernel vs fp64 max {e_64.max():.2e} median {np.median(e_64):.2e} | reference fp32 vs "
          f"fp64 max {r_64.max():.2e} median {np.median(r_64):.2e}", flush=True)
