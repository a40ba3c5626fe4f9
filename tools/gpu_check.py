#!/usr/bin/env python3
"""Quick on-GPU sanity + timing of the HIP path against the fp64 C oracle (developer tool).
    python tools/gpu_check.py [N]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu  # noqa: E402
from oracle import c_oracle  # noqa: E402


def sphere(n, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    nrm = x / x.norm(dim=-1, keepdim=True)
    pc = torch.cat([nrm, nrm], dim=1)
    pc[:, :3] -= pc[:, :3].mean(dim=0)
    pc[:, :3] /= (pc[:, :3].max(dim=0)[0] - pc[:, :3].min(dim=0)[0]).max()
    return pc


def rel_err(a, b):
    return (np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)).max()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    dev = torch.device("cuda:0")
    for m in (1000, 11031, 40000):
        pc = sphere(m)
        E = fu.field_grad(pc.to(dev), pc.to(dev)).cpu().numpy()
        rows = np.arange(0, m, max(1, m // 512))
        ref = c_oracle.field_grad_f64(pc.numpy(), pc.numpy()[rows])
        print(f"N={m}: max rel err vs fp64 oracle = {rel_err(E[rows], ref):.3e}", flush=True)
        phi = fu.potential(pc.to(dev), fu.util.gen_grid().to(dev)).cpu().numpy()
        pref = c_oracle.potential_f64(pc.numpy(), fu.util.gen_grid().numpy())
        print(f"      potential max rel err = {np.abs(phi - pref).max() / np.abs(pref).max():.3e}", flush=True)

    pc = sphere(n).to(dev)
    for _ in range(2):
        E = fu.field_grad(pc, pc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        E = fu.field_grad(pc, pc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"field_grad {n}x{n}: {dt * 1e3:.2f} ms  -> {n * n / dt / 1e9:.1f} Gpairs/s, "
          f"{n * n / dt * 33 / 1e12:.1f} TFLOP/s (33 flop/pair)", flush=True)
    rows = np.arange(0, n, n // 256)
    ref = c_oracle.field_grad_f64(pc.cpu().numpy(), pc.cpu().numpy()[rows])
    print(f"  max rel err vs fp64 oracle on {len(rows)} rows = {rel_err(E.cpu().numpy()[rows], ref):.3e}")


if __name__ == "__main__":
    main()
