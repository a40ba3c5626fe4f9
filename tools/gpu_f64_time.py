#!/usr/bin/env python3
"""Throughput of the fp64 entry points (dnp_field_grad_f64, dnp_potential_f64, dnp_reference_field_f64; round-4 verdict:
parity-tested to 1e-12, never timed): time per call, pairs/s and the fraction of the FP64 vector peak (78.6 TFLOP/s,
MI355X_MICROARCH.md; issue rate confirmed by tools/ubench_f64.hip) at the 33 flop per pair of the fp32 roofline, with the
fp32 call of the same shape beside it.  DNP_LIB selects a library build.  -> profiles/r05_f64_time.txt

    python tools/gpu_f64_time.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402
from tools.workloads import headline_workload  # noqa: E402

FP64_PEAK, FP32_PEAK, FLOP = 78.6e12, 157.3e12, 33
dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def timeit(fn, reps):
    t_warm = time.perf_counter()
    fn(); torch.cuda.synchronize()
    while time.perf_counter() - t_warm < 0.05:
        fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def line(name, pairs, t32, t64):
    print(f"{name:44s} fp32 {t32 * 1e3:9.3f} ms {pairs / t32 / 1e9:8.1f} Gpairs/s ({pairs * FLOP / t32 / FP32_PEAK:5.3f} of 157.3 T) | "
          f"fp64 {t64 * 1e3:9.3f} ms {pairs / t64 / 1e9:8.1f} Gpairs/s ({pairs * FLOP / t64 / FP64_PEAK:5.3f} of 78.6 T) | x{t64 / t32:5.2f}",
          flush=True)


pc, patches, _ = headline_workload()
off, idx, sizes = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
pts64 = pts.double()
N = pts.shape[0]
gdir = os.path.join(ROOT, "tests", "golden")
fd = torch.from_numpy(np.load(os.path.join(gdir, "G5_fandisk_allpairs.npz"))["pc"]).to(dev)
fd64 = fd.double()
grid = util.gen_grid().to(dev)
grid64 = grid.double()

print(f"library: {os.environ.get('DNP_LIB', 'dipole_normal_prop_amd/libdnp.so')}")
line("fandisk all-pairs field_grad (11 031^2)", float(fd.shape[0]) ** 2,
     timeit(lambda: fu.field_grad(fd, fd), 20), timeit(lambda: fu.field_grad(fd64, fd64), 20))
for n in ((3000, 30000) if not quick else (3000,)):
    sub, sub64 = pts[:n].contiguous(), pts64[:n].contiguous()
    line(f"field_grad {n}^2 (sorted sphere rows)", float(n) ** 2,
         timeit(lambda: fu.field_grad(sub, sub), 10), timeit(lambda: fu.field_grad(sub64, sub64), 10))
line("potential 100 000 x 1000 lattice", float(N) * 1000,
     timeit(lambda: fu.potential(pts, grid), 20), timeit(lambda: fu.potential(pts64, grid64), 20))
line("all-pairs field_grad 100 000^2", float(N) ** 2,
     timeit(lambda: fu.field_grad(pts, pts), 5), timeit(lambda: fu.field_grad(pts64, pts64), 3))
g = torch.Generator().manual_seed(3)
tgt = (pts[:, :3].cpu() + 1e-3 * torch.randn(N, 3, generator=g)).to(dev)
tgt64 = tgt.double()
line("reference_field 100 000 -> 100 000 (3-col)", float(N) ** 2,
     timeit(lambda: fu.reference_field(pts, tgt), 5), timeit(lambda: fu.reference_field(pts64, tgt64), 3))
# accuracy of the fp64 field against a double-double-free check: fp64 kernel against itself with sources permuted (order
# sensitivity ~ rounding level) and against the fp32 kernel
E64 = fu.field_grad(pts64[:20000], pts64[:4000])
perm = torch.randperm(20000, device=dev)
E64p = fu.field_grad(pts64[:20000][perm].contiguous(), pts64[:4000])
rel = ((E64 - E64p).norm(dim=1) / E64.norm(dim=1)).max().item()
E32 = fu.field_grad(pts[:20000], pts[:4000]).double()
rel32 = ((E64 - E32).norm(dim=1) / E64.norm(dim=1)).max().item()
print(f"fp64 field, sources permuted: max row deviation {rel:.2e}; fp32 kernel against fp64 kernel: {rel32:.2e}")
