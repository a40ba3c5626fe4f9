#!/usr/bin/env python3
"""Size sweep of the field / potential entry points (developer tool): time per call and pairs/s."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere

dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    """median of `reps` synchronised calls after >= 50 ms of warm-up (the clocks drop within milliseconds of idling and
    the first ~40 ms of work after that run up to 25 % slower; an un-synchronised loop once hid a one-off stall in a mean)"""
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.05:
        fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))

from conftest import load_golden
grid = fu.util.gen_grid().to(dev)
fd = torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]).to(dev)
t = timeit(lambda: fu.field_grad(fd, fd), reps=50)
print(f"fandisk all-pairs (config 2): {t*1e6:.1f} us ({fd.shape[0]**2/t/1e9:.1f} Gpairs/s)", flush=True)
for n in (1000, 3000, 11031, 30000, 100000):
    pc = sphere(n).to(dev)
    t = timeit(lambda: fu.field_grad(pc, pc), reps=20 if n < 50000 else 5)
    tp = timeit(lambda: fu.potential(pc, grid))
    t1 = timeit(lambda: fu.field_grad(pc[:400], pc))
    print(f"N={n:6d}: field_grad NxN {t*1e6:9.1f} us ({n*n/t/1e9:7.1f} Gpairs/s) | potential Nx1000 {tp*1e6:7.1f} us "
          f"({n*1000/tp/1e9:6.1f} Gp/s) | field_grad 400xN {t1*1e6:7.1f} us ({400*n/t1/1e9:6.1f} Gp/s)", flush=True)
