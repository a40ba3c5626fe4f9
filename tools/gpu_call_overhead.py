#!/usr/bin/env python3
"""Host-side cost of one field_utils.field_grad / potential call: the rate a Python loop of small calls reaches
(no sync inside the loop), the same loop synchronised per call, and a cProfile of the unsynchronised loop."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere

dev = torch.device("cuda:0")
for n in (256, 1000, 3000, 11031):
    pc = sphere(n).to(dev)
    for _ in range(20):
        fu.field_grad(pc, pc)
    torch.cuda.synchronize()
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        fu.field_grad(pc, pc)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(200):
        fu.field_grad(pc, pc)
        torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / 200
    print(f"N={n:6d}  issue {t_issue / reps * 1e6:7.1f} us/call   loop incl. drain {t_all / reps * 1e6:7.1f} us/call   "
          f"synchronised {t_sync * 1e6:7.1f} us/call", flush=True)
pc = sphere(1000).to(dev)
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    fu.field_grad(pc, pc)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
