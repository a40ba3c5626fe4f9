"""cProfile of the Python side of the batched patch driver on the 100k sphere (where the host time of one propagation goes)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere
pc = sphere(3000).to("cuda:0"); a = pc[:400]
for _ in range(10): fu.field_grad(a, pc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): fu.field_grad(a, pc)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"cpu issue {1e6*(t1-t0)/2000:.1f} us/call, drain {1e3*(t2-t1):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): fu.field_grad(a, pc)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
