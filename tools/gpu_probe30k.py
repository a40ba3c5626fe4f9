"""field_grad at N = 30 000: per-call time synchronised and unsynchronised (the sweep's outlier there was a timing-loop artifact)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere
dev = torch.device("cuda:0")
for n in (30000, 29000, 31000, 30000):
    pc = sphere(n).to(dev)
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fu.field_grad(pc, pc); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    print(n, " ".join(f"{t:.0f}" for t in ts), flush=True)
