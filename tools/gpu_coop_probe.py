"""Runs the per-point propagation in one chosen form (argv[1] = coop | single) - the process rocprofv3 was wrapped around to
find that hipLaunchCooperativeKernel makes the profiler crash at exit (the launch was replaced by an occupancy check)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dipole_normal_prop_amd import field_utils as fu
from dipole_normal_prop_amd import point_driver as ptd  # noqa: E402
from tools.gpu_check import sphere
which = sys.argv[1]
dev = torch.device("cuda:0")
if which == "coop":
    ptd.POINT_GREEDY_FORM = 2
    fu.strongest_field_propagation_points(sphere(3000).to(dev), diffuse=True)
elif which == "single":
    ptd.POINT_GREEDY_FORM = 1
    fu.strongest_field_propagation_points(sphere(3000).to(dev), diffuse=True)
else:
    fu.field_grad(sphere(3000).to(dev), sphere(3000).to(dev))
torch.cuda.synchronize()
print("done", which)
