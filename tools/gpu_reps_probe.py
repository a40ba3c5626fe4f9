#!/usr/bin/env python3
"""BASELINE config 3 (boxunion, the reference's 369 patches / representatives): 4 synchronised calls (lazy initialisation, clock
ramp), then 12 calls back to back as bench.py's leg makes them - run under rocprofv3 --kernel-trace (tools/profile_config3.sh);
tools/summarize_trace.py turns the trace into the per-call kernel breakdown (profiles/r05_config3_kernels.txt)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipole_normal_prop_amd import field_utils as fu, util
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "G15_boxunion_config3.npz"))
cloud = torch.from_numpy(g["pc"]).clone()
cloud[~torch.from_numpy(g["prefilter_sign"]), 3:] *= -1
cloud = cloud.to(dev)
i64 = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev)
reps = util.RepLists(util.PatchList(i64(g["rep_idx"]), np.diff(g["rep_off"]), disjoint=True),
                     util.PatchList(i64(g["rest_idx"]), np.diff(g["rest_off"]), disjoint=True))
for _ in range(4):
    fu.strongest_field_propagation_reps(cloud.clone(), reps, diffuse=True)
    torch.cuda.synchronize()
for _ in range(12):
    fu.strongest_field_propagation_reps(cloud.clone(), reps, diffuse=True)
torch.cuda.synchronize()
