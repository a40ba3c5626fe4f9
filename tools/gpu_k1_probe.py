#!/usr/bin/env python3
"""Kernel-level view of the generic field_grad path (run under rocprofv3 --kernel-trace --stats):
fandisk all-pairs (BASELINE config 2) x 50 and the 100k sphere all-pairs x 5."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from dipole_normal_prop_amd import field_utils as fu
from tools.gpu_check import sphere
dev = torch.device("cuda:0")
fd = torch.from_numpy(load_golden("G5_fandisk_allpairs")["pc"]).to(dev)
big = sphere(100000).to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "fandisk"):
    for _ in range(50):
        fu.field_grad(fd, fd)
if which in ("both", "big"):
    for _ in range(5):
        fu.field_grad(big, big)
torch.cuda.synchronize()
