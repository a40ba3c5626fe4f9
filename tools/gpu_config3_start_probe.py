#!/usr/bin/env python3
"""BASELINE config 3 (boxunion, the reference's 369 patches / representatives, G15) from the reference's start patch and
from the driver's own start: which points end differently, are they representatives, which patches got another sign."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dipole_normal_prop_amd import field_utils as fu, util  # noqa: E402

dev = torch.device("cuda:0")
g = load_golden("G15_boxunion_config3")
cloud = torch.from_numpy(g["pc"]).clone()
cloud[~torch.from_numpy(g["prefilter_sign"]), 3:] *= -1
i64 = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev)
reps = util.RepLists(util.PatchList(i64(g["rep_idx"]), np.diff(g["rep_off"]), disjoint=True),
                     util.PatchList(i64(g["rest_idx"]), np.diff(g["rest_off"]), disjoint=True))
res = {}
for name, start in (("pinned", int(g["order"][0])), ("default", None)):
    pts = cloud.clone().to(dev)
    fu.strongest_field_propagation_reps(pts, reps, diffuse=True, start_patch=start)
    tr = fu.last_trace("reps")
    flipped_globally = bool(fu.measure_mean_potential(pts) < 0)
    if flipped_globally:
        pts[:, 3:] *= -1
    res[name] = (pts.cpu(), tr["sigma"] * (-1 if flipped_globally else 1), tr["start"])
a, b = res["pinned"], res["default"]
print("starts", a[2], b[2])
differ = np.nonzero(((a[0][:, 3:] * b[0][:, 3:]).sum(-1) < 0).numpy())[0]
is_rep = np.zeros(cloud.shape[0], dtype=bool)
is_rep[g["rep_idx"]] = True
point_patch = np.zeros(cloud.shape[0], dtype=np.int64)
for k in range(len(g["rep_off"]) - 1):
    point_patch[g["rep_idx"][g["rep_off"][k]:g["rep_off"][k + 1]]] = k
    point_patch[g["rest_idx"][g["rest_off"][k]:g["rest_off"][k + 1]]] = k
print("differing points", len(differ), "of which representatives", int(is_rep[differ].sum()), "patches of them", sorted(set(point_patch[differ].tolist())))
print("patches with another sigma", np.nonzero(a[1] != b[1])[0].tolist())
