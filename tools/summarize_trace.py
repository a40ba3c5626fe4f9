#!/usr/bin/env python3
"""Per-call kernel breakdown from a rocprofv3 --kernel-trace CSV of a probe that makes C identical calls back to back
(tools/gpu_reps_probe.py, tools/gpu_config4_kernels.py ...): the dispatches between two consecutive launches of an ANCHOR kernel
(the first product kernel of a call) form one call; the first `skip` calls (lazy initialisation, clock ramp) are dropped and
every kernel is reported with its launches per call, median / min duration, its share of the call - plus the call's wall time
from first kernel start to the next call's first kernel start and the idle gaps between kernels inside it.

    python tools/summarize_trace.py <kernel_trace.csv> <anchor substring> [skip=3] [> profiles/rNN_<leg>_kernels.txt]"""
import csv
import re
import statistics
import sys
from collections import defaultdict

path, anchor = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["VGPR_Count"]), int(r["LDS_Block_Size"]),
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(starts) < skip + 2:
    sys.exit(f"anchor {anchor!r} found {len(starts)} times: need more than skip + 1 = {skip + 1} calls")
calls = [(starts[i], starts[i + 1]) for i in range(skip, len(starts) - 1)]          # the last call has no end marker: dropped


def short(name):
    name = re.sub(r"\(.*", "", name.replace("void ", "").replace("dnp::", ""))
    name = re.sub(r"at::native::", "", name)
    return name if len(name) <= 96 else name[:93] + "..."


per = defaultdict(lambda: {"dur": [], "n": [], "vgpr": 0, "lds": 0, "wg": 0})
wall, busy = [], []
for lo, hi in calls:
    seen = defaultdict(int)
    b = 0
    for s, e, name, vgpr, lds, wg in rows[lo:hi]:
        k = short(name)
        per[k]["dur"].append((e - s) / 1e3)
        per[k]["vgpr"], per[k]["lds"], per[k]["wg"] = vgpr, lds, wg
        seen[k] += 1
        b += e - s
    for k, n in seen.items():
        per[k]["n"].append(n)
    wall.append((rows[hi][0] - rows[lo][0]) / 1e3)
    busy.append(b / 1e3)
w = statistics.median(wall)
print(f"# {path}")
print(f"# {len(calls)} steady calls (first {skip} and the last dropped); call = from one launch of '{anchor}' to the next")
print(f"# wall time per call: median {w:.1f} us (min {min(wall):.1f}, max {max(wall):.1f}); kernels busy {statistics.median(busy):.1f} us; "
      f"idle between kernels {w - statistics.median(busy):.1f} us")
print(f"{'kernel':96s} {'n/call':>6s} {'med us':>9s} {'min us':>9s} {'us/call':>9s} {'share':>6s} {'LDS':>6s} {'wgs':>8s}")
table = []
for k, d in per.items():
    n = statistics.median(d["n"])
    med = statistics.median(d["dur"])
    table.append((med * n, k, n, med, min(d["dur"]), d["vgpr"], d["lds"], d["wg"]))
for tot, k, n, med, mn, vgpr, lds, wg in sorted(table, reverse=True):
    print(f"{k:96s} {n:6.0f} {med:9.1f} {mn:9.1f} {tot:9.1f} {100 * tot / w:5.1f}% {lds:6d} {wg:8d}")
