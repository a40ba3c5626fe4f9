#!/usr/bin/env python3
"""profiles/SUMMARY_<round>.md from the per-leg kernel traces of tools/profile_legs.sh (gpurun_out/prof_<round>_legs/<leg>.
kernel_trace.csv + <leg>.json): for every leg the call's wall time and its kernels (launches per call, median / min us,
us per call, share), and for the product kernels the sidecar names their work per launch and the roofline that bounds them -
FP32 / FP64 vector ALU at 33 (13 for the potential) flop per pair against 157.3 / 78.6 TFLOP/s, HBM bytes against 8 TB/s, or
microseconds per dependent step for the latency-bound loops.  One table of the product kernels at the top.

    python tools/make_summary.py r05 [commit]"""
import csv
import json
import os
import re
import statistics
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_legs")
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
PEAK = {"valu32": 157.3e12, "valu64": 78.6e12, "hbm": 8.0e12}


def short(name):
    name = re.sub(r"\(.*", "", name.replace("void ", "").replace("dnp::", ""))
    name = re.sub(r"at::native::", "", name)
    return name if len(name) <= 100 else name[:97] + "..."


def analyse(leg):
    meta = json.load(open(os.path.join(src, leg + ".json")))
    rows = []
    for r in csv.DictReader(open(os.path.join(src, leg + ".kernel_trace.csv"))):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    anchor, warm, reps = meta.get("anchor", ""), meta["warm"], meta["reps"]
    if anchor:
        starts = [i for i, r in enumerate(rows) if anchor in r[2]]
        per_call = max(1, round(len(starts) / (warm + reps)))        # an anchor kernel may run more than once per call
        starts = starts[::per_call]
        calls = [(starts[i], starts[i + 1]) for i in range(warm + 1, len(starts) - 1)]
    else:                                                            # no product kernel to anchor on: the steady region in equal parts
        calls = []
    per = defaultdict(lambda: {"dur": [], "n": []})
    wall, busy = [], []
    for lo, hi in calls:
        seen = defaultdict(int)
        b = 0
        for s, e, name in rows[lo:hi]:
            k = short(name)
            per[k]["dur"].append((e - s) / 1e3)
            seen[k] += 1
            b += e - s
        for k, n in seen.items():
            per[k]["n"].append(n)
        wall.append((rows[hi][0] - rows[lo][0]) / 1e3)
        busy.append(b / 1e3)
    return meta, per, wall, busy, len(calls)


legs = sorted(f[:-5] for f in os.listdir(src) if f.endswith(".json"))
order = ["config4_driver", "config3_reps", "config2_fandisk", "allpairs_100k", "config5_reference_field", "potential_lattice",
         "config1_points", "config1_points_f64", "config4_driver_f64", "config2_fandisk_f64", "allpairs_100k_f64",
         "config5_reference_field_f64", "potential_lattice_f64", "xie_order", "xie_order_f64", "prep_partition"]
legs = [l for l in order if l in legs] + [l for l in legs if l not in order]
top, body = [], []
for leg in legs:
    meta, per, wall, busy, ncalls = analyse(leg)
    body.append(f"\n## {leg} - {meta.get('what', '')}\n")
    if not ncalls:
        body.append(f"wall time per call (host clock, under the profiler): {meta['wall_ms_per_call_under_profiler'] * 1e3:.1f} us; no product kernel to cut calls at - "
                    "torch and rocPRIM kernels only (sort, searchsorted, gathers) + the host-side merge.\n")
        continue
    w = statistics.median(wall)
    body.append(f"{ncalls} steady calls; wall per call (first kernel start to the next call's) median {w:.1f} us, min {min(wall):.1f}; kernels busy "
                f"{statistics.median(busy):.1f} us; idle between kernels {w - statistics.median(busy):.1f} us; host clock {meta['wall_ms_per_call_under_profiler'] * 1e3:.1f} us\n")
    body.append("| kernel | launches / call | median us | min us | us / call | share | work / launch | roofline |")
    body.append("|---|---|---|---|---|---|---|---|")
    table = []
    for k, d in per.items():
        n, med = statistics.median(d["n"]), statistics.median(d["dur"])
        table.append((med * n, k, n, med, min(d["dur"])))
    for tot, k, n, med, mn in sorted(table, reverse=True):
        if tot < 0.002 * w and not any(sub in k for sub in meta["kernels"]):
            continue
        work = roof = ""
        for sub, m in meta["kernels"].items():
            if sub in k:
                if m["bound"] in PEAK and m["units"] > 0:
                    rate = m["units"] * m["per_unit"] / (med * 1e-6)
                    frac = rate / PEAK[m["bound"]]
                    unit = "TB/s" if m["bound"] == "hbm" else "TFLOP/s"
                    work = f"{m['units']:.3e} {m['unit']}"
                    roof = f"{rate / 1e12:.2f} {unit} = **{frac:.3f}** of {'8 TB/s HBM' if m['bound'] == 'hbm' else ('157.3 T FP32 VALU' if m['bound'] == 'valu32' else '78.6 T FP64 VALU')}"
                    top.append((leg, k, m, med, frac, roof, work))
                elif m["bound"] == "latency":
                    work = f"{m['units']:.0f} {m['unit']}"
                    roof = f"latency-bound: {med / m['units']:.3f} us per step"
                    top.append((leg, k, m, med, None, roof, work))
                elif m["bound"] == "hbm":
                    roof = "HBM-bound second pass"
                break
        body.append(f"| `{k}` | {n:.0f} | {med:.1f} | {mn:.1f} | {tot:.1f} | {100 * tot / w:.1f} % | {work} | {roof} |")
out = [f"# SUMMARY {rnd} - per-kernel profile of the final tree (commit {commit})\n",
       "rocprofv3 `--kernel-trace`, one process per leg (`tools/profile_legs.sh`, `tools/gpu_leg.py`): 3 synchronised calls, then the "
       "calls back to back; durations are medians over the steady calls, one MI355X.  Generated by `tools/make_summary.py`; peaks from "
       "`MI355X_MICROARCH.md` (FP32 vector 157.3 TFLOP/s, FP64 vector 78.6, HBM 8 TB/s); 33 flop per pair (the exact chain's count), 13 "
       "for the potential.\n",
       "## Product kernels\n",
       "| leg | kernel | work per launch | median us | roofline |", "|---|---|---|---|---|"]
for leg, k, m, med, frac, roof, work in top:
    out.append(f"| {leg} | `{k}` | {work} | {med:.1f} | {roof} |")
open(os.path.join(ROOT, "profiles", f"SUMMARY_{rnd}.md"), "w").write("\n".join(out + body) + "\n")
print("\n".join(out[:5] + out[5:40]))
