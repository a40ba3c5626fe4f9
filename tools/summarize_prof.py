#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs under gpurun_out/prof (tools/profile_bench.sh <round>) into the committed summaries
under profiles/:  <round>_kernel_stats.csv (per kernel: the --stats figures over ALL dispatches and, from the
kernel trace, over the STEADY dispatches only - bench.py's 10 warm-up steps run while the clocks ramp and are not
part of what the bench times), <round>_pmc_summary.md and <round>_pmc_traffic.json (HBM bytes per launch of the
dominant kernel, read back by bench.py).

    python tools/summarize_prof.py r03 [warmup_steps=10]

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KB, collected in separate
--pmc passes; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by exactly 2x and is
uncalibrated for other widths - this kernel's global reads are 4-byte-per-lane staging loads, so both the
raw figure and the 2x upper bound are recorded; WRITE_SIZE is taken as is."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 10
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
if not os.path.isdir(src):
    src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
# the bench line printed under rocprof says how many untimed steps really ran (its clock warm-up + the W warm-up steps)
_bl = os.path.join(src, "bench_line_under_rocprof.json")
if len(sys.argv) <= 2 and os.path.exists(_bl) and os.path.getsize(_bl):
    try:
        _d = json.load(open(_bl))
        warmup = int(_d["warmup"]) + int(_d.get("clock_warmup", {}).get("untimed_steps_before_the_warmup_steps", 0))
    except Exception:
        pass

os.makedirs(dst, exist_ok=True)


def counters(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(src, sub, f"{rnd}_counter_collection.csv")
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "trace", f"{rnd}_kernel_stats.csv")))}
# steady-state figures from the per-dispatch trace: a kernel launched once per step drops its first `warmup` dispatches
import numpy as np  # noqa: E402
per_kernel = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(src, "trace", f"{rnd}_kernel_trace.csv"))):
    per_kernel[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
steady = {}
for name, rows in per_kernel.items():
    rows.sort()
    d = np.array([x[1] for x in rows], dtype=np.float64)
    tail = d[warmup:] if len(d) > warmup else d
    steady[name] = (len(tail), float(tail.mean()), float(np.median(tail)), float(tail.min()), float(tail.max()))
with open(os.path.join(dst, f"{rnd}_kernel_stats.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats of bench.py --steps 40 --warmup 10 --headline-only; Steady* = without each "
            f"kernel's first {warmup} dispatches (the bench's untimed steps: clock warm-up + warm-up steps)\n")
    f.write("Name,Calls,AverageNs,MinNs,MaxNs,Percentage,SteadyCalls,SteadyAverageNs,SteadyMedianNs,SteadyMinNs,SteadyMaxNs\n")
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        sc = steady.get(name, (0, 0, 0, 0, 0))
        f.write(f'"{name}",{r["Calls"]},{r["AverageNs"]},{r["MinNs"]},{r["MaxNs"]},{r["Percentage"]},{sc[0]},{sc[1]:.0f},{sc[2]:.0f},'
                f"{sc[3]:.0f},{sc[4]:.0f}\n")
bl = os.path.join(src, "bench_line_under_rocprof.json")
if os.path.exists(bl) and os.path.getsize(bl):
    shutil.copy(bl, os.path.join(dst, f"{rnd}_bench_line_under_rocprof.json"))
allc = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc"):
    for k, v in counters(sub).items():
        for c, xs in v.items():
            allc[k][c] = sum(xs) / len(xs)

lines = [f"# {rnd}: rocprofv3 summary of `python bench.py --steps 40 --warmup 10 --no-cpu-baseline --headline-only` (1x MI355X)", "",
         "Per-dispatch averages. Kernel-trace stats and every PMC group come from separate runs "
         "(tools/profile_bench.sh).  'steady' = without each kernel's untimed dispatches (the bench's clock warm-up + warm-up steps).", ""]
traffic = {}
for k in sorted(allc, key=lambda n: -float(stats.get(n, {"TotalDurationNs": 0})["TotalDurationNs"])):
    if "dnp::" not in k:
        continue
    st = stats.get(k)
    c = allc[k]
    lines.append(f"## `{k[:110]}`")
    if st:
        sc = steady.get(k)
        lines.append(f"- calls {st['Calls']}, average {float(st['AverageNs']) / 1e3:.1f} us, min {float(st['MinNs']) / 1e3:.1f} us, "
                     f"{st['Percentage']} % of GPU time" + (f"; steady ({sc[0]} dispatches): average {sc[1] / 1e3:.1f} us, "
                                                            f"median {sc[2] / 1e3:.1f} us" if sc else ""))
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        f, w = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
        lines.append(f"- HBM: FETCH_SIZE {f / 1e6:.1f} MB raw ({2 * f / 1e6:.1f} MB with the gfx950 2x wide-read correction as "
                     f"an upper bound), WRITE_SIZE {w / 1e6:.1f} MB -> {(f + w) / 1e6:.1f} .. {(2 * f + w) / 1e6:.1f} MB per launch")
        if "pair_kernel" in k and "float, float" in k:
            traffic = {"kernel": k, "fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
                       "hbm_bytes_per_launch_upper": 2 * f + w,
                       "avg_launch_us": float(st["AverageNs"]) / 1e3 if st else None,
                       "steady_avg_launch_us": steady[k][1] / 1e3 if k in steady else None}
    if "SQ_INSTS_VALU" in c:
        lines.append(f"- SQ: waves {c.get('SQ_WAVES', 0):.0f}, VALU wave-instructions {c['SQ_INSTS_VALU']:.4g}, "
                     f"LDS instructions {c.get('SQ_INSTS_LDS', 0):.4g}, SALU {c.get('SQ_INSTS_SALU', 0):.4g}, "
                     f"LDS bank-conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0):.4g}")
        lines.append(f"- SQ cycles (quad-cycle units): WAVE_CYCLES {c.get('SQ_WAVE_CYCLES', 0):.4g}, WAIT_INST_ANY "
                     f"{c.get('SQ_WAIT_INST_ANY', 0):.4g}, WAIT_ANY {c.get('SQ_WAIT_ANY', 0):.4g}, ACTIVE_INST_ANY "
                     f"{c.get('SQ_ACTIVE_INST_ANY', 0):.4g}, ACTIVE_INST_VALU {c.get('SQ_ACTIVE_INST_VALU', 0):.4g}")
    if "GRBM_GUI_ACTIVE" in c and st:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (float(st["AverageNs"]) * 1e-9) / 1e9
        lines.append(f"- GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.4g} (sum over 8 XCDs) -> effective clock {clk:.2f} GHz")
    lines.append("")
open(os.path.join(dst, f"{rnd}_pmc_summary.md"), "w").write("\n".join(lines))
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
