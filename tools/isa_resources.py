#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one csrc file, from hipcc's -Rpass-analysis=kernel-resource-usage
(compile only, no GPU): the check that goes with any edit of pair_kernel.h - the scalar-unit kernels must stay at <= 64 VGPRs
(8 wavefronts per SIMD) and the tabled ones at 0 bytes of LDS.
    python tools/isa_resources.py dnp_patch.hip [-DFLAG ...]          (writes the ISA to /tmp/isa/<file>.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = sys.argv[2:]
os.makedirs("/tmp/isa", exist_ok=True)
out = os.path.join("/tmp/isa", os.path.basename(src) + ".s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "-fno-slp-vectorize", "-S",
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", *flags,
       os.path.join(ROOT, "dipole_normal_prop_amd", "csrc", src), "-o", out]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                          capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, demangle):
    d = re.sub(r"\(dnp::.*", "", d.replace("void dnp::", ""))
    print(f"{d:100s} VGPR {r.get('VGPRs', '?'):>3s} SGPR {r.get('TotalSGPRs', '?'):>3s} LDS {r.get('LDS Size [bytes/block]', '?'):>6s} "
          f"occ {r.get('Occupancy [waves/SIMD]', '?')} scratch {r.get('ScratchSize [bytes/lane]', '?')}")
