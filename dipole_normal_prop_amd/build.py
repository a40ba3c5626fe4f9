"""Build libdnp.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

    python -m dipole_normal_prop_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdnp.so")
SOURCES = ["dnp_api.hip", "dnp_field.hip", "dnp_patch.hip", "dnp_greedy.hip", "dnp_xie.hip"]
HEADERS = ["dnp_common.h", "pair_kernel.h", os.path.join("..", "..", "include", "dnp.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-fno-slp-vectorize"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    flags = [f for f in FLAGS if not f.endswith("-dummy")]
    cmd = [_hipcc()] + flags + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
