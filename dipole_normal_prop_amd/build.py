"""Build libdnp.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

    python -m dipole_normal_prop_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
The build is safe under torchrun / mp.spawn: one process compiles under an fcntl lock into a temporary
file and renames it into place, the others wait for the lock and find the finished library.
"""
import fcntl
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdnp.so")
SOURCES = ["dnp_api.hip", "dnp_field.hip", "dnp_patch.hip", "dnp_greedy.hip", "dnp_xie.hip", "dnp_prep.hip", "dnp_io.hip"]
HEADERS = ["dnp_common.h", "pair_kernel.h", os.path.join("..", "..", "include", "dnp.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-pthread", "-Wall", "-Wno-unused-function",
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


def build(force=False, verbose=True, extra_flags=(), out=None):
    """Compile the library.  `extra_flags` / `out` build a tuning variant next to the product library
    (tools/ A/B scripts); the product is always LIB with FLAGS."""
    target = out or LIB
    if out is None and not force and not needs_build():
        return LIB
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if out is None and not force and not needs_build():      # a peer built it while we waited
                return LIB
            tmp = f"{target}.{os.getpid()}.tmp"
            cmd = [_hipcc()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, target)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
