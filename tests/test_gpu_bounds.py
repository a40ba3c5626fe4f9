"""The class-level guard behind round 3's memory fault (a wave-uniform table read past its end, found only because the
table happened to end on a page boundary): a -DDNP_BOUNDS build of the library checks every such table access against the
table's length and counts violations in a device error word (csrc/pair_kernel.h, PairBounds).  Here: the deterministic
shape that faulted, under the check build of TODAY's kernel (no violation) and under the check build of the kernel as it
was BEFORE the fix (-DDNP_BUG_A8D48F5: the violation is caught - the guard works), and a dozen seconds of tools/gpu_fuzz.py
under the check build.  Each runs as a subprocess (a process binds one library)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_build(name, flags):
    from dipole_normal_prop_amd import build
    path = os.path.join(ROOT, "tools", "bin", f"libdnp_{name}.so")
    deps = [os.path.join(build.CSRC, f) for f in build.SOURCES + build.HEADERS]
    if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        build.build(extra_flags=flags, out=path, verbose=False)
    return path


def _probe(lib_path, seconds):
    env = dict(os.environ, DNP_LIB=lib_path)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_bounds_probe.py"), str(seconds)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])


def test_todays_kernels_stay_inside_their_tables(dev):
    out = _probe(_check_build("bounds", ["-DDNP_BOUNDS"]), 12)
    assert out["round3_shape_plain"][0] == 0, out["round3_shape_plain"]
    assert out["round3_shape_split_tail"][0] == 0, out["round3_shape_split_tail"]
    assert out["round3_shape_results_equal"]
    assert out["fuzz"]["cases"] > 20 and not out["fuzz"]["failures"], out["fuzz"]
    assert out["fuzz"]["bounds"][0] == 0, out["fuzz"]["bounds"]


def test_the_check_build_catches_the_round3_fault(dev):
    """The kernel as it was before commit a8d48f5 (the tile index of the last workgroup's target-less wavefronts not clamped)
    under the check build: the tile-box counter is hit on the deterministic shape, in both launch forms - and nothing faults,
    the check clamps."""
    out = _probe(_check_build("bounds_bug", ["-DDNP_BOUNDS", "-DDNP_BUG_A8D48F5"]), 0)
    for form in ("round3_shape_plain", "round3_shape_split_tail"):
        total, by_table = out[form]
        assert by_table["tile_box"] > 0 and total == by_table["tile_box"], (form, out[form])
