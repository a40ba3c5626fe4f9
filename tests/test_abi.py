"""The C-ABI library builds, loads and exports every symbol include/dnp.h declares (CPU only:
no compute is launched here)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from dipole_normal_prop_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dnp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dnp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for name in ("dnp_field_grad_f32", "dnp_potential_f32", "dnp_patch_fields_f32", "dnp_interactions_f32",
                 "dnp_point_greedy_f32", "dnp_last_error", "dnp_version", "dnp_device_count"):
        assert name in syms


def test_every_declared_symbol_is_exported_and_bound(lib):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(raw, name), f"{name} declared in include/dnp.h but not exported by libdnp.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype in _lib.SIGNATURES"
    for name in _lib.SIGNATURES:
        assert name in header_symbols(), f"{name} bound in _lib.py but not declared in include/dnp.h"


def test_version_and_error_string(lib):
    assert lib.dnp_version() == 503
    assert isinstance(lib.dnp_last_error(), bytes)
    assert lib.dnp_device_count() >= 0


def test_argument_validation_needs_no_device(lib):
    # negative sizes / bad leading dimensions are rejected before any HIP call
    rc = lib.dnp_field_grad_f32(None, -1, 6, None, None, 4, 3, None, 1e-5, 0, None, 3, 0, 0, None, None, None, 0, None)
    assert rc == -1 and b"negative" in lib.dnp_last_error()
    dummy = ctypes.c_void_p(16)
    rc = lib.dnp_field_grad_f32(dummy, 4, 5, None, dummy, 4, 3, None, 1e-5, 0, dummy, 3, 0, 0, None, None, None, 0, None)
    assert rc == -1 and b"ld_src" in lib.dnp_last_error()
    rc = lib.dnp_field_grad_f32(dummy, 4, 6, None, dummy, 4, 3, None, 1e-5, 0, dummy, 3, 0, 0, None, None, None, 0, None)
    assert rc == -3 and b"workspace" in lib.dnp_last_error()
    rc = lib.dnp_point_greedy_f32(dummy, 2 ** 20, 6, 0, 1e-6, 0, None, None, 0, 0, None, 0, None)
    assert rc == -1 and b"exceeds" in lib.dnp_last_error()
    rc = lib.dnp_patch_fields_f32(dummy, 10, 6, dummy, dummy, 3, dummy, 2, 1, 1e-5, dummy, None)
    assert rc == -1


def test_workspace_queries(lib):
    assert lib.dnp_field_grad_workspace_bytes(0, 10, 0) > 0
    small = lib.dnp_field_grad_workspace_bytes(100, 100, 15000)
    big = lib.dnp_field_grad_workspace_bytes(100000, 100000, 15000)
    assert 0 < small < big < (2 << 30)
    # 8 recursion leaves of 12 500 rows each at S = 100 000 (field_utils.py:73-94)
    assert big >= 8 * 100000 * 3 * 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a device")
def test_product_path_fails_loudly_without_a_device():
    from dipole_normal_prop_amd import field_utils as fu
    with pytest.raises(_lib.DnpError, match="no CPU fallback"):
        fu.field_grad(torch.zeros(4, 6), torch.zeros(3, 3))
    with pytest.raises(_lib.DnpError):
        fu.potential(torch.zeros(4, 6), torch.zeros(3, 3))
    with pytest.raises(_lib.DnpError):
        fu.strongest_field_propagation_points(torch.zeros(4, 6))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dipole_normal_prop_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), f"{fn} imports the oracle"
            assert "c_oracle" not in src and "dipole_oracle" not in src, f"{fn} references an oracle module"


def test_field_utils_re_exports_the_split_modules():
    """Round 5 moved the staging plumbing, the xie family and the drivers out of field_utils.py: the public surface (INTEGRATION.md
    section 1) is unchanged - every name of __all__ exists, the xie names, the drivers and last_trace / flush_warnings are the split
    modules' own objects, the reference's star-import helpers (torch, np, util) are still there - and the drivers' knobs exist in ONE
    place (a copy in field_utils would be a knob that does nothing)."""
    from dipole_normal_prop_amd import _staging, patch_drivers, point_driver, xie
    from dipole_normal_prop_amd import field_utils as fu
    for name in fu.__all__:
        assert hasattr(fu, name), name
    for name in ("xie_field", "xie_intersaction", "xie_distance", "xie_propagation_points_in_order", "xie_propagation_points_onbfstree",
                 "align_votes"):
        assert getattr(fu, name) is getattr(xie, name)
    for name in ("strongest_field_propagation", "strongest_field_propagation_reps", "greedy_order_from_interactions", "_batched_begin",
                 "_batched_end", "_patch_slabs", "_TileTables", "_balanced_blocks", "_pick_source_split"):
        assert getattr(fu, name) is getattr(patch_drivers, name), name
    assert fu.strongest_field_propagation_points is point_driver.strongest_field_propagation_points
    assert fu.last_trace is _staging.last_trace and fu.flush_warnings is _staging.flush_warnings and fu._tls is _staging._tls
    for knob in ("PATCH_MODE", "SLAB_BUDGET_BYTES", "SLAB_BLOCK_BYTES", "SLAB_FREE_CHECK_BYTES", "PATCH_GREEDY_MAX", "TAIL_SOURCES"):
        assert hasattr(patch_drivers, knob) and not hasattr(fu, knob), knob
    for knob in ("POINT_GREEDY_FORM", "POINT_GREEDY_GROUPS", "POINT_GREEDY_MAX_PER_GROUP"):
        assert hasattr(point_driver, knob) and not hasattr(fu, knob), knob
    for mod, limit in (("field_utils.py", 300), ("patch_drivers.py", 900), ("point_driver.py", 150), ("_staging.py", 400), ("xie.py", 300)):
        n = len(open(os.path.join(ROOT, "dipole_normal_prop_amd", mod)).read().splitlines())
        assert n <= limit, (mod, n)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_graft_entry_build_runs_and_agrees_on_the_abi_version():
    """The driver's own build check (__graft_entry__.build: compile every native piece, load the library, compare versions) - run
    here too, so that an ABI bump that forgets one of header / library / binding / entry point fails in the CPU suite (round 5: the
    entry point had pinned the number itself)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "dnp_version" in out.stdout
