"""The HOST functions of libdnp (text I/O, voxel-cell merge, the launch planner) under AddressSanitizer + UBSan: a CPU
build of csrc/dnp_io.hip, csrc/dnp_prep.hip, csrc/dnp_field.hip and csrc/dnp_api.hip with host-side instrumentation only
(GPU sanitizers are not available on the pool), driven by tests/fuzz/fuzz_host.cpp - including allocation failures injected
into operator new: nothing may leave an extern "C" entry point as an exception.  No device code runs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")
def test_host_functions_are_clean_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "dipole_normal_prop_amd", "csrc")
    exe = str(tmp_path / "fuzz_host")
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "--offload-arch=gfx950", "-x", "hip",
           os.path.join(csrc, "dnp_io.hip"), os.path.join(csrc, "dnp_prep.hip"), os.path.join(csrc, "dnp_field.hip"),
           os.path.join(csrc, "dnp_api.hip"),
           "-x", "c++", os.path.join(ROOT, "tests", "fuzz", "fuzz_host.cpp"),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-2000:]
    run = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    assert "merge:" in run.stdout and "text:" in run.stdout and "alloc:" in run.stdout and "planner:" in run.stdout
    shutil.rmtree(tmp_path, ignore_errors=True)
