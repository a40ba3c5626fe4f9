"""The register / LDS budget of the hot kernels, from a compile-only pass (hipcc cross-compiles gfx950 without a GPU;
tools/isa_resources.py).  The scalar-unit pair kernel runs at 8 wavefronts per SIMD because it stays at <= 64 VGPRs, and its
tabled forms - the bench's kernel and the exchange-tail kernel - carry no LDS at all (round 3: merely carrying 9 KB cost
2.4 %); hipcc's allocation for this kernel is fragile (codegen caveat: a side effect in the prologue
moved it from 61 to 72-78 VGPRs), so an edit that breaks the budget should fail here, not be found in a profile."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_tabled_pair_kernels_keep_their_register_and_lds_budget():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_resources.py"), "dnp_patch.hip"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(\S.*?)\s+VGPR\s+(\d+) SGPR\s+(\d+) LDS\s+(\d+) occ (\d+) scratch (\d+)", line)
        if m:
            rows[m.group(1).strip()] = tuple(int(x) for x in m.groups()[1:])
    tabled = {k: v for k, v in rows.items() if k.startswith("pair_kernel_scalar<float, float, 0, 2, 0, true, true, true")}
    assert len(tabled) == 6, sorted(rows)                      # plain / exchange-tail form, each without the partials and with 2 / 3 group slots
    for name, (vgpr, sgpr, lds, occ, scratch) in tabled.items():
        assert vgpr <= 64 and occ == 8, (name, vgpr, occ)      # 8 wavefronts per SIMD
        assert sgpr <= 80, (name, sgpr)                        # above 80 the hardware admits 7 (MI355X_MICROARCH.md)
        assert lds == 0 and scratch == 0, (name, lds, scratch)
    for name, (vgpr, sgpr, lds, occ, scratch) in rows.items():
        if name.startswith("pair_kernel_scalar<float"):
            assert vgpr <= 64 and scratch == 0, (name, vgpr, scratch)
        if name.startswith("pair_kernel_scalar<double"):           # round 5, fp64 slabs: KT = 2 in doubles, >= 4 wavefronts per SIMD
            assert vgpr <= 128 and occ >= 4 and scratch == 0 and lds == 0, (name, vgpr, occ, scratch, lds)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_no_pair_kernel_needs_a_private_segment():
    """Round 4 found every LDS-staged pair kernel with 12 bytes of scratch per lane (two of six staged scalars captured by
    reference went through the private segment: a store -> load round trip in each workgroup's prologue); the staged row
    travels by value now.  No pair kernel of the generic entry points may need scratch, and the small-call kernel
    (KT = 1, fp32) keeps its 8 wavefronts per SIMD."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_resources.py"), "dnp_field.hip"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    seen = 0
    for line in out.stdout.splitlines():
        m = re.match(r"(pair_kernel\S*<.*?>)\s+VGPR\s+(\d+) SGPR\s+(\d+) LDS\s+(\d+) occ (\d+) scratch (\d+)", line)
        if not m:
            continue
        seen += 1
        assert int(m.group(6)) == 0, line
        if m.group(1).startswith("pair_kernel<float, double, 0, 1,") or m.group(1).startswith("pair_kernel<float, double, 1, 1,"):
            assert int(m.group(5)) == 8, line
    assert seen >= 24


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_exchange_hand_off_is_write_through_stores_then_a_wait_then_the_ticket():
    """Round-4 advisor: the split tail's hand-off uses relaxed agent-scope atomics only and is correct because of two target facts -
    the run terms are stored write-through (sc1) and an s_waitcnt vmcnt(0) stands between them and the ticket's atomic add
    (gfx9: vmcnt counts stores) - plus the last arriver's acquire (buffer_inv sc1) before it reads the terms back.  A compiler
    change that drops any of the three would fail silently on the GPU; it fails here."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_resources.py"), "dnp_patch.hip"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = open("/tmp/isa/dnp_patch.hip.s").read()
    names = re.findall(r"^(_ZN3dnp18pair_kernel_scalarIffLi0ELi2ELi0ELb1ELb1ELb1ELi\dELi4ELi4ELb1EEEvNS_8PairArgsIT_T0_EE):", text, flags=re.M)
    assert len(set(names)) == 3, names                     # the exchange-tail kernel without partials and with 2 / 3 group slots
    for name in set(names):
        body = text[text.index(name + ":"):text.index(".end_amdhsa_kernel", text.index(name + ":"))]
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(".")]
        # the ticket is the one atomic add that RETURNS a value (sc0); the others are the Inf / NaN counters of the direct epilogue
        at = [i for i, ln in enumerate(lines) if ln.startswith("global_atomic_add") and ln.endswith("sc0")]
        assert len(at) == 1, (name, at)
        before = lines[:at[0]]
        last_wait = max(i for i, ln in enumerate(before) if ln.startswith("s_waitcnt") and "vmcnt(0)" in ln)
        stores = [i for i, ln in enumerate(before) if ln.startswith("global_store_dwordx2") and "sc1" in ln]
        assert len(stores) >= 6 and max(stores) < last_wait, (name, stores[-3:], last_wait)     # 6 doubles per lane, all before the wait
        assert not any(ln.startswith(("global_store", "global_load", "flat_")) for ln in before[last_wait:]), name
        after = lines[at[0]:]
        inv = [i for i, ln in enumerate(after) if ln.startswith("buffer_inv") and "sc1" in ln]
        loads = [i for i, ln in enumerate(after) if ln.startswith("global_load_dwordx2") and "sc1" in ln]
        assert inv and loads and min(inv) < min(loads), (name, inv[:2], loads[:2])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_blocked_xie_solve_loop_waits_for_its_oldest_column_only():
    """The solve kernel of the blocked ordered propagation keeps 8 corner columns in flight ACROSS the back edge of its step loop:
    the loads are hand-issued (inline asm) and every step waits with s_waitcnt vmcnt(7) (fp64: two loads per column, vmcnt(14)) -
    compiler-visible loads had made the loop header drain everything (vmcnt(0)) every 8 steps, 250 cycles per step instead of
    ~130.  Pinned here: inside the loop, one such wait and one column load per step, and no full drain."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_resources.py"), "dnp_xie.hip"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = open("/tmp/isa/dnp_xie.hip.s").read()
    for mangled, younger, loads_per_step in (("_ZN3dnp22xie_block_solve_kernelIfEE", 7, 1), ("_ZN3dnp22xie_block_solve_kernelIdEE", 14, 2)):
        start = re.search(r"^" + mangled + r"\w*:", text, flags=re.M)
        assert start, mangled
        body = text[start.start():text.index(".end_amdhsa_kernel", start.start())]
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln]
        # the step loop = from the first loop label that is branched back to, to that branch
        labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":") and ln.startswith(".LBB")}
        back = [(labels[ln.split()[-1]], i) for i, ln in enumerate(lines)
                if ln.startswith("s_cbranch") and ln.split()[-1] in labels and labels[ln.split()[-1]] < i]
        # the unrolled step loop: the shortest backward branch whose body issues the 8 columns of an iteration
        cands = [be for be in back if sum(ln.startswith("global_load_dwordx4") for ln in lines[be[0]:be[1]]) >= 8 * loads_per_step]
        assert cands, (mangled, back)
        lo, hi = min(cands, key=lambda be: be[1] - be[0])
        loop = lines[lo:hi]
        waits = [ln for ln in loop if ln.startswith("s_waitcnt") and "vmcnt" in ln]
        assert len(waits) == 8 and all(f"vmcnt({younger})" in ln for ln in waits), (mangled, waits)
        assert sum(ln.startswith("global_load_dwordx4") for ln in loop) == 8 * loads_per_step, mangled
