#!/usr/bin/env python3
"""Headline benchmark: dipole field-evaluations (source-target pairs) per second on the 100 000-point
synthetic sphere of BASELINE.json (config 4: 256 patches), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A STEP is one pass of the data-parallel hot path over the cloud: every one of the 256 per-patch
field evaluations `field_grad(pts[patch_k], pts[~patch_k])` (dnp_patch_fields_tiled_f32 - all of them in
one launch per rank, the per-tile interaction partials out of the same kernel's epilogue), the patch
interaction matrix rows (dnp_interactions_from_tiles) and, for N > 1, the RCCL all-gather of those rows
that hands every rank what the sequential greedy loop needs - IN THE LAUNCH STREAM'S ORDER, exactly as the
product's sharded driver issues it (parallel.sharded_patch_propagation -> patch_drivers._batched_patch_propagation:
slabs -> W rows -> synchronous parallel.gather_rows -> greedy loop; the greedy loop needs all of W, so the
driver cannot overlap the collective with anything).  `value` and `ms_per_step` are THAT step.  At N > 1 over
RCCL a second timed loop runs the same steps with the all-gather issued asynchronously (two in flight, each
overlapping the next step's pair kernel: what parallel.sharded_patch_propagation_many reaches when it is handed
several clouds) and reports it as `value_pipelined` / `ms_per_step_pipelined` - never as `value`; the gathered
matrix of the pipelined loop is compared with the in-order one (`pipelined_matches_in_order`).  After the timed
steps the gathered matrix is run through the device greedy loop and compared with the REFERENCE's own trace on
this cloud (tests/golden/G19).
Patches are sharded over the ranks in contiguous size-balanced blocks, so the total work is fixed
as N grows ("strong").  Inputs are resident in HBM before the timed region.  `value` counts the
algorithmic pairs sum_k |patch_k| * (N - |patch_k|) of all ranks per second of the slowest rank.

Also printed on the same JSON line:
  roofline      the pair kernel against the FP32 vector-ALU roofline that binds it (33 flop/pair,
                DESIGN.md), timed by HIP events recorded INSIDE the timed steps (mean / median / min;
                launch_ms <= ms_per_step is asserted); `hbm` carries the algorithmic HBM GB/s the metric
                name asks for (<< 1 % of 8 TB/s by construction).
  step_parts    pair kernel / interaction kernel / all-gather times of a step; per_rank at N > 1.
  sharded_driver (N > 1) the whole product call parallel.sharded_patch_propagation(diffuse=True) on this cloud: slabs,
                W rows, all-gather, greedy loop, fp64 combine, the all-reduce of the partial fields, tail - what a caller gets.
  clock_warmup  the untimed steps run before the W warm-up steps (80 ms of load: a device climbs to its
                sustained clock in ~50 ms, and at N ranks a step is N times shorter).
  cpu_baseline  the oracle's dense-broadcast PyTorch port of the reference path, timed on this
                box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import datetime
import json
import os
import signal
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.workloads import N_PATCHES, N_POINTS, fibonacci_patches, headline_workload, sphere_cloud  # noqa: E402,F401

FLOP_PER_PAIR = 33            # 3 sub, 5 r.r, 5 p.r, sqrt, 2 fma (4), rcp, 2 mul, 12 accumulate
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md (= the dense f32 MFMA peak)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md; v_fma_f64 measured at 4.3 cycles per wave64 instruction (profiles/r05_ubench_f64.txt)
HBM_PEAK_GBS = 8000.0
# a rank that dies must not leave the others hanging until the driver's limit (round-4 verdict); the rehearsal shortens it
COLLECTIVE_TIMEOUT_S = float(os.environ.get("BENCH_COLLECTIVE_TIMEOUT_S", "120"))
DEADLINE_S = float(os.environ.get("BENCH_DEADLINE_S", "900"))   # host-side deadline of the whole run


class Deadline:
    """Host-side watchdog: if the run is not finished `seconds` after start (a hung collective, a dead peer), rank 0 prints ONE
    JSON line carrying "error" - so the driver records a failure with its cause instead of a silence - every rank says so on
    stderr, and the process ends with a fresh non-zero exit (os._exit: no exec, no atexit handlers that could touch the GPU)."""

    def __init__(self, seconds, rank, world):
        self.rank, self.world, self.seconds = rank, world, seconds
        self.phase = "start"
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def _fire(self):
        msg = f"bench.py rank {self.rank} of {self.world}: not finished after {self.seconds:.0f} s (phase: {self.phase})"
        print(msg, file=sys.stderr, flush=True)
        if self.rank == 0:
            print(json.dumps({"metric": "dipole field-evals/sec (N x N pairs), 100k pts", "value": None, "unit": "pairs/s",
                              "n_gpus": self.world, "error": msg}), flush=True)
        os._exit(3)

    def cancel(self):
        self.timer.cancel()


def fail(rank, world, msg):
    """A wrong result is a failed run: rank 0 prints the JSON line with "error", every rank exits non-zero."""
    print(f"bench.py rank {rank}: {msg}", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"metric": "dipole field-evals/sec (N x N pairs), 100k pts", "value": None, "unit": "pairs/s",
                          "n_gpus": world, "error": msg}), flush=True)
    os._exit(4)


def cpu_baseline(pc_cpu, seconds_target=15.0):
    """Dense-broadcast PyTorch port of field_grad (oracle/dipole_oracle.py) on the host cores.
    The thread count is calibrated (a GPU box exposes far more logical CPUs than its share: running
    torch on all of them is several times slower than on 16), then a bounded sample of the same
    workload - all sources x the first n_t targets - is timed."""
    from oracle import dipole_oracle as O
    n = pc_cpu.shape[0]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best_rate, best_threads = 0.0, 1
    for th in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        O.field_grad(pc_cpu, pc_cpu[:256])
        rate = n * 256 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_rate, best_threads = rate, th
    torch.set_num_threads(best_threads)
    n_t = int(min(max(best_rate * seconds_target / n, 256), 12500))   # at most one recursion column
    t0 = time.perf_counter()
    O.field_grad(pc_cpu, pc_cpu[:n_t])
    dt = time.perf_counter() - t0
    return {"value": n * n_t / dt, "unit": "pairs/s", "cores": best_threads, "kind": "port",
            "sample": f"all {n} sources x first {n_t} targets of the same cloud ({n * n_t:.3g} pairs, {dt:.1f} s), "
                      f"dense-broadcast PyTorch fp32 with the reference's 15000-row leaf recursion, "
                      f"{best_threads} threads (best of 8/16/32/64 on {avail} visible CPUs)"}


def other_configs(dev, fu, util, pts_sorted, patch_ranges):
    """The remaining BASELINE configs, timed in the same run (N = 1 only; reported, not the headline):
    config 1 ok.xyz per-point propagation (fp32 file path and the fp64 socket path), config 2 fandisk all-pairs
    field, config 3 boxunion representative propagation (the reference's own 369 patches / representatives),
    config 4's whole greedy driver, config 5's source -> target transfer at S = T = 100 000 (fp32; the fp16
    wording of BASELINE is costed and rejected in DESIGN.md section 4)."""
    out = {}

    def timed(fn, reps):
        # untimed calls first: after the host-side preparation of a leg the clocks have dropped, and the first ~40 ms of
        # GPU work run up to 25 % slower (profiles/r02_kernel_trace_durations.txt)
        t_warm = time.perf_counter()
        for _ in range(3):
            fn()
            torch.cuda.synchronize()
        while time.perf_counter() - t_warm < 0.05:
            fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def timed_each(fn, reps):
        # long calls (the per-point propagation: ~30 ms each): every repetition timed on its own, synchronised - median, min
        # and max go on the line (round-3 verdict: 3 / 2 repetitions of this leg read 28, 35 and 47 ms in three runs)
        for _ in range(2):
            fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts)
        return float(np.median(ts)), float(ts.min()), float(ts.max())

    gdir = os.path.join(ROOT, "tests", "golden")
    try:
        fandisk = torch.from_numpy(np.load(os.path.join(gdir, "G5_fandisk_allpairs.npz"))["pc"]).to(dev)
        t = timed(lambda: fu.field_grad(fandisk, fandisk), 20)
        out["config2_fandisk_allpairs"] = {"points": int(fandisk.shape[0]), "us_per_call": t * 1e6,
                                           "pairs_per_s": fandisk.shape[0] ** 2 / t}
        # config 1: the cloud is resident in HBM before the timed region (as for every leg: the copy of a HOST tensor inside
        # the loop - torch's multi-threaded CPU clone under a CPU quota - was what made this leg read 28 ... 50 ms in round 3;
        # the kernel itself is steady within 5 %, profiles/r04_k4_spread.txt); 12 repetitions, each timed on its own
        ok = torch.from_numpy(np.load(os.path.join(gdir, "G8_point_propagation.npz"))["pc_full"]).to(dev)
        med, lo, hi = timed_each(lambda: fu.strongest_field_propagation_points(ok.clone(), diffuse=True), 12)
        out["config1_ok_point_propagation"] = {"points": int(ok.shape[0]), "ms": med, "ms_min": lo, "ms_max": hi, "repetitions": 12,
                                               "us_per_step": med / ok.shape[0] * 1e3}
        ok64 = ok.double()
        med, lo, hi = timed_each(lambda: fu.strongest_field_propagation_points(ok64.clone(), diffuse=True), 12)
        out["config1_ok_point_propagation_f64"] = {"points": int(ok.shape[0]), "ms": med, "ms_min": lo, "ms_max": hi,
                                                   "repetitions": 12, "us_per_step": med / ok.shape[0] * 1e3}
        g15 = np.load(os.path.join(gdir, "G15_boxunion_config3.npz"))
        cloud = torch.from_numpy(g15["pc"]).clone()
        cloud[~torch.from_numpy(g15["prefilter_sign"]), 3:] *= -1
        cloud = cloud.to(dev)
        i64 = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev)
        reps = util.RepLists(util.PatchList(i64(g15["rep_idx"]), np.diff(g15["rep_off"]), disjoint=True),
                             util.PatchList(i64(g15["rest_idx"]), np.diff(g15["rest_off"]), disjoint=True))
        t = timed(lambda: fu.strongest_field_propagation_reps(cloud.clone(), reps, diffuse=True), 10)
        nrep = int(g15["rep_off"][-1])
        out["config3_boxunion_reps_propagation"] = {
            "points": int(cloud.shape[0]), "patches": len(reps), "representatives": nrep, "ms": t * 1e3,
            "pairs": float(nrep) ** 2 + float(nrep) * (cloud.shape[0] - nrep),
            "note": "reference's partition/representatives (tests/golden/G15); reference CPU run: 360-440 s"}
        t = timed(lambda: util.divide_pc(cloud[:, :3], 41, min_patch=100), 5)
        out["config3_boxunion_partition_and_merge"] = {"ms": t * 1e3, "patches": len(reps)}
    except FileNotFoundError:
        pass
    t = timed(lambda: fu.strongest_field_propagation(pts_sorted.clone(), list(enumerate(patch_ranges)), patch_ranges,
                                                     diffuse=True), 5)
    out["config4_patch_driver_end_to_end"] = {"points": N_POINTS, "patches": N_PATCHES, "ms": t * 1e3}
    t = timed(lambda: fu.field_grad(pts_sorted, pts_sorted), 10)
    out["allpairs_100k_field_grad"] = {"points": N_POINTS, "ms": t * 1e3, "pairs_per_s": float(N_POINTS) ** 2 / t}
    g = torch.Generator().manual_seed(3)
    tgt = (pts_sorted[:, :3].cpu() + 1e-3 * torch.randn(N_POINTS, 3, generator=g)).to(dev)
    t = timed(lambda: fu.reference_field(pts_sorted, tgt), 10)
    out["config5_reference_field_100k_to_100k"] = {"sources": N_POINTS, "targets": N_POINTS, "ms": t * 1e3,
                                                   "pairs_per_s": float(N_POINTS) ** 2 / t, "dtype": "f32"}
    # ---- the float64 entry points (round 5; the reference's socket path hands float64 clouds to the same functions): pairs/s
    # and the fraction of the FP64 vector peak at the 33 flop per pair of the fp32 roofline
    def f64_leg(pairs, t, flop=FLOP_PER_PAIR):
        return {"ms": t * 1e3, "pairs_per_s": pairs / t, "dtype": "f64", "flop_per_pair": flop,
                "frac_of_fp64_valu_peak": pairs * flop / t / 1e12 / FP64_VALU_PEAK_TFLOPS}
    grid = util.gen_grid().to(dev)
    t = timed(lambda: fu.potential(pts_sorted, grid), 200)       # 60-us calls: a window long enough to dilute one host hiccup
    out["potential_100k_x_1000_lattice"] = {"ms": t * 1e3, "pairs_per_s": float(N_POINTS) * grid.shape[0] / t, "dtype": "f32"}
    pts64, tgt64, grid64 = pts_sorted.double(), tgt.double(), grid.double()
    out["potential_100k_x_1000_lattice_f64"] = f64_leg(float(N_POINTS) * grid.shape[0], timed(lambda: fu.potential(pts64, grid64), 200), 13)
    out["allpairs_100k_field_grad_f64"] = f64_leg(float(N_POINTS) ** 2, timed(lambda: fu.field_grad(pts64, pts64), 3))
    out["config5_reference_field_100k_to_100k_f64"] = f64_leg(float(N_POINTS) ** 2, timed(lambda: fu.reference_field(pts64, tgt64), 3))
    t = timed(lambda: fu.strongest_field_propagation(pts64.clone(), list(enumerate(patch_ranges)), patch_ranges, diffuse=True), 3)
    out["config4_patch_driver_end_to_end_f64"] = dict(f64_leg(float(N_POINTS) ** 2, t), points=N_POINTS, patches=N_PATCHES,
                                                      note="float64 cloud: fp64 slabs, W, combine and tail (dnp_patch_fields_tiled_f64 ...)")
    del pts64, tgt64, grid64
    # the fork's ordered propagation (the socket path's xie_propagation_points_onbfstree runs it once per route): the
    # interaction matrix, the blocked ordered loop and the diffuse pass for 5 visiting orders of 10 000 points, orders uploaded per call
    nx = 10000
    xg = torch.randn(nx, 6, generator=torch.Generator().manual_seed(5))
    xpc = torch.cat([0.4 * xg[:, :3] / xg[:, :3].norm(dim=1, keepdim=True), torch.nn.functional.normalize(xg[:, 3:], dim=1)], 1).to(dev)
    xorders = np.stack([np.random.default_rng(s).permutation(nx) for s in range(5)])
    for name, cloud in (("xie_ordered_propagation_10k_points_5_orders", xpc), ("xie_ordered_propagation_10k_points_5_orders_f64", xpc.double())):
        t = timed(lambda: fu.xie_propagation_points_in_order(cloud, 0.1, xorders, diffuse=True), 10)
        out[name] = {"ms": t * 1e3, "points": nx, "orders": 5, "us_per_step": t * 1e6 / nx,
                     "dtype": "f64" if cloud.dtype == torch.float64 else "f32", "diffuse": True}
    del xpc
    # the same two calls with HOST tensors in and out (the reference's functions take either): H2D + D2H of the cloud
    # over PCIe inside the timed call.  Never the headline value (inputs resident in HBM there) - DESIGN.md section 5.
    host = pts_sorted.cpu()       # oriented in place call after call (a torch CPU clone of 2.4 MB inside the timed loop costs
    #                               3-50 ms where the process sees more cores than its CPU quota - not the path's time)
    t = timed(lambda: fu.strongest_field_propagation(host, list(enumerate(patch_ranges)), patch_ranges,
                                                     diffuse=True), 5)
    out["config4_patch_driver_host_tensors"] = {"points": N_POINTS, "patches": N_PATCHES, "ms": t * 1e3,
                                                "pairs_per_s_pcie_inclusive": float(N_POINTS) ** 2 / t}
    t = timed(lambda: fu.field_grad(host, host), 5)
    out["allpairs_100k_field_grad_host_tensors"] = {"points": N_POINTS, "ms": t * 1e3,
                                                    "pairs_per_s_pcie_inclusive": float(N_POINTS) ** 2 / t}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    # 10 untimed steps by default: the first ~8 launches after an idle period run up to 25 % slower while the clocks
    # ramp (5.3, 5.1, 4.8, 4.7, 4.6, 4.4, 4.4, 4.3 ms, then 4.26 +- 0.03; profiles/r02_kernel_trace_durations.txt)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the other_configs legs (profiling runs: only the headline kernels in the trace)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    if world > 1:
        # N ranks on one node's cores: torch's default (every core it sees, in every rank) makes the small host-side torch ops of
        # the set-up thrash - five such ranks on a 16-core share spent 180 s building this workload (tools/gpu_dist_probe.py).
        # torch.distributed.run sets OMP_NUM_THREADS=1 itself; this covers every other launcher.
        torch.set_num_threads(max(1, min(torch.get_num_threads(), (os.cpu_count() or world) // world, 4)))
    # one rank per GPU; BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # BENCH_ONE_RANK_RCCL=1 (one GPU, no launcher): a ONE-rank nccl group, so that the pipelined loop's asynchronous all-gather
    # runs through RCCL itself on a single-GPU box (rehearsal of that code path; the line says "rehearsal")
    one_rank_rccl = world == 1 and bool(os.environ.get("BENCH_ONE_RANK_RCCL"))
    deadline = Deadline(DEADLINE_S, rank, world)
    deadline.phase = "init_process_group"
    if world > 1 and rank == 0:
        # torch.distributed.run answers a rank that failed by sending SIGTERM to the others: rank 0 then still leaves its line
        def _terminated(signum, frame):
            print(json.dumps({"metric": "dipole field-evals/sec (N x N pairs), 100k pts", "value": None, "unit": "pairs/s",
                              "n_gpus": world, "error": f"terminated by the launcher (signal {signum}) in phase '{deadline.phase}': "
                              "a peer rank failed or the run was cancelled"}), flush=True)
            os._exit(6)
        signal.signal(signal.SIGTERM, _terminated)
    ctimeout = datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S)     # instead of the default 10 minutes per collective
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=ctimeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=ctimeout)
    elif one_rank_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=ctimeout)
    if world > 1 and os.environ.get("BENCH_TEST_DIE_RANK") == str(rank):
        # rehearsal hook (tools/rehearse.sh): this rank leaves before its first collective - the others must end with an
        # "error" line within the collective timeout, not hang
        print(f"bench.py rank {rank}: BENCH_TEST_DIE_RANK set, leaving", file=sys.stderr, flush=True)
        os._exit(7)

    from dipole_normal_prop_amd import field_utils as fu
    from dipole_normal_prop_amd import parallel, util

    # whole patches flipped at random (seed 0): the propagation has 256 sign decisions to get right, checked below
    pc_cpu, patches, scramble = headline_workload()
    sizes = np.array([len(p) for p in patches])
    pairs_total = float((sizes * (N_POINTS - sizes)).sum())
    # layout: the cloud sorted by patch (what the drivers do, patch_drivers._batched_patch_propagation), so a
    # patch is a contiguous row range and every slab / interaction access is coalesced
    off, idx, _ = util.patch_csr(patches, dev)
    pts = pc_cpu.to(dev)[idx].contiguous()
    point_patch = torch.repeat_interleave(torch.arange(N_PATCHES, device=dev), off[1:] - off[:-1])
    idx = None
    boxes = fu._patch_boxes(pts, off, idx)       # per-cloud set-up like the CSR offsets (the drivers do the same once)
    tiles = fu._TileTables(pts, sizes)           # target-tile boxes; W rows out of the pair kernel's epilogue if tiles allow
    bounds = fu._balanced_blocks(sizes, world)
    p_lo, p_hi = int(bounds[rank]), int(bounds[rank + 1])
    fake = int(os.environ.get("BENCH_FAKE_WORLD", "0"))      # developer aid: time one rank's share of an N-rank run
    if fake > 1 and world == 1:
        fb = fu._balanced_blocks(sizes, fake)
        p_lo, p_hi = int(fb[0]), int(fb[1])
    my_pairs = float((sizes[p_lo:p_hi] * (N_POINTS - sizes[p_lo:p_hi])).sum())
    # the drivers' launch plan (patch_drivers._launch_plan): longest patches first, a split tail when even the shortest are long
    order, split = fu._launch_plan(sizes[p_lo:p_hi], N_POINTS, dev)
    # The headline step gathers in the launch stream's order (what the product's driver does).  At N > 1 over RCCL a second
    # timed loop overlaps the all-gather with the next step's pair kernel (BENCH_NO_PIPELINED=1 skips it; BENCH_PIPELINED=1
    # forces it on other backends / one rank, where gather_rows_async falls back to the in-order form).
    can_pipeline = ((world > 1 and backend == "nccl" and not os.environ.get("BENCH_NO_PIPELINED")) or bool(os.environ.get("BENCH_PIPELINED"))
                    or one_rank_rccl)
    pending = []
    overlap_state = {"on": False, "async_seen": False}

    # HIP events on the stream the kernels are launched on (torch's current stream = what _lib.current_stream() hands
    # to the C ABI), recorded INSIDE the timed steps: the pair kernel's duration, the interaction kernel's and the
    # all-gather's come from the same launches, buffers and clocks as ms_per_step
    def step(marks=None):
        if marks is not None:
            marks[0].record()
        if tiles.fused:              # what the drivers do (patch_drivers._slabs_and_rows), opened up for the event marks
            w_part = torch.empty((p_hi - p_lo, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=dev)
            dE = fu._patch_slabs(pts, off, idx, point_patch, p_lo, p_hi, 1e-5, boxes, tiles.boxes, w_part, split, order)
        else:
            dE = fu._patch_slabs(pts, off, idx, point_patch, p_lo, p_hi, 1e-5, boxes, tiles.boxes, None, split, order)
        if marks is not None:
            marks[1].record()
        if tiles.fused:
            W = torch.empty((p_hi - p_lo, N_PATCHES), dtype=torch.float64, device=dev)
            fu._lib.check(fu._lib.require_device().dnp_interactions_from_tiles(
                fu._lib.ptr(w_part), tiles.slots, p_hi - p_lo, N_POINTS, fu._lib.ptr(point_patch), fu._lib.ptr(off), N_PATCHES,
                fu._lib.ptr(W), fu._lib.current_stream()))
        else:
            W = fu._interaction_rows(dE, pts, off, idx)
        if marks is not None and marks[2] is not None:
            marks[2].record()
        if overlap_state["on"]:
            # pipelined loop only: the all-gather of this step's W rows runs on RCCL's stream and overlaps the NEXT step's pair
            # kernel (the steps are independent batches); two in flight at most: the gather of step i - 2 is consumed before
            # step i issues its own
            try:
                W, work = parallel.gather_rows_async(W, bounds, force=one_rank_rccl)
            except Exception as exc:           # never lose a scaling run to the overlap: fall back to the in-order gather
                print(f"bench: asynchronous all-gather unavailable ({exc!r}); continuing with the in-order form", file=sys.stderr)
                overlap_state["on"] = False
                W, work = parallel.gather_rows(W, bounds), None
            if work is not None:
                overlap_state["async_seen"] = True
                pending.append((W, work))
                while len(pending) > 2:
                    pending.pop(0)[1].wait()
        else:
            W = parallel.gather_rows(W, bounds)
        if marks is not None and marks[2] is not None:
            marks[3].record()
        return W

    def fence():
        while pending:                         # every all-gather issued so far is complete before the barrier
            pending.pop(0)[1].wait()
        if world > 1:
            # device_ids: under nccl a barrier without it guesses the device from the rank
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def check_against_g19(Wfull):
        """The greedy loop on the gathered matrix against the REFERENCE's own run on this cloud (tests/golden/G19: visit order, flips
        and chosen interactions of field_utils.strongest_field_propagation, start patch = the reference's), and the patch
        scramble undone: every scrambled patch ends with one sign, every untouched patch with the other."""
        g19_path = os.path.join(ROOT, "tests", "golden", "G19_headline_sphere_patch_propagation.npz")
        g19 = np.load(g19_path) if os.path.exists(g19_path) else None
        start = int(g19["order"][0]) if g19 is not None else 0
        order, sigma, chosen = fu._greedy_on_device(Wfull, torch.full((1,), start, dtype=torch.int64, device=dev))
        sigma = sigma.cpu().numpy()
        flipped_sign, kept_sign = sigma[scramble], sigma[~scramble]
        s_ok = bool(np.all(flipped_sign == flipped_sign[0]) and np.all(kept_sign == kept_sign[0]) and flipped_sign[0] == -kept_sign[0])
        t_ok = c_dev = None
        if g19 is not None:
            order = order.cpu().numpy()
            t_ok = bool(np.array_equal(order, g19["order"]) and np.array_equal(sigma[order[1:]] < 0, g19["flipped"][1:]))
            c_dev = float(np.max(np.abs(chosen.cpu().numpy() - g19["chosen"]) / np.abs(g19["chosen"])))
        return s_ok, t_ok, c_dev

    # ---- before anything is timed (N > 1; round-4 verdict: the first run with more than one RCCL rank must not be trusted
    # blindly): every rank's rows of the gathered matrix are the rows it computed, the matrix reproduces the reference's trace,
    # and every rank answered the collective
    precheck = None
    if world > 1 or one_rank_rccl:
        deadline.phase = "precheck (first collectives)"
        torch.cuda.synchronize()
        w_part0 = torch.empty((p_hi - p_lo, tiles.n_tiles, tiles.slots), dtype=torch.float64, device=dev) if tiles.fused else None
        dE0 = fu._patch_slabs(pts, off, idx, point_patch, p_lo, p_hi, 1e-5, boxes, tiles.boxes, w_part0, split, order)
        if tiles.fused:
            W_local = torch.empty((p_hi - p_lo, N_PATCHES), dtype=torch.float64, device=dev)
            fu._lib.check(fu._lib.require_device().dnp_interactions_from_tiles(
                fu._lib.ptr(w_part0), tiles.slots, p_hi - p_lo, N_POINTS, fu._lib.ptr(point_patch), fu._lib.ptr(off), N_PATCHES,
                fu._lib.ptr(W_local), fu._lib.current_stream()))
        else:
            W_local = fu._interaction_rows(dE0, pts, off, idx)
        del dE0
        Wg = parallel.gather_rows(W_local, bounds)
        rows_ok = tuple(Wg.shape) == (N_PATCHES, N_PATCHES) and bool(torch.equal(Wg[p_lo:p_hi], W_local))
        s_ok, t_ok, _ = check_against_g19(Wg) if rows_ok else (False, False, None)
        cpu_side = not (world > 1 and backend == "nccl")
        flags = torch.tensor([1 if rows_ok else 0, 1 if (s_ok and t_ok is not False) else 0], dtype=torch.int32, device="cpu" if cpu_side else dev)
        seen = torch.zeros(max(world, 1), dtype=torch.int32, device="cpu" if cpu_side else dev)
        seen[rank] = 1
        if world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        precheck = {"gathered_rows_equal_local_rows_on_every_rank": bool(int(flags[0].item())),
                    "trace_matches_reference_G19_on_every_rank": bool(int(flags[1].item())),
                    "ranks_seen": int((seen > 0).sum().item())}
        if not (precheck["gathered_rows_equal_local_rows_on_every_rank"] and precheck["trace_matches_reference_G19_on_every_rank"]
                and precheck["ranks_seen"] == max(world, 1)):
            fail(rank, world, f"pre-check of the gathered interaction matrix failed: {precheck}")
        del Wg, W_local, w_part0
    deadline.phase = "warm-up"

    # Clock ramp: after an idle phase the first ~40-60 ms of GPU work run up to 25 % slower (the device climbs to its
    # sustained clock under load; profiles/r02_kernel_trace_durations.txt, and for an eighth-size step 30 launches after 20
    # warm-up launches still fall from 0.66 to 0.57 ms).  The W warm-up steps cover that at N = 1 (10 x 4.1 ms); at N ranks a
    # step is N times shorter, so a fixed number of them does not.  Before the W steps the bench therefore runs untimed
    # steps until CLOCK_WARMUP_MS of wall time have passed - the same number on every rank (decided by rank 0's clock,
    # broadcast), outside the timed region, reported on the JSON line.
    CLOCK_WARMUP_MS = 80.0
    for _ in range(8):             # absorbs lazy initialisation (RCCL communicator, allocator, code objects): not load
        W = step()
    fence()
    pre_steps = 8
    t_pre = time.perf_counter()
    while True:
        for _ in range(8):
            W = step()
        pre_steps += 8
        torch.cuda.synchronize()
        done = torch.tensor([1 if (time.perf_counter() - t_pre) * 1e3 >= CLOCK_WARMUP_MS else 0], dtype=torch.int32,
                            device=dev if (world > 1 and backend == "nccl") else "cpu")
        if world > 1:
            dist.broadcast(done, src=0)
        if int(done.item()):
            break
    for _ in range(args.warmup):
        W = step()
    # every step carries the two marks around the pair kernel; the interaction / all-gather marks only every 4th step at
    # N = 1 (an event costs a few microseconds of stream time, and there is no collective to watch), every step at N > 1
    def marks_for(i):
        detail = world > 1 or i % 4 == 0
        return [torch.cuda.Event(enable_timing=True) if (j < 2 or detail) else None for j in range(4)]
    ev = [marks_for(i) for i in range(args.steps)]
    deadline.phase = "timed steps"
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        W = step(ev[i])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    W_in_order = W

    # ---- the same steps with the all-gather overlapped (N > 1 over RCCL): reported beside the headline, never as it ---
    pipelined = None
    if can_pipeline:
        deadline.phase = "pipelined steps (asynchronous all-gather)"
        overlap_state["on"] = True
        for _ in range(max(args.warmup, 4)):
            W = step()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            W = step()
        fence()
        el2 = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el2 = float(tmax.item())
        same = bool(torch.equal(W, W_in_order))
        if world > 1:
            ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            same = bool(int(ok.item()))
        pipelined = {"ms_per_step": el2 / args.steps * 1e3, "asynchronous": overlap_state["async_seen"], "matches_in_order": same}
        overlap_state["on"] = False
        W = W_in_order
    # developer aid BENCH_FAKE_WORLD: only one rank's share was computed, so only that share is credited
    value = (my_pairs if (fake > 1 and world == 1) else pairs_total) * args.steps / elapsed

    # ---- roofline of the dominant kernel, from the events of the timed steps ---------------------------------
    k_all = np.array([m[0].elapsed_time(m[1]) for m in ev])
    k3_all = np.array([m[1].elapsed_time(m[2]) for m in ev if m[2] is not None])
    ag_all = np.array([m[2].elapsed_time(m[3]) for m in ev if m[2] is not None])
    k_ms, k_med, k_min = float(k_all.mean()), float(np.median(k_all)), float(k_all.min())
    # the launch is part of the step: its mean over the timed steps cannot exceed the mean step of the slowest rank
    assert k_ms <= ms_per_step * 1.0005, f"pair-kernel launch {k_ms:.4f} ms > step {ms_per_step:.4f} ms: timing is inconsistent"
    # flops: the kernel also evaluates (and zeroes) the pairs inside each source patch
    launch_pairs = float((sizes[p_lo:p_hi] * N_POINTS).sum())
    tflops = launch_pairs * FLOP_PER_PAIR / (k_ms * 1e-3) / 1e12
    algo_bytes = 36.0 * N_POINTS + 12.0 * N_POINTS * (p_hi - p_lo)   # cloud read once + [K,N,3] slab written once
    roofline = {"bound": "valu", "achieved": tflops,
                "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VALU_PEAK_TFLOPS,
                "traffic": None, "launch_ms": k_ms, "launch_ms_median": k_med, "launch_ms_min": k_min,
                "flop_per_pair": FLOP_PER_PAIR, "pairs_per_launch": launch_pairs, "source_split": split,
                "launch_order": "longest patch first" if order is not None else "patch order",
                "timed": f"HIP events around dnp_patch_fields_tiled_f32 (one pair_kernel_scalar launch) on torch's current "
                         f"stream, recorded inside the {args.steps} timed steps; achieved = 33 flop x pairs_per_launch / "
                         f"mean launch_ms",
                "note": "FP32 vector ALU binds (no MFMA on this path); peak equals the dense f32 MFMA peak; 33 flop "
                        "per pair is the exact chain's count - pairs that take the far-field chain execute 45"}
    hbm = {"bound": "hbm", "achieved": algo_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes,
           "traffic": None}
    step_parts = {"pair_kernel_ms": k_ms, "interactions_kernel_ms": float(k3_all.mean()),
                  "interactions_from": "pair-kernel epilogue partials + dnp_interactions_from_tiles" if tiles.fused
                  else "dnp_interactions_f32 (second pass over the slabs)",
                  "gather_rows_ms": float(ag_all.mean()),
                  "gather_rows_timed": "in the launch stream's order (the product's sharded driver)"}
    per_rank = None
    if world > 1:
        # so that a scaling run explains itself: every rank's kernel time and its wait in the all-gather
        mine = torch.tensor([k_ms, k_min, float(k3_all.mean()), float(ag_all.mean()), float(ag_all.min()), my_pairs],
                            dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "pair_kernel_ms": float(t[0]), "pair_kernel_ms_min": float(t[1]),
                     "interactions_kernel_ms": float(t[2]), "gather_rows_ms": float(t[3]),
                     "gather_rows_ms_min": float(t[4]), "pairs": float(t[5])} for r, t in enumerate(allr)]
    # HBM traffic cannot be read inside the run (PMC counters need rocprofv3): it is PROFILE-DERIVED, from the
    # committed summary of `rocprofv3 --pmc` passes over this same command, and says so
    for prof_name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        prof = os.path.join(ROOT, "profiles", prof_name)
        if os.path.exists(prof) and world == 1 and not fake:
            try:
                tr = json.load(open(prof))
                roofline["traffic"] = hbm["traffic"] = tr.get("hbm_bytes_per_launch")
                roofline["traffic_source"] = hbm["traffic_source"] = f"profile-derived: profiles/{prof_name}"
                roofline["kernel"] = tr.get("kernel")
                break
            except Exception:
                pass

    # sanity: the gathered matrix must be the full P x P on every rank, and the greedy loop on it must reproduce the
    # REFERENCE's own run on this cloud (tests/golden/G19: visit order, flips and the chosen interactions of
    # field_utils.strongest_field_propagation, start patch = the reference's) and undo the patch scramble: every
    # scrambled patch ends with one sign, every untouched patch with the other
    signs_ok = None
    trace_matches_reference = chosen_dev = None
    if not (fake > 1 and world == 1):
        assert W.shape == (N_PATCHES, N_PATCHES)
        signs_ok, trace_matches_reference, chosen_dev = check_against_g19(W)

    # ---- N > 1: the whole product call, every rank in it (what a caller of parallel.sharded_patch_propagation gets) ------
    sharded_driver = None
    deadline.phase = "sharded driver leg / other configs"
    if world > 1 and not args.headline_only:
        ranges = util.PatchList(torch.arange(N_POINTS, device=dev), sizes, disjoint=True)
        plist = list(enumerate(ranges))

        def call():
            parallel.sharded_patch_propagation(pts.clone(), plist, ranges, diffuse=True)
        for _ in range(3):
            call()
        fence()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        fence()
        el3 = time.perf_counter() - t0
        tmax = torch.tensor([el3], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tr = fu.last_trace("sharded")
        ok_trace = None
        g19_path = os.path.join(ROOT, "tests", "golden", "G19_headline_sphere_patch_propagation.npz")
        if os.path.exists(g19_path):
            ok_trace = bool(np.array_equal(np.asarray(tr["order"]), np.load(g19_path)["order"]))
        sharded_driver = {"call": "parallel.sharded_patch_propagation(pts, patches, patches, diffuse=True)", "ms": float(tmax.item()) / reps * 1e3,
                          "repetitions": reps, "includes": "layout, box tables, slabs + W rows, all-gather, greedy loop, fp64 combine, "
                          "all-reduce of the partial fields, tail", "order_matches_reference_G19": ok_trace}

    out = None
    if rank == 0:
        out = {"metric": "dipole field-evals/sec (N x N pairs), 100k pts", "value": value, "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "clock_warmup": {"ms": CLOCK_WARMUP_MS, "untimed_steps_before_the_warmup_steps": pre_steps},
               "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": "synthetic 100k-point sphere (seed 1234), 256 Fibonacci patches (whole patches "
                                      "sign-scrambled, seed 0), all per-patch fields + interaction matrix (BASELINE "
                                      "config 4)", "points": N_POINTS,
                          "patches": N_PATCHES, "pairs_per_step": pairs_total,
                          "parallelism": f"patch-sharded x{world}, " + ("RCCL" if backend == "nccl" else backend) +
                                         " all-gather of W rows, in the launch stream's order"},
               "roofline": roofline, "hbm": hbm, "step_parts": step_parts, "signs_ok": signs_ok,
               "trace_matches_reference_G19": trace_matches_reference, "chosen_max_rel_dev_vs_G19": chosen_dev}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if precheck is not None:
            # checked BEFORE the timed steps: a failure there ends the run with an "error" line instead of a number
            out["precheck"] = precheck
            out["rccl_ranks_seen" if backend == "nccl" else "ranks_seen"] = precheck["ranks_seen"]
        out["collective_timeout_s"] = COLLECTIVE_TIMEOUT_S if (world > 1 or one_rank_rccl) else None
        if pipelined is not None:
            # NOT the headline: the step with its all-gather overlapped with the next step's pair kernel - what a caller that
            # hands parallel.sharded_patch_propagation_many several clouds reaches; one cloud at a time cannot (the greedy
            # loop needs all of W)
            pv = (my_pairs if (fake > 1 and world == 1) else pairs_total) / (pipelined["ms_per_step"] * 1e-3)
            out["value_pipelined"] = pv if pipelined["asynchronous"] else None
            out["ms_per_step_pipelined"] = pipelined["ms_per_step"] if pipelined["asynchronous"] else None
            out["pipelined_matches_in_order"] = pipelined["matches_in_order"]
            out["pipelined_note"] = ("all-gather of step i overlaps the pair kernel of step i + 1 (two in flight)" if pipelined["asynchronous"]
                                     else "this backend stages through the host: the asynchronous form fell back to the in-order gather")
        if sharded_driver is not None:
            out["sharded_driver"] = sharded_driver
        if fake > 1 and world == 1:
            out["fake_world"] = fake          # NOT a measurement of `fake` GPUs: one rank's share on one GPU
        if one_rank_rccl:
            out["rehearsal"] = "ONE-rank nccl group on one GPU: exercises the asynchronous RCCL all-gather of the pipelined loop, not a scaling number"
        if world > 1 and backend != "nccl":
            out["rehearsal"] = (f"{backend} collectives; ranks may share a GPU (device_count "
                                f"{torch.cuda.device_count()}): a functional rehearsal of the N>1 path, not a scaling number")
        if world == 1 and not (fake > 1) and not args.headline_only:
            ranges = util.PatchList(torch.arange(N_POINTS, device=dev), sizes, disjoint=True)
            out["other_configs"] = other_configs(dev, fu, util, pts, ranges)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pc_cpu)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        fu.flush_warnings()
        print(json.dumps(out), flush=True)
    deadline.phase = "shutdown"
    if world > 1:
        dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    elif one_rank_rccl:
        dist.destroy_process_group()
    deadline.cancel()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:      # a collective that timed out, a peer that died, a failed launch: say so on the JSON line
        import traceback
        traceback.print_exc()
        _rank, _world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if _rank == 0:
            print(json.dumps({"metric": "dipole field-evals/sec (N x N pairs), 100k pts", "value": None, "unit": "pairs/s",
                              "n_gpus": _world, "error": f"{type(exc).__name__}: {exc}"[:2000]}), flush=True)
        os._exit(5)
