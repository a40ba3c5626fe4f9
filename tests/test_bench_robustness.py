"""bench.py's guards for the first real multi-GPU run (round-4 verdict, item 4) - the parts that need no GPU: the host-side deadline and
the failure path leave ONE JSON line carrying "error" on rank 0's stdout and end the process with a fresh non-zero exit; other ranks
say so on stderr only.  (The collective timeout, the pre-check of the gathered matrix and a rank that dies are rehearsed on the GPU box:
tools/rehearse.sh -> profiles/r05_bench_gloo_rehearsal.txt.)"""
import json
import os
import subprocess
import sys

from conftest import ROOT

CODE = """
import sys, time
sys.path.insert(0, {root!r})
import bench
{body}
"""


def run(body):
    return subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, body=body)], capture_output=True, text=True, timeout=300)


def test_deadline_fires_with_an_error_line_and_a_nonzero_exit():
    out = run("d = bench.Deadline(0.3, 0, 4); d.phase = 'timed steps'; time.sleep(30)")
    assert out.returncode == 3
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 4 and "timed steps" in line["error"] and "not finished" in line["error"]
    out = run("d = bench.Deadline(0.3, 2, 4); time.sleep(30)")          # another rank: stderr only, same exit
    assert out.returncode == 3 and out.stdout.strip() == "" and "rank 2 of 4" in out.stderr


def test_deadline_cancelled_in_time_is_silent():
    out = run("d = bench.Deadline(0.5, 0, 1); d.cancel(); time.sleep(1.0); print('done')")
    assert out.returncode == 0 and out.stdout.strip() == "done"


def test_fail_prints_the_error_line_on_rank_zero_only():
    out = run("bench.fail(0, 8, 'pre-check of the gathered interaction matrix failed: x')")
    assert out.returncode == 4
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["value"] is None and "pre-check" in line["error"]
    out = run("bench.fail(3, 8, 'boom')")
    assert out.returncode == 4 and out.stdout.strip() == "" and "boom" in out.stderr


def test_collective_timeout_and_deadline_come_from_the_environment():
    env = dict(os.environ, BENCH_COLLECTIVE_TIMEOUT_S="15", BENCH_DEADLINE_S="77")
    out = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, body="print(bench.COLLECTIVE_TIMEOUT_S, bench.DEADLINE_S)")],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and out.stdout.split() == ["15.0", "77.0"]
