"""bench.py --gpus N without a launcher: the parent spawns N workers (one per GPU), relays rank 0's line, fails fast when
a worker dies (instead of leaving the others in the rendezvous) and refuses when the box has fewer GPUs.  CPU test with
a stand-in worker; the parent itself never touches a GPU."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_bench_worker.py")


def _run(mode, gpus, device_count):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import argparse, bench; "
            f"bench.spawn_workers(argparse.Namespace(gpus={gpus}), [{mode!r}], script={FAKE!r}, device_count={device_count})")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=100)
    return r, time.time() - t0


def test_spawn_relays_rank0_line():
    r, _ = _run("ok", 4, 8)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d == {"n_gpus": 4, "local_rank": "0", "master": "127.0.0.1"}


def test_spawn_fails_fast_when_a_worker_dies():
    r, dt = _run("fail-rank1", 2, 2)
    assert r.returncode == 1 and "worker(s) failed" in r.stderr and "(1, 3)" in r.stderr
    assert dt < 60, "the parent must end the surviving ranks, not wait for their rendezvous timeout"


def test_spawn_refuses_more_gpus_than_visible():
    r, _ = _run("ok", 2, 1)
    assert r.returncode == 2 and "only 1 GPU(s) visible" in r.stderr


def test_plain_command_on_a_box_without_gpus_exits_with_a_message():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr
