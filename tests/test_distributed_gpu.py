"""The multi-GPU orchestration (gst-plugin-rs_amd/distributed.py) with the REAL pieces on the one-GPU box: per-rank compute
= the HIP band kernels through the C ABI, collective = RCCL (torch.distributed backend "nccl", world_size 1), in a fresh
process like a bench.py worker (tests/rccl_world1_worker.py).  The world-2 logic is covered on CPU by
tests/test_distributed_cpu.py (gloo); this test makes sure the RCCL communicator is really initialised by this code and
that HIP kernels and all-reduces are ordered correctly on one stream.  Results are compared with the oracle here."""
import json
import os
import subprocess
import sys

import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def worker_result(gpu):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-3000:]
    return json.loads(line[-1][len("RESULT "):])


def test_rccl_communicator_is_initialised(worker_result):
    assert worker_result["backend"] == "nccl" and worker_result["rccl_ranks"] == 1


def test_videocompare_sharded_hip_bands_over_rccl(worker_result):
    """band kernels -> ONE all-reduce of n_pads x 64 sums -> bits + Hamming on every rank."""
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    pads = [a, b, 255 - a, a]
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in pads]
    want = [float(orc.hamming(hs[0], x)) for x in hs[1:]]
    assert worker_result["videocompare"] == want and want[2] == 0.0
    # mvfx_videocompare_sharded_distances over the library's own communicator: ncclAllReduce inside the C entry, bits on the device
    assert worker_result["videocompare_c_entry"] == want
    assert worker_result["hashes_c_entry"] == hs
    assert worker_result["allreduce_f64_world1"] == [float(i) for i in range(10)]


def test_ssim_and_colordetect_sharded_over_rccl(worker_result):
    w, h = 320, 240
    a = frames.random_frame(0x5EED0002, w, h)
    b = a.copy()
    b[5::7, 3:w * 4:11] ^= 0x15
    rc, want, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    assert rc == 0 and worker_result["ssim"] == pytest.approx(want, rel=1e-5, abs=2e-9)  # the default f32 pipeline
    # mvfx_videocompare_sharded_dssim: the two all-reduces inside the library (RCCL, world 1) and without a communicator -- same bits
    assert worker_result["ssim_c_entry"] == worker_result["ssim"] == worker_result["ssim_c_entry_no_comm"]
    rc, want_pal = orc.colordetect_palette(a, "RGBA", 10, 5)
    assert rc >= 0 and worker_result["palette"] == [int(x) for x in want_pal]


# ---------------------------------------------------------------------------------------------------------------- world 2
# Two ranks on the box's ONE GPU (own process and HIP context each), every rank holding only its share of the inputs, HIP
# kernels for the per-rank compute, gloo for the collectives (RCCL refuses two ranks on one device): the N > 1 orchestration
# with the real kernels, which the CPU tests (oracle compute) and the world-1 RCCL test (one rank) each cover only half of.

@pytest.fixture(scope="module")
def world2_results(gpu):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hip_world2_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    results = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
        assert line, out[-3000:]
        results.append(json.loads(line[-1][len("RESULT "):]))
    return sorted(results, key=lambda r: r["rank"])


def test_world2_every_rank_holds_a_different_share(world2_results):
    r0, r1 = world2_results
    assert (r0["world"], r1["world"]) == (2, 2)
    assert r0["band"] == [0, 540] and r1["band"] == [540, 1080]
    assert r0["ssim_band"][1] == r1["ssim_band"][0] and r0["ssim_band"][0] == 0 and r1["ssim_band"][1] == 240
    assert r0["samples"][0] + r0["samples"][1] == r1["samples"][0] and sum(r["samples"][1] for r in world2_results) == 7680


def test_world2_videocompare_hip_bands(world2_results):
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in (a, b, 255 - a, a)]
    want = [float(orc.hamming(hs[0], x)) for x in hs[1:]]
    assert world2_results[0]["videocompare"] == want and world2_results[1]["videocompare"] == want


def test_world2_ssim_and_colordetect(world2_results):
    w, h = 320, 240
    a = frames.random_frame(0x5EED0002, w, h)
    b = a.copy()
    b[5::7, 3:w * 4:11] ^= 0x15
    rc, want, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    rc2, want_pal = orc.colordetect_palette(a, "RGBA", 10, 5)
    assert rc == 0 and rc2 >= 0
    for r in world2_results:
        assert r["ssim"] == pytest.approx(want, rel=1e-5, abs=2e-9)  # the default f32 pipeline
        assert r["palette"] == [int(x) for x in want_pal]
    assert world2_results[0]["ssim"] == world2_results[1]["ssim"]      # every rank derives the same value
    # unequal bands: 250 rows -> [0, 128) and the remainder rank's [128, 250); the sharded value is the whole frame's
    assert [r["ssim_uneven_band"] for r in world2_results] == [[0, 128], [128, 250]]
    ua = frames.random_frame(0x5EED0007, w, 250)
    ub = ua.copy()
    ub[3::5, 1:w * 4:13] ^= 0x21
    rc3, want_u, _ = orc.ssim_distance(ua, ub, w, 250, w * 4, w * 4, "RGBA")
    assert rc3 == 0
    for r in world2_results:
        assert r["ssim_uneven"] == pytest.approx(want_u, rel=1e-5, abs=2e-9)
    assert world2_results[0]["ssim_uneven"] == world2_results[1]["ssim_uneven"]


def test_bench_two_ranks_control_flow_on_one_gpu(gpu, tmp_path):
    """The N > 1 branch of bench.py has only ever been read, never run: the driver's scaling run is its first execution on real
    hardware.  Here the launcher's exact command (`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`) runs with both ranks
    on the one GPU and the rendezvous over gloo (MVFX_BENCH_TEST_SHARED_GPU=1; RCCL refuses two ranks on one device): barriers, the
    max-over-ranks timing, the whole-job aggregate and the watchdog around the band-sharded leg (whose RCCL communicator cannot
    come up here: it must report an error, not cost the line).  Not a measurement."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["MVFX_BENCH_TEST_SHARED_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--pool", "6",
           "--settle-seconds", "0.1", "--content-sweep", "0", "--side-leg-timeout", "60"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 3, r.stdout[-2000:]  # rank 0 prints the two sharded side legs as their own small lines, then THE line
    assert all(len(ln) < 3000 for ln in lines)
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]
    assert c["rccl_ranks"] == 2 and c["rendezvous_backend"] == "gloo"
    assert len(c["per_rank_frames_per_sec"]) == 2
    # whole-job value = frames of both ranks / the slowest rank's time: never more than the sum of the per-rank rates
    assert d["value"] <= sum(c["per_rank_frames_per_sec"]) * 1.0001
    assert d["roofline"]["bound"] in ("valu", "hbm") and d["roofline"]["frac"] > 0
    subs = {json.loads(ln)["sub"]: json.loads(ln) for ln in lines[:-1]}
    for key in ("videocompare_blockhash_sharded", "videocompare_dssim_sharded"):
        side = subs[key]
        assert "error" in side or side["n_gpus"] == 2, key
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_a_rank_stuck_in_the_side_leg_fails_the_run_after_the_line(gpu, tmp_path):
    """a rank that never comes back from the sharded leg (here: rank 1 sleeps, MVFX_BENCH_TEST_STUCK_RANK) is a FAILED
    run.  The ranks agree over the rendezvous store who is stuck, rank 0 still prints the headline line -- with `config.error` -- and
    every rank leaves with exit code 5, so the launcher reports failure (round 4 left with 0)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MVFX_BENCH_TEST_SHARED_GPU="1", MVFX_BENCH_TEST_STUCK_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--pool", "6",
           "--settle-seconds", "0.1", "--content-sweep", "0", "--side-leg-timeout", "6"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "stuck in its collective on rank(s) [1]" in d["config"]["error"]
    assert "exitcode  : 5" in r.stderr or "exitcode: 5" in r.stderr or "exit code 5" in d["config"]["error"]
