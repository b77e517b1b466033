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


def _fat_document():
    """a document as large as round 3's 20 KB line: prose notes, nested legs, per-rank lists"""
    prose = "x" * 1500
    roof = {"bound": "valu", "achieved": 5731.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.716390432, "frac_kernel": 0.7237,
            "achieved_kernel": 5790.1, "frac_valu": 0.98, "traffic": None, "traffic_committed": 1063442837.92, "traffic_source": prose,
            "traffic_over_algorithmic": 1.0016574, "kernel": "hsvfilter4_typed_kernel", "bytes_per_step": 1061683200, "avg_step_us": 183.3,
            "p50_step_us": 185.3, "note": prose, "ceiling_note": prose, "ceilings": {"a": {"GBs": 6000.0, "us_per_launch": 100.0}},
            "step_us": {"n": 200, "p10": 183.3, "p50": 185.3, "p90": 188.2, "mean": 185.5, "unit": "us"}}
    cpu = {"value": 3.21, "unit": "frames/s", "cores": 1, "kind": "port", "sample": prose,
           "all_cores": {"value": 400.2, "unit": "frames/s", "cores": 256, "nproc": 256, "sample": prose}}
    leg = {"metric": "colorlut_frames_per_sec", "value": 72107.9, "unit": "frames/s", "ms_per_step": 0.22, "data": prose,
           "config": {"workload": prose, "per_rank_units_per_sec": [72107.9]}, "roofline": dict(roof), "cpu_baseline": dict(cpu),
           "sub_extra": {"noise_fps": {"0": 80000, "3": 70000, "5": 60000, "8": 55000, "16": 30000}}}
    doc = {"metric": "hsvfilter_4k_rgba_frames_per_sec", "value": 86640.5123, "unit": "frames/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 0.1847, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": prose,
           "config": {"workload": "hsvfilter 3840x2160 RGBA in place, hue-shift=90 saturation-mul=1.25 saturation-off=-0.05 value-mul=0.9 value-off=0.02",
                      "launch_model": "1 launch x 16 frames (mvfx_hsvfilter_transform_frames_ip, blockIdx.z = stream)", "frames_per_step_per_gpu": 16,
                      "parallelism": "8 independent stream shards, no data-path collective", "rccl_ranks": 8,
                      "per_rank_frames_per_sec": [86640.5123] * 8, "other_launch_model": {"note": prose},
                      "element_path": {"threads16_fps": 80000.0, "threads16_frac": 0.66, "one_thread_fps": 81000.0, "one_thread_frac": 0.67},
                      "other_configs": {k: dict(leg) for k in ("hsv1080p", "colorlut_natural", "colorlut_random", "videofx", "videocompare_blockhash",
                                                                "videocompare_dssim", "hsvfilter_rgb", "hsvdetector_rgb")}},
           "roofline": roof, "cpu_baseline": cpu}
    return doc


def test_final_line_is_small_and_carries_roofline_and_cpu_baseline(tmp_path, capsys, monkeypatch):
    """the driver keeps a tail of stdout; a 20 KB final line arrived beheaded and nothing was parsed.  The LAST line must stay
    below 3000 bytes for any document, carry `roofline` and `cpu_baseline`, and every earlier line must be a small JSON line of its own."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "FULL_DOC", str(tmp_path / "bench_out" / "last_run.json"))
    doc = _fat_document()
    assert len(json.dumps(doc)) > 20000
    bench.emit(doc, list(doc["config"]["other_configs"].items()))
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 9
    last = json.loads(lines[-1])
    assert len(lines[-1]) < 3000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in last, k
    assert last["config"]["workload"].startswith("hsvfilter 3840x2160 RGBA") and "model" not in last["config"]
    r = last["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["traffic"] is None and r["traffic_committed"] > 0 and "frac_kernel" in r and "frac_valu" in r
    c = last["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["all_cores"]["cores"] == 256 and len(c["sample"]) <= 200
    for ln in lines[:-1]:
        assert len(ln) <= 1000
        s = json.loads(ln)
        assert s["sub"] and s["value"] > 0 and "frac_wall" in s and "cpu_value" in s
    # the whole document travels in the file
    full = json.loads((tmp_path / "bench_out" / "last_run.json").read_text())
    assert full["config"]["other_configs"]["colorlut_natural"]["data"] == doc["config"]["other_configs"]["colorlut_natural"]["data"]


def test_a_failed_side_leg_is_a_small_error_line(tmp_path, capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "FULL_DOC", str(tmp_path / "last_run.json"))
    doc = _fat_document()
    bench.emit(doc, [("videofx", {"error": "RuntimeError: " + "y" * 5000})])
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 2 and len(lines[0]) < 1000 and json.loads(lines[0])["sub"] == "videofx" and len(lines[1]) < 3000


def test_a_hung_side_leg_fails_the_run_and_ends_the_survivors():
    """a rank stuck in the collective is a failure.  Rank 0 prints its line and leaves non-zero; the parent relays the
    line, ends the hung rank (exactly the PID it started) and exits non-zero itself."""
    r, dt = _run("stuck-side-leg", 2, 2)
    assert r.returncode != 0 and "worker(s) failed" in r.stderr and "(0, 5)" in r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["config"]["error"] == "side leg stuck"
    assert dt < 60


def test_ranks_agree_on_who_is_stuck_without_a_collective():
    """bench.agree_on_stuck: flags over the rendezvous store; a rank that never reports counts as stuck"""
    sys.path.insert(0, ROOT)
    import threading
    import torch.distributed as dist
    import bench
    store = dist.HashStore()
    res = {}

    def rank(r, stuck):
        res[r] = bench.agree_on_stuck(argparse.Namespace(rank=r, world=3), stuck, 5.0, store=store)

    ths = [threading.Thread(target=rank, args=(r, r == 1)) for r in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert res == {0: [1], 1: [1], 2: [1]}
    store = dist.HashStore()
    res.clear()
    ths = [threading.Thread(target=rank, args=(r, False)) for r in (0, 2)]  # rank 1 never arrives
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert res == {0: [1], 2: [1]}
    store = dist.HashStore()
    assert bench.agree_on_stuck(argparse.Namespace(rank=0, world=1), False, 5.0, store=store) == []
    bench.mark_emitted(None, store=store)
    t0 = time.time()
    bench.wait_emitted(None, 5.0, store=store)
    assert time.time() - t0 < 2


def test_an_early_rank_waits_for_the_late_ranks_flags_until_the_common_deadline():
    """advisor r5: a rank whose side leg fails at once publishes up to --side-leg-timeout (180 s) before the ranks that sit out their watchdog.
    With the old 60 s-per-rank patience it counted them as never arrived and left; with ONE deadline of watchdog + slack it waits.  Simulated
    time: the late rank publishes at t = 70 s of a clock that runs 100 x real time; both ranks must return the same list, and the early
    rank must not have returned before the late one published."""
    sys.path.insert(0, ROOT)
    import threading
    import torch.distributed as dist
    import bench
    store = dist.HashStore()
    t0 = time.monotonic()
    clock = lambda: (time.monotonic() - t0) * 100.0   # simulated seconds
    deadline = 180.0 + bench.SIDE_LEG_FLAG_SLACK_S
    res, returned_at = {}, {}

    class FastStore:
        """store.wait() timeouts are real time: scale them like the clock"""
        def __init__(self, inner): self.inner = inner
        def set(self, k, v): self.inner.set(k, v)
        def get(self, k): return self.inner.get(k)
        def wait(self, keys, td): self.inner.wait(keys, td / 100.0)

    def rank(r, publish_at):
        while clock() < publish_at:
            time.sleep(0.005)
        res[r] = bench.agree_on_stuck(argparse.Namespace(rank=r, world=2), r == 1, 180.0, store=FastStore(store), deadline=deadline, clock=clock)
        returned_at[r] = clock()

    ths = [threading.Thread(target=rank, args=(0, 0.0)), threading.Thread(target=rank, args=(1, 70.0))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert res == {0: [1], 1: [1]}
    assert returned_at[0] >= 70.0 and returned_at[0] < deadline
    # a rank that never publishes: the others give up AT the deadline, not 60 s per rank later
    store2 = dist.HashStore()
    t0 = time.monotonic()
    got = bench.agree_on_stuck(argparse.Namespace(rank=0, world=3), False, 180.0, store=FastStore(store2), deadline=deadline, clock=clock)
    assert got == [1, 2] and clock() < deadline + 150.0   # (each wait has a floor of one simulated second x the real-time store)


def test_final_line_sheds_instead_of_asserting(tmp_path, capsys, monkeypatch):
    """advisor r4: a line that is still too long after the first shedding must come out shorter, not as an AssertionError"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "FULL_DOC", str(tmp_path / "last_run.json"))
    doc = _fat_document()
    doc["config"]["workload"] = "w" * 4000
    doc["config"]["rccl_ranks"] = 8
    bench.emit(doc, [])
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(last) <= 3000
    d = json.loads(last)
    assert d["metric"] == doc["metric"] and d["value"] > 0 and d["config"]["full_document"]
