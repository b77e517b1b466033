#!/usr/bin/env python3
"""bench.py -- hsvfilter on 3840x2160 RGBA frames, device-resident, on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # spawns 8 worker processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of `--batch` synthetic 4K RGBA frames (one frame from each of
`--batch` independent streams), in place.  Two launch models are measured in the same run:

  * "batch"   -- mvfx_hsvfilter_transform_frames_ip: the `--batch` frames in ONE launch (blockIdx.z = stream);
  * "streams" -- what the GStreamer element does: `--batch` host threads (one streaming thread per stream), each
                 with its own HIP stream, each calling the SINGLE-frame mvfx_hsvfilter_transform_frame_ip once per
                 buffer (hsvfilter/imp.rs:322-326), no synchronisation between launches (libmvfxbench.so).

`value` is the model named by `config.launch_model` (--launch-model, default "batch"); the other model's number
travels in `config.other_launch_model`.  Frames are independent, so ranks shard streams with no data-path
collective ("weak" scaling: every GPU gets its own `--batch` streams); the only collectives are the timing barrier,
the max-over-ranks reduction and the gather of the per-rank rates.

Before the W warmup steps the same step runs untimed for --settle-seconds (0.6 s): the MI355X clock governor starts
every process in a low-power state and needs ~0.2 s of sustained load to reach its steady clocks.  Settle and warm-up
run on scratch batches; the K timed steps start on frames NO kernel has touched (fresh uniform-random bytes), so the
`data` field is literally true for the first `resident_batches` steps (K beyond that re-filters filtered frames).

Inputs are resident in HBM before the timed region.  The frame pool is much larger than the 256 MiB Infinity Cache
and every step touches a different batch, so reads come from HBM.

The JSON line carries `roofline` (algorithmic bytes per launch / average launch duration from HIP events on the
launch stream = `frac` / `frac_kernel`; the same bytes over the wall clock `value` is made of = `frac_wall`; p10/p50/p90 of
the per-launch time; HBM traffic per launch from the committed rocprofv3 PMC passes; the RMW-probe ceiling measured in the
same run) and, at N=1, `cpu_baseline` (the oracle's loop on one host thread -- what the reference does on its one streaming
thread -- and on nproc threads, bounded sample).

At N=1 the same run then measures BASELINE configs 2-5 and the side legs as SUB-LINES: one JSON line each (<= 1 KB, `"sub":
"<name>"`), printed BEFORE the final line -- hsv1080p, hsvfilter_rgb, hsvdetector_rgb, colorlut_natural (with the noise sweep),
colorlut_random, videofx, videocompare_blockhash, videocompare_dssim, gst_element_pipeline -- each with its own roofline fraction,
per-step percentiles, committed PMC traffic and a bounded CPU-port baseline.  The final line (the LAST line of stdout, <= 3 000
bytes) is the contract line of the headline only; the whole document is also written to bench_out/last_run.json.  At N>1 the
band-sharded videocompare leg with its RCCL all-reduce is a sub-line too.  `--workload <name>` prints one leg as its own final
line (what tools/r5_traffic.sh profiles).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
FRAME_BYTES = W4K * H4K * 4
SETTINGS = (90.0, 1.25, -0.05, 0.9, 0.02)  # SURVEY.md 8d hsvfilter settings
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


# ------------------------------------------------------------------------------------------------ launching

def spawn_workers(args, argv, script=None, device_count=None):
    """`python bench.py --gpus N` without a launcher: start N worker processes (one per GPU) BEFORE anything in this
    process touches the GPU.  The parent never initialises HIP and never exec()s; it relays rank 0's JSON line and
    exits non-zero if any worker failed."""
    import socket
    if device_count is None:
        import torch  # device_count() does not initialise the GPU on this image (unlike is_available())
        device_count = torch.cuda.device_count()
    have = device_count
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "MVFX_BENCH_WORKER": "1"})
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # one worker failing (e.g. no such device) must not leave the others waiting in the rendezvous: poll, and end the
    # rest -- exactly the PIDs started here -- as soon as one has exited non-zero
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()))
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join()
    sys.stdout.write("".join(chunks))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: worker(s) failed (rank, exit code): {bad}\n")
        sys.exit(1)


FINAL_LINE_LIMIT = 3000   # bytes: the driver keeps only a tail of stdout, so the LAST line must be small (round 3's 20 KB line arrived beheaded)
SUB_LINE_LIMIT = 1000
FULL_DOC = os.path.join(ROOT, "bench_out", "last_run.json")


def _r(x, digits=6):
    """floats to `digits` significant digits: the compact lines carry numbers, not float64 noise"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _flush_c_stdio():
    """native libraries (RCCL's version banner) write to C stdio, which is block-buffered on a pipe and would otherwise be
    flushed at exit, after the line"""
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def compact_cpu(cpu):
    if not cpu:
        return None
    out = {"value": _r(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": cpu["sample"][:200]}
    if "all_cores" in cpu:
        out["all_cores"] = {"value": _r(cpu["all_cores"]["value"]), "cores": cpu["all_cores"]["cores"]}
    return out


def compact_roofline(r):
    """scalars only.  `frac` = `achieved` / `peak` with `achieved` the algorithmic bytes over the WALL clock `value` is made of;
    `frac_kernel` the same bytes over the average launch duration between two HIP events on the launch stream; `frac_valu` the
    committed SQ_INSTS_VALU of the step x the measured 1.07 ns issue interval over 1024 SIMDs x the measured step time (a lower bound)."""
    keys = ("bound", "bound_counters", "achieved", "peak", "unit", "frac", "frac_kernel", "achieved_kernel", "frac_valu", "traffic", "traffic_committed",
            "traffic_over_algorithmic", "kernel", "bytes_per_step", "avg_step_us", "p50_step_us", "frac_of_measured_ceiling", "ceiling_measured_GBs")
    return {k: _r(r[k]) for k in keys if k in r}


def compact_headline(out):
    """The driver's line: what the bench contract names and nothing else (the whole document is FULL_DOC)."""
    c = out.get("config", {})
    cfg = {k: c[k] for k in ("workload", "launch_model", "frame_content", "frames_per_step_per_gpu", "parallelism", "rccl_ranks",
                             "rendezvous_backend", "collective", "error") if c.get(k) is not None}
    for k in ("per_rank_frames_per_sec", "per_rank_units_per_sec"):
        if k in c:
            cfg[k] = [_r(v, 5) for v in c[k]]
    for k in ("element_path", "value_p50", "last_distance", "allreduce_us", "side_legs"):
        if c.get(k) is not None:
            cfg[k] = _r(c[k]) if not isinstance(c[k], dict) else c[k]
    sweep = c.get("other_frame_contents") or {}
    if sweep:  # the same timed leg on the other frame contents (SURVEY 8d ii: smpte bars are never the only input): fraction of 8 TB/s over the wall clock
        cfg["other_contents_frac"] = {k: _r(v.get("frac_wall"), 4) for k, v in sweep.items()}
    cfg["full_document"] = os.path.relpath(FULL_DOC, ROOT)
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype") if k in out}
    line["data"] = out.get("data", "synthetic")[:160]
    if "verified" in out:
        line["verified"] = out["verified"]
    line["config"] = cfg
    line["roofline"] = compact_roofline(out.get("roofline", {}))
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = compact_cpu(out["cpu_baseline"])
    if "wall_s" in out:
        line["wall_s"] = _r(out["wall_s"], 4)
    return line


def compact_sub(key, r):
    """one of configs 2-5 (or a side measurement) as its own small line, printed BEFORE the final line"""
    if "error" in r:
        return {"sub": key, "error": str(r["error"])[:300]}
    rf = r.get("roofline", {})
    out = {"sub": key, "metric": r.get("metric"), "value": _r(r.get("value")), "unit": r.get("unit"), "ms_per_step": _r(r.get("ms_per_step")),
           "bound": rf.get("bound"), "frac_wall": _r(rf.get("frac")), "frac_kernel": _r(rf.get("frac_kernel")), "frac_valu": _r(rf.get("frac_valu")),
           "traffic_ratio": _r(rf.get("traffic_over_algorithmic"), 4)}
    cpu = r.get("cpu_baseline")
    if cpu:
        out["cpu_value"] = _r(cpu["value"])
        out["cpu_all_cores"] = _r(cpu.get("all_cores", {}).get("value"))
        out["cpu_cores"] = cpu.get("all_cores", {}).get("cores")
    for k in ("element_model", "last_distance", "n_gpus", "allreduce_us", "value_p10", "value_p50", "value_p90"):
        v = r.get("config", {}).get(k, r.get(k))
        if v is not None and not isinstance(v, (list, dict)):
            out[k] = _r(v, 5)
    if "verified" in r:
        out["verified"] = r["verified"]
        if r.get("verified") is False:
            out["verified_what"] = str(r.get("verified_what"))[:200]
    out.update(r.get("sub_extra", {}))
    return {k: v for k, v in out.items() if v is not None}


def emit(doc, subs=(), full=False):
    """Side lines first (one small JSON line each), then the ONE final line, last on stdout; the whole document goes to
    bench_out/last_run.json.  full=True (--full 1, the profiling tools): the final line is the whole document."""
    _flush_c_stdio()
    try:
        os.makedirs(os.path.dirname(FULL_DOC), exist_ok=True)
        with open(FULL_DOC, "w") as f:
            json.dump(doc, f, indent=1, default=str)
    except OSError as e:  # a read-only tree must not cost the line
        sys.stderr.write(f"bench.py: could not write {FULL_DOC}: {e}\n")
    for key, r in subs:
        s = json.dumps(compact_sub(key, r))
        if len(s) > SUB_LINE_LIMIT:
            s = json.dumps({"sub": key, "error": f"sub-line of {len(s)} bytes dropped (limit {SUB_LINE_LIMIT}); see the full document"})
        print(s, flush=True)
    if full:
        print(json.dumps(doc, default=str), flush=True)
        return
    line = compact_headline(doc)
    s = json.dumps(line)
    if len(s) > FINAL_LINE_LIMIT:  # never hand the driver a line it will cut: shed the optional parts
        for k in ("side_legs", "element_path", "per_rank_frames_per_sec", "per_rank_units_per_sec"):
            line["config"].pop(k, None)
        line["data"] = line["data"][:60]
        s = json.dumps(line)
    if len(s) > FINAL_LINE_LIMIT:  # still too long (many ranks, long notes): keep shedding -- a valid short line beats no line at all
        for shed in (lambda: line.get("cpu_baseline", {}).pop("sample", None),
                     lambda: line.__setitem__("roofline", {k: v for k, v in line.get("roofline", {}).items()
                                                           if k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}),
                     lambda: line.__setitem__("config", {k: v for k, v in line["config"].items() if k in ("workload", "full_document", "error")}),
                     lambda: line["config"].__setitem__("workload", str(line["config"].get("workload", ""))[:200])):
            shed()
            s = json.dumps(line)
            if len(s) <= FINAL_LINE_LIMIT:
                break
    if len(s) > FINAL_LINE_LIMIT:
        s = json.dumps({k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                 "scaling", "vs_baseline", "dtype")} | {"config": {"full_document": os.path.relpath(FULL_DOC, ROOT)}})
    print(s, flush=True)


class Worker:
    """One rank: device, torch.distributed over RCCL when WORLD_SIZE > 1, the C ABI."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import _pkg
        self.torch, self.dist = torch, dist
        self.vfx = _pkg.vfx
        self.lib = self.vfx.lib()  # raises if libmi355vfx.so is missing: no fallback
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and self.rank == 0:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; using WORLD_SIZE\n")
        if not torch.cuda.is_available():
            sys.stderr.write("bench.py: no GPU visible; the HIP path has no CPU fallback\n")
            sys.exit(3)
        # MVFX_BENCH_TEST_SHARED_GPU=1 (tests/test_distributed_gpu.py only): every rank on GPU local_rank % device_count and the
        # rendezvous over gloo, so that the N > 1 control flow of this file runs on the one-GPU box (RCCL refuses two ranks on one GPU;
        # the line then says backend "gloo" and is not a measurement)
        shared = os.environ.get("MVFX_BENCH_TEST_SHARED_GPU") == "1"
        self.backend = "gloo" if shared else "nccl"
        self.device_index = self.local_rank % max(torch.cuda.device_count(), 1) if shared else self.local_rank
        torch.cuda.set_device(self.device_index)
        self.dev = torch.device("cuda", self.device_index)
        self.coll_dev = torch.device("cpu") if shared else self.dev  # where the small tensors of the timing collectives live
        self.rccl_ranks = 1
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if shared:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=self.dev)  # "nccl" IS RCCL on ROCm
            one = torch.ones(1, dtype=torch.int32, device=self.coll_dev)
            dist.all_reduce(one)  # an actual collective: how many ranks RCCL sees
            self.rccl_ranks = int(one[0])
        self.vfx.check(self.lib.mvfx_set_device(self.device_index))
        self.stream = torch.cuda.current_stream(self.dev)
        self.sptr = ctypes.c_void_p(self.stream.cuda_stream)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, *vals):
        if self.world == 1:
            return vals
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    def gather(self, val):
        if self.world == 1:
            return [val]
        t = self.torch.tensor([val], dtype=self.torch.float64, device=self.coll_dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(x[0]) for x in out]

    def timed(self, step, steps, first_index=0, events=False, join=None):
        """barrier + synchronize on both sides; returns (wall seconds, average ms between the two HIP events)."""
        torch = self.torch
        self.sync()
        self.barrier()
        self.sync()
        ev0 = ev1 = None
        if events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if events:
            ev0.record(self.stream)
        for i in range(steps):
            step(first_index + i)
        if events:
            if join is not None:
                join()  # a leg that uses a second stream: the launch stream waits for it, so the closing event covers both
            ev1.record(self.stream)
            # poll the closing event before the synchronize below: a sleeping hipDeviceSynchronize wakes up 40-100 us late, which is
            # 1-3 % of a 3.7 ms timed region (20 launches); the region stays bracketed by barrier + synchronize on both sides
            while not ev1.query():
                pass
        self.sync()
        self.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        return elapsed, (ev0.elapsed_time(ev1) / max(steps, 1) if events else None)

    def event_times(self, step, steps, first_index=0, join=None):
        """`steps` steps with a HIP event on the launch stream between every two: per-step microseconds (launch + whatever
        the step leaves between two launches), for the p10 / p50 / p90 of the line."""
        torch = self.torch
        self.sync()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record(self.stream)
        for i in range(steps):
            step(first_index + i)
            if join is not None:
                join()
            evs[i + 1].record(self.stream)
        self.sync()
        return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(steps)]

    def finish(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def settle(step, seconds, sync, fixed_steps=None):
    """Untimed run of `step` so the clock governor leaves its low-power state (see --settle-seconds).
    fixed_steps: for steps that contain a collective every rank must run the same number of them."""
    n = 0
    if seconds > 0 and fixed_steps is not None:
        for n in range(fixed_steps):
            step(n)
        sync()
        return fixed_steps
    if seconds > 0:
        t = time.perf_counter()
        while time.perf_counter() - t < seconds:
            for _ in range(50):
                step(n)
                n += 1
            sync()
    return n


def percentiles(us):
    """p10 / p50 / p90 / mean of a list of per-step microseconds (median and p10/p90 over >= 200 iterations)."""
    v = sorted(us)
    n = len(v)
    if n == 0:
        return None
    pick = lambda q: v[min(n - 1, max(0, int(round(q * (n - 1)))))]
    return {"n": n, "p10": pick(0.10), "p50": pick(0.50), "p90": pick(0.90), "mean": sum(v) / n, "unit": "us"}


def bench_harness():
    """libmvfxbench.so (measurement only: host threads of the per-stream launch model, the RMW ceiling probe)."""
    hb = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))  # raises when build() has not run
    hb.mvfxbench_rmw_ceiling.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return hb


def measured_rmw_ceiling(w, base_ptr, bytes_per_launch, regions):
    """The practical HBM ceiling SURVEY 8d asks to quote beside the 8 TB/s spec, measured in THIS run on the same resident
    pool: the memory shape of the headline kernel with trivial arithmetic (gst-plugin-rs_amd/bench/probe_rmw.hip: one
    16-byte non-temporal load + store per lane, in place; the out-of-place twin for the filters that write another frame)."""
    hb = bench_harness()
    out = {}
    for mode, key in ((0, "in_place_nt"), (1, "in_place_cached"), (2, "out_of_place_nt"), (3, "in_place_write_through")):
        gbs, us = ctypes.c_double(), ctypes.c_double()
        n = 4 * regions  # every region XOR-ed an even number of times: the pool is left as it was
        rc = hb.mvfxbench_rmw_ceiling(ctypes.c_void_p(base_ptr), bytes_per_launch, regions, n, n, mode, w.sptr,
                                      ctypes.byref(gbs), ctypes.byref(us))
        if rc != 0:
            raise RuntimeError(f"mvfxbench_rmw_ceiling status {rc}")
        out[key] = {"GBs": gbs.value, "us_per_launch": us.value}
    return out


# ------------------------------------------------------------------------------------------------ frame contents

FRAME_CONTENTS = ("videotestsrc", "natural", "random")
CONTENT_TEXT = {
    "videotestsrc": "videotestsrc pattern=smpte RGBA frames (colour bars + LCG snow, byte-identical to GStreamer's generator: "
                    "tests/test_videotestsrc_frames_cpu.py; consecutive frames of the stream, i.e. distinct snow in each)",
    "natural": "smooth colour gradients (another phase per frame) + uniform noise of +-3 codes (natural-like)",
    "random": "uniform-random u8 RGBA (torch.randint)",
}


def natural_frame(torch, dev, gen, k, W, H, noise=3):
    """One natural-like RGBA frame as a flat u8 tensor: smooth 2-D colour gradients (phase k) + uniform noise of +-`noise` codes."""
    x = torch.linspace(0, 1, W, device=dev).view(1, W)
    y = torch.linspace(0, 1, H, device=dev).view(H, 1)
    ph = 0.37 * k
    img = torch.stack([(0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)).expand(H, W),
                       (0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)).expand(H, W),
                       (0.5 + 0.45 * torch.cos(4.0 * x * y + ph)).expand(H, W),
                       torch.ones((H, W), device=dev)], dim=-1) * 255.0
    if noise > 0:
        nz = torch.randint(-noise, noise + 1, img.shape, device=dev, generator=gen).float()
        nz[..., 3] = 0
        img = img + nz
    return img.clamp(0, 255).to(torch.uint8).view(-1)


def fill_frames(torch, dev, gen, flat, kind, W, H, first_frame=0, noise=3):
    """Fill flat[n, W*H*4] (device, u8) with `kind` frames; frame j is frame first_frame + j of its stream."""
    n = flat.shape[0]
    if kind == "random":
        flat.random_(0, 256, generator=gen)
    elif kind == "natural":
        for j in range(n):
            flat[j] = natural_frame(torch, dev, gen, first_frame + j, W, H, noise)
    elif kind == "videotestsrc":
        import numpy as np
        from tests import frames as _frames
        base, _ = _frames.videotestsrc_smpte(W, H, 1)
        flat[:] = torch.from_numpy(base.reshape(-1)).to(dev).unsqueeze(0)
        x0, y0 = _frames.vts_snow_geometry(W, H)
        per_frame = (W - x0) * (H - y0)
        a_np, c_np = _frames.vts_lcg_affine(per_frame)
        a_full, c_full = int(a_np[-1]), int(c_np[-1])           # the per_frame-step map: state at the start of the next frame
        a = torch.from_numpy(a_np.astype(np.int64)).to(dev)
        c = torch.from_numpy(c_np.astype(np.int64)).to(dev)
        state = 0
        for _ in range(first_frame):
            state = (a_full * state + c_full) & 0xFFFFFFFF
        for j in range(n):
            st = (a * state + c) & 0xFFFFFFFF                   # int64 products wrap mod 2^64: the low 32 bits are exact
            grey = ((st >> 16) & 0xFF).to(torch.uint8).view(H - y0, W - x0, 1)
            flat[j].view(H, W, 4)[y0:, x0:, :3] = grey
            state = (a_full * state + c_full) & 0xFFFFFFFF
    else:
        raise ValueError(kind)
    if dev.type == "cuda":
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ CPU baselines
# The reference is Rust and cannot be built here or on the GPU box (no rustc): the timed CPU path is oracle/*.c, the
# statement-by-statement port of the reference loops (gcc -O3 -ffp-contract=off like its profile.release), "kind": "port".
# Only this leg of bench.py touches oracle/.  One thread = what the reference does (its transform_frame runs on ONE streaming
# thread per element); nproc threads = that many independent streams.  Bounded: ~`seconds` per leg.

def _nproc():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_rate(make_unit, seconds, n_threads):
    """make_unit(t) -> (prepare, unit): prepare() is untimed (fresh input for in-place loops), unit() is one unit of work,
    timed; ctypes releases the GIL for the oracle calls.  Returns (units/s summed over the threads, units, longest span)."""
    import threading
    counts = [0] * n_threads
    spans = [0.0] * n_threads
    go = threading.Event()

    def body(t):
        prepare, unit = make_unit(t)
        prepare()
        unit()  # warm: page faults, code
        go.wait()
        n, dt = 0, 0.0
        while dt < seconds:
            prepare()
            t1 = time.perf_counter()
            unit()
            dt += time.perf_counter() - t1
            n += 1
        counts[t], spans[t] = n, dt

    ths = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    go.set()
    for th in ths:
        th.join()
    return sum(c / s for c, s in zip(counts, spans) if s > 0), sum(counts), max(spans)


NO_VERIFY = [False]       # --no-verify
# MVFX_BENCH_SIDE_NT=0 (A/B runs against older libraries): the hsv side legs without MVFX_OPT_NONTEMPORAL, as rounds 1-5 ran them
SIDE_NT = [int(os.environ.get("MVFX_BENCH_SIDE_NT", "1"))]
ALL_CORES_SECONDS = [None]  # budget of the nproc-thread leg (None: same as the 1-thread leg; 0: skip it) -- set from the command line


def cpu_baseline_of(make_unit, seconds, unit, what, scale=1.0, scale_note=None):
    nproc = _nproc()
    one, one_n, one_dt = cpu_rate(make_unit, seconds, 1)
    out = {"value": one * scale, "unit": unit, "cores": 1, "kind": "port",
           "sample": f"{one_n} x {what}, 1 thread, {one_dt:.1f} s" + (f"; {scale_note}" if scale_note else "")}
    all_s = seconds if ALL_CORES_SECONDS[0] is None else ALL_CORES_SECONDS[0]
    if all_s > 0:
        allr, all_n, all_dt = cpu_rate(make_unit, all_s, nproc) if nproc > 1 else (one, one_n, one_dt)
        out["all_cores"] = {"value": allr * scale, "unit": unit, "cores": nproc, "nproc": nproc,
                            "sample": f"{all_n} x the same unit on {nproc} threads (independent streams), {all_dt:.1f} s per thread"}
    return out


def cpu_baseline_hsvfilter(seconds, content="videotestsrc"):
    """hsvfilter/imp.rs:76-120 on 3840x2160 RGBA: 4 distinct frames of the same content as the GPU legs rotate so the 33 MB
    input is not cache resident; every thread filters a fresh copy (in-place loop), the copy is not timed."""
    import numpy as np
    from tests import frames
    from tests import oracle_binding as orc
    distinct = 4
    if content == "videotestsrc":
        vts, _ = frames.videotestsrc_smpte(W4K, H4K, distinct)
        src = [vts[k] for k in range(distinct)]
        what = "3840x2160 RGBA frame (videotestsrc pattern=smpte, 4 consecutive frames rotating)"
    else:
        src = [frames.random_frame(0x5EED0001 + k, W4K, H4K) for k in range(distinct)]
        what = "3840x2160 RGBA frame (uniform random, 4 distinct frames rotating, seeds 0x5EED0001..4)"

    def make_unit(t):
        work = src[0].copy()
        k = [t]

        def prepare():
            np.copyto(work, src[k[0] % distinct])
            k[0] += 1
        return prepare, lambda: orc.hsvfilter(work, W4K, W4K * 4, "RGBA", SETTINGS)
    return cpu_baseline_of(make_unit, seconds, "frames/s", what + ", oracle/hsv_oracle.c orc_hsvfilter_transform_frame_ip, gcc -O3 -ffp-contract=off")


def cpu_baseline_hsv1080p(seconds, host_frames):
    """config 2: hsvfilter in place (hsvfilter/imp.rs:76-120) then hsvdetector RGBx -> RGBA (hsvdetector/imp.rs:100-160)
    on one 1920x1080 frame = one unit."""
    import numpy as np
    from tests import oracle_binding as orc
    W, H = 1920, 1080

    def make_unit(t):
        work = host_frames[0].copy()
        out = np.empty_like(work)
        k = [t]

        def prepare():
            np.copyto(work, host_frames[k[0] % len(host_frames)])
            k[0] += 1

        def unit():
            orc.hsvfilter(work, W, W * 4, "RGBx", SETTINGS)
            orc.hsvdetector(work, W * 4, "RGBx", out, W * 4, "RGBA", W, DETECT_SETTINGS)
        return prepare, unit
    return cpu_baseline_of(make_unit, seconds, "frames/s", "1920x1080 RGBx frame through orc_hsvfilter_transform_frame_ip + "
                           "orc_hsvdetector_transform_frame (oracle/hsv_oracle.c)")


def cpu_baseline_colorlut(seconds, host_frames, cube_text, content):
    """config 3: transform_rgba_3d (colorlut/imp.rs:267-294, 431-535) on one 3840x2160 RGBA frame = one unit."""
    import numpy as np
    from tests import oracle_binding as orc
    lut = orc.CubeLut(cube_text)

    def make_unit(t):
        out = np.empty_like(host_frames[0])
        k = [t]

        def unit():
            src = host_frames[k[0] % len(host_frames)]
            k[0] += 1
            lut.apply(src, W4K * 4, out, W4K * 4, W4K, H4K, "RGBA")
        return (lambda: None), unit
    return cpu_baseline_of(make_unit, seconds, "frames/s", f"3840x2160 RGBA frame ({content}) through orc_colorlut_transform_frame, "
                           "33^3 cube (oracle/colorlut_oracle.c)")


def cpu_baseline_videofx(seconds, host_rgba):
    """config 4 per frame in the reference: roundedcorners appends the shared alpha GstMemory (border/imp.rs:527-557, O(1), no
    pixel work) and colordetect runs color_thief::get_palette on the whole plane (colordetect/imp.rs:57-86) = one unit."""
    from tests import oracle_binding as orc

    def make_unit(t):
        k = [t]

        def unit():
            f = host_rgba[k[0] % len(host_rgba)]
            k[0] += 1
            orc.colordetect_palette(f, "RGBA", 10, 2)
        return (lambda: None), unit
    return cpu_baseline_of(make_unit, seconds, "frames/s", "3840x2160 RGBA frame through orc_colordetect_palette quality=10 max-colors=2 "
                           "(histogram + median cut, oracle/videofx_oracle.c); the reference's roundedcorners does no per-frame pixel work")


def cpu_baseline_blockhash(seconds, host_pair):
    """config 5, hash-algo=blockhash: HasherEngine::hash_image x2 + compare (hashed_image.rs:24-79) on one 7680x4320 pair."""
    from tests import oracle_binding as orc
    W, H = 7680, 4320

    def make_unit(t):
        def unit():
            _, ha = orc.blockhash(host_pair[0], W, H, W * 4, "RGBA")
            _, hb = orc.blockhash(host_pair[1], W, H, W * 4, "RGBA")
            orc.hamming(ha, hb)
        return (lambda: None), unit
    return cpu_baseline_of(make_unit, seconds, "pairs/s", "7680x4320 RGBA pair through orc_blockhash x2 + orc_hamming64 (oracle/videofx_oracle.c)")


def cpu_baseline_dssim(seconds, host_pair_crop, cw, ch):
    """config 5, hash-algo=dssim: one 8K pair is ~18 s of one host thread, so the bounded sample is a cw x ch crop of the same
    pair; the cost of every pyramid level is linear in the pixel count, so pairs/s is scaled by the pixel ratio."""
    from tests import oracle_binding as orc
    scale = (cw * ch) / (7680.0 * 4320.0)

    def make_unit(t):
        def unit():
            orc.ssim_distance(host_pair_crop[0], host_pair_crop[1], cw, ch, cw * 4, cw * 4, "RGBA")
        return (lambda: None), unit
    return cpu_baseline_of(make_unit, seconds, "pairs/s", f"{cw}x{ch} RGBA crop pair through orc_ssim_distance (oracle/ssim_oracle.c, f64)",
                           scale=scale, scale_note=f"scaled to 7680x4320 pairs by the pixel ratio {scale:.5f}")


# ------------------------------------------------------------------------------------------------ HBM traffic (committed PMC passes)

def committed_traffic(key, units_per_step):
    """HBM bytes per step from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot be collected live inside
    a timed run): profiles/traffic.json, written by tools/r3_traffic.py from `rocprofv3 --pmc` runs of `bench.py --workload
    <key>`.  Returns (bytes per step scaled to this run's units per step, source text) or (None, reason)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[key]
        return t["hbm_bytes_per_step"] * units_per_step / t["units_per_step"], t["source"]
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    if key == "hsvfilter":
        try:
            with open(os.path.join(ROOT, "profiles", "hsvfilter_traffic.json")) as f:
                t = json.load(f)
            return t["hbm_bytes_per_launch"] * units_per_step / t["frames_per_launch"], "committed rocprofv3 pass: " + t["source"]
        except (OSError, KeyError, ValueError):
            pass
    return None, "no committed rocprofv3 PMC pass for this workload"


VALU_SIMDS = 256 * 4       # 256 CUs x 4 SIMDs
# one wave64 VALU instruction per 1.07 ns per SIMD is the fastest a SIMD issues them (v_fma/add/mul_f32, v_add_u32, v_and ...: 2 cycles
# at the ~2.1 GHz the chip holds under an all-VALU load); min/max/cvt/perm/cmp/SDWA and any operation with an SGPR source take 1.75 ns
# on a second pipe that overlaps with the first, v_rcp_f32 3.46 ns overlapping with nothing: tools/probes/valu_issue.hip,
# profiles/r4/valu_issue_probe.txt.  (SQ_ACTIVE_INST_VALU / the VALUBusy metric count a flat 4 cycles per instruction and pass 100 %.)
VALU_ISSUE_NS = 1.07
# which ceiling binds, from the committed rocprofv3 counters (VALUBusy / MemUnitBusy / SQ_INSTS_VALU passes under profiles/); the
# committed profiles/traffic.json overrides this table per workload when it carries a "bound" field
BOUND_FROM_COUNTERS = {"hsvfilter": "valu", "hsv1080p": "valu", "hsvfilter_rgb": "valu", "hsvdetector_rgb": "valu",
                       "colorlut_natural": "valu", "colorlut_random": "hbm", "colorlut_smpte": "valu",
                       "videofx": "hbm", "videocompare_blockhash": "hbm", "videocompare_dssim": "valu"}


def committed_counters(key, units_per_step):
    """{"bound", "valu_insts" (SQ_INSTS_VALU wave-instructions per step, or None)} from the committed PMC passes"""
    bound, valu = BOUND_FROM_COUNTERS.get(key, "hbm"), None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[key]
        bound = t.get("bound") or bound
        if t.get("valu_insts_per_step"):
            valu = t["valu_insts_per_step"] * units_per_step / t["units_per_step"]
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    return {"bound": bound, "valu_insts": valu}


def frac_valu(valu_insts, step_seconds):
    """fraction of the VALU issue ceiling: SQ_INSTS_VALU x 1.07 ns / (1024 SIMDs x step time).  Every instruction is priced as the
    fastest class, so this is a lower bound of the time the step's VALU work needs (tools/valu_cost.py prices a kernel's static mix)."""
    if not valu_insts or not step_seconds:
        return None
    return valu_insts * VALU_ISSUE_NS * 1e-9 / (VALU_SIMDS * step_seconds)


def roofline_of(key, units_per_step, bytes_per_step, wall_s_per_step, event_s_per_step, kernel, pct=None):
    """The `roofline` object of a line: `achieved` / `frac` over the wall clock `value` is made of, `*_kernel` over the HIP-event
    average on the launch stream, `bound` + `frac_valu` and the HBM `traffic_committed` from the committed PMC passes."""
    traffic, traffic_source = committed_traffic(key, units_per_step)
    ctr = committed_counters(key, units_per_step)
    achieved_wall = bytes_per_step / wall_s_per_step / 1e9
    achieved_kernel = bytes_per_step / event_s_per_step / 1e9 if event_s_per_step else None
    out = {"bound": ctr["bound"], "achieved": achieved_wall, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_wall / HBM_PEAK_GBS,
           "achieved_kernel": achieved_kernel, "frac_kernel": achieved_kernel / HBM_PEAK_GBS if achieved_kernel else None,
           "frac_valu": frac_valu(ctr["valu_insts"], event_s_per_step or wall_s_per_step),
           "traffic": None, "traffic_committed": traffic, "traffic_source": traffic_source,
           "traffic_over_algorithmic": (traffic / bytes_per_step) if traffic else None,
           "kernel": kernel, "bytes_per_step": bytes_per_step,
           "avg_step_us": event_s_per_step * 1e6 if event_s_per_step else None, "step_us": pct,
           "p50_step_us": pct["p50"] if pct else None,
           "note": "traffic: null because FETCH_SIZE / WRITE_SIZE cannot be collected inside a timed run; traffic_committed = the committed "
                   "rocprofv3 --pmc passes of this workload (profiles/traffic.json), scaled to this run's units per step"}
    return out


# ------------------------------------------------------------------------------------------------ the legs (configs 2-5)

DETECT_SETTINGS = (120.0, 40.0, 0.6, 0.4, 0.6, 0.4)  # SURVEY.md 8d hsvdetector settings


class Leg:
    """One workload on this rank: `step(i)` enqueues one pass of the path over one batch on w.stream."""

    def __init__(self, key, metric, unit, units_per_step, bytes_per_step, dtype, data, workload, step, kernels, cpu=None,
                 fixed_settle=None, extra=None, note=None, scaling="weak", total_units=None):
        self.key, self.metric, self.unit = key, metric, unit
        self.units_per_step, self.bytes_per_step = units_per_step, bytes_per_step
        self.dtype, self.data, self.workload, self.step, self.kernels = dtype, data, workload, step, kernels
        self.cpu, self.fixed_settle, self.extra, self.note, self.scaling = cpu, fixed_settle, extra or {}, note, scaling


def measure_leg(w, leg, steps, warmup, settle_seconds, pct_steps, cpu_seconds):
    """settle -> W warm-up -> K steps between barriers (wall clock + one pair of HIP events on the launch stream) -> pct_steps
    steps with a HIP event between every two (p10 / p50 / p90 of the per-step time) -> the CPU port on the host cores."""
    executed = [0]
    t_measure = time.perf_counter()
    verified = run_verify(w, leg)
    t_verified = time.perf_counter()

    def counted(i):
        executed[0] += 1
        return leg.step(i)

    settle(counted, settle_seconds, w.sync, fixed_steps=leg.fixed_settle)
    for i in range(warmup):
        counted(i)
    join = getattr(leg, "join", None)
    secs, ev_ms = w.timed(counted, steps, first_index=warmup, events=True, join=join)
    secs, ev_ms = w.max_over_ranks(secs, ev_ms)
    per_rank = w.gather(steps * leg.units_per_step / secs)
    pct = percentiles(w.event_times(counted, pct_steps, first_index=warmup + steps, join=join)) if pct_steps > 0 else None
    world = w.world
    per_gpu_bytes = leg.bytes_per_step  # per rank and step
    value = steps * leg.units_per_step * (world if leg.scaling == "weak" else 1) / secs
    roof = roofline_of(leg.key, leg.units_per_step, per_gpu_bytes, secs / steps, ev_ms * 1e-3, ", ".join(leg.kernels), pct)
    roof["avg_step_ms"] = ev_ms
    if leg.note:
        roof["leg_note"] = leg.note
    out = {
        "metric": leg.metric, "value": value, "unit": leg.unit, "steps": steps, "warmup": warmup,
        "ms_per_step": secs / steps * 1e3, "scaling": leg.scaling, "dtype": leg.dtype, "data": leg.data,
        "config": {"workload": leg.workload, "units_per_step_per_gpu": leg.units_per_step, "per_rank_units_per_sec": per_rank,
                   "steps_executed": executed[0], **leg.extra},
        "roofline": roof,
    }
    if pct:
        out["config"]["value_p50"] = leg.units_per_step * (world if leg.scaling == "weak" else 1) / (pct["p50"] * 1e-6)
        for q in ("p10", "p90"):  # the spread of the event-timed steps, as rates: p10 of the step time is the FAST end
            out["config"]["value_" + q] = leg.units_per_step * (world if leg.scaling == "weak" else 1) / (pct[q] * 1e-6)
    out.update(verified)
    t_gpu = time.perf_counter()
    if leg.cpu is not None and w.rank == 0 and world == 1 and cpu_seconds > 0:
        out["cpu_baseline"] = leg.cpu(cpu_seconds)
    out["measure_seconds"] = {"gpu": t_gpu - t_verified, "verify": t_verified - t_measure, "cpu": time.perf_counter() - t_gpu}
    return out


def run_verify(w, leg):
    """{"verified": true | false | null, "verified_what": ...}: ONE step of the leg -- the same C entry, the same batch shape, the leg's own
    frames -- compared with the CPU oracle (tests/oracle_binding) before anything is timed.  The oracle is the checker here, never the thing
    measured.  null: the leg has no verifier on this rank (N > 1 ranks other than 0)."""
    fn = getattr(leg, "verify", None)
    if fn is None or w.rank != 0 or NO_VERIFY[0]:
        return {"verified": None}
    try:
        what = fn()
        return {"verified": True, "verified_what": what}
    except Exception as e:  # noqa: BLE001  a mismatch is reported in the line, the measurement still runs (and is worthless)
        return {"verified": False, "verified_what": f"{type(e).__name__}: {e}"[:300]}


def _same(got, want, what):
    import numpy as np
    if not np.array_equal(got, want):
        raise AssertionError(f"{what}: {int(np.count_nonzero(np.asarray(got) != np.asarray(want)))} of {np.asarray(want).size} values differ from the oracle")


def make_leg_hsv1080p(w, args):
    """config 2: args.batch independent 1080p streams per step: one frame of each through hsvfilter (in place, RGBx) then
    hsvdetector (RGBx -> RGBA): two launches per step (a single 1080p frame is ~3 + ~6 us of GPU work: launch-bound alone)."""
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    W, H, nb = 1920, 1080, args.batch
    pool = max(2, 96 // nb)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0200 + w.rank)
    src = torch.empty((pool * nb, W * H * 4), dtype=torch.uint8, device=dev)
    fill_frames(torch, dev, gen, src, "videotestsrc", W, H, first_frame=w.rank * pool * nb)
    dst = torch.empty((pool * nb, W * H * 4), dtype=torch.uint8, device=dev)
    fs = vfx.HsvFilterSettings(*SETTINGS)
    ds = vfx.HsvDetectorSettings(*DETECT_SETTINGS)
    fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBx") for i in range(nb)]) for b in range(pool)]
    fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
    host = [src[k].cpu().numpy().reshape(H, W * 4).copy() for k in range(4)] if w.rank == 0 and w.world == 1 else None

    nt = SIDE_NT[0] and vfx.OPT_NONTEMPORAL

    def step(i):
        # the filter's output is the detector's input (ordinary cached stores: the detector finds the frame in the caches); the detector's output
        # is read by nobody on the device: MVFX_OPT_NONTEMPORAL = write-through stores (csrc/device_store.hpp)
        k = i % pool
        vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(fi[k], nb, ctypes.byref(fs), sptr))
        lib.mvfx_thread_set_options(nt)
        vfx.check(lib.mvfx_hsvdetector_transform_frames(fi[k], fo[k], nb, ctypes.byref(ds), sptr))
        lib.mvfx_thread_set_options(0)
    leg = Leg("hsv1080p", "hsv1080p_frames_per_sec", "frames/s", nb, nb * 4 * W * H * 4, "f32",
              "synthetic videotestsrc pattern=smpte 1920x1080 RGBx frames, device-resident",
              f"hsvfilter (RGBx, in place) + hsvdetector RGBx->RGBA, {nb} streams of 1920x1080 per launch; 8 + 8 algorithmic B/px",
              step, ["hsvfilter4_typed_kernel", "hsvdetector_typed_kernel"],
              cpu=(lambda s: cpu_baseline_hsv1080p(s, host)) if host else None)
    leg.keep = (src, dst, fi, fo)

    def verify():
        import numpy as np
        from tests import oracle_binding as orc
        picks = (0, nb - 1)
        before = {i: src[i].cpu().numpy().reshape(H, W * 4).copy() for i in picks}
        step(0)
        w.sync()
        for i in picks:
            mid = before[i]
            orc.hsvfilter(mid, W, W * 4, "RGBx", SETTINGS)
            _same(src[i].cpu().numpy().reshape(H, W * 4), mid, f"hsvfilter frame {i}")
            want = np.empty_like(mid)
            orc.hsvdetector(mid, W * 4, "RGBx", want, W * 4, "RGBA", W, DETECT_SETTINGS)
            _same(dst[i].cpu().numpy().reshape(H, W * 4), want, f"hsvdetector frame {i}")
        return f"frames 0 and {nb - 1} of the first {nb}-frame step, every byte of both outputs, against oracle/hsv_oracle.c"
    leg.verify = verify
    return leg


def make_leg_colorlut(w, args, content):
    """config 3: args.batch streams graded with the same 33^3 LUT, one 4K frame of each per launch."""
    from tests import cubes
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    W, H, nb = W4K, H4K, args.batch
    pool = max(2, 32 // nb)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0300 + w.rank)
    cube_text = cubes.analytic_3d(33)
    lut = vfx.CubeLut(cube_text)
    src = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
    if content == "random":
        src.random_(0, 256, generator=gen)
        data = "synthetic uniform-random u8 RGBA (every pixel another LUT cell: the gather worst case), device-resident"
    elif content == "smpte":
        fill_frames(torch, dev, gen, src, "videotestsrc", W, H, first_frame=w.rank * pool * nb)
        data = "synthetic " + CONTENT_TEXT["videotestsrc"] + ", device-resident"
    else:
        fill_frames(torch, dev, gen, src, "natural", W, H, first_frame=w.rank * pool * nb, noise=args.noise)
        data = "synthetic " + CONTENT_TEXT["natural"].replace("+-3", f"+-{args.noise}") + ", device-resident"
    dst = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
    fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
    fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
    host = [src[k].cpu().numpy().reshape(H, W * 4).copy() for k in range(3)] if w.rank == 0 and w.world == 1 else None

    def step(i):
        k = i % pool
        vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[k], fo[k], nb, sptr))

    def streams_leg():
        """the element's launch model: --stream-threads host threads x own HIP stream x single-frame mvfx_colorlut_transform_frame"""
        hb = bench_harness()
        nthr = args.stream_threads
        fpt = max(1, (pool * nb) // nthr)
        fin = (vfx.Frame * (nthr * fpt))(*[vfx.make_frame(src[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nthr * fpt)])
        fout = (vfx.Frame * (nthr * fpt))(*[vfx.make_frame(dst[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nthr * fpt)])
        launches, reps = 100, 5
        secs, per = (ctypes.c_double * reps)(), (ctypes.c_double * nthr)()
        w.sync()
        w.barrier()
        rc = hb.mvfxbench_colorlut_streams(w.device_index, nthr, 200, launches, reps, lut.h, fin, fout, fpt, 0, secs, per)
        if rc != 0:
            raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
        w.barrier()
        (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
        fps = nthr * launches * w.world / med
        return {"launch_model": f"{nthr} threads x 1 frame (own HIP stream each, single-frame mvfx_colorlut_transform_frame)",
                "value": fps, "unit": "frames/s", "launches_per_thread": launches, "statistic": "median of 5 repetitions",
                "frac_wall": fps / w.world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
    def lane_leg():
        """the element's contract through the direct-dispatch lane: ONE thread, one mvfx_colorlut_transform_frame per frame pair, a fence per frame; the
        frames are independent, so the packets go out without the barrier bit (MVFX_OPT_DIRECT_UNORDERED: what the element does when neither buffer's
        acquire rested on queue order)"""
        hb = bench_harness()
        nfr = min(pool * nb, 12)
        fin = (vfx.Frame * nfr)(*[vfx.make_frame(src[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nfr)])
        fout = (vfx.Frame * nfr)(*[vfx.make_frame(dst[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nfr)])
        launches, reps = 1500, 5
        res = {}
        settle(step, 0.4, w.sync)  # (the CPU baseline of the leg ran in between: the clocks are down, and 3000 warm-up frames are 45 ms)
        for name, opt in (("two_streams", 0), ("lane_in_order", vfx.OPT_DIRECT_DISPATCH), ("lane", vfx.OPT_DIRECT_DISPATCH | vfx.OPT_DIRECT_UNORDERED)):
            secs, took = (ctypes.c_double * reps)(), ctypes.c_uint64()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_colorlut_direct(w.device_index, 3000, launches, reps, lut.h, fin, fout, nfr, opt, secs, ctypes.byref(took))
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fps = launches * w.world / med
            res[name] = {"value": fps, "unit": "frames/s", "frac_wall": fps / w.world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS,
                         "share_through_the_lane": took.value / float(reps * launches)}
        res["launch_model"] = ("1 thread x single-frame mvfx_colorlut_transform_frame, a fence per frame: on two alternating HIP streams / as packets of the "
                               "library's own queues in queue order / the same without the barrier bit (independent frames)")
        res["statistic"] = "median of 5 repetitions x 1500 frames"
        lib.mvfx_direct_lane_park()  # (the legs that follow run on streams: no idle hardware queues beside them)
        return res
    leg = Leg("colorlut_" + content, "colorlut_frames_per_sec", "frames/s", nb, nb * 2 * FRAME_BYTES, "f32", data,
              f"colorlut 33^3 .cube (575 KB of nodes), {nb} streams of 3840x2160 RGBA per launch, content={content}; 4 + 4 algorithmic B/px "
              "(the LUT gathers are content dependent: random colours are the worst case, flat bars the best); kernels: the x-prelerped window "
              "kernels (6.9 MB table of the four x-lerps + the y-difference per r byte, built once per LUT) -- per-wave windows on calm pictures, "
              "one window per workgroup on busy ones, chosen by a content probe of an earlier frame",
              step, ["colorlut_xtile_kernel", "colorlut_xwg_kernel"],
              cpu=(lambda s: cpu_baseline_colorlut(s, host, cube_text, content)) if host else None)
    def noise_sweep():
        """frames/s of the same batched launch on natural-like frames with +-0 / 3 / 5 / 8 / 16 codes of noise (camera footage is
        noisy: the more codes of noise, the more LUT cells a tile of pixels touches): HIP events around 30 launches each."""
        res = {}
        for amp in (0, 3, 5, 8, 16):
            fill_frames(torch, dev, gen, src, "natural", W, H, first_frame=w.rank * pool * nb, noise=amp)
            for i in range(40):  # (the content probe of the LUT looks at every 32nd launch's frame: its verdict for the new content is in by then)
                step(i)
            secs_, ev_ms_ = w.timed(step, 30, events=True)
            res[str(amp)] = round(nb * w.world / (ev_ms_ * 1e-3))
        return res
    leg.keep = (src, dst, fi, fo, lut)

    def verify():
        import numpy as np
        from tests import oracle_binding as orc
        olut = orc.CubeLut(cube_text)
        step(0)
        w.sync()
        for i in (0, nb - 1):
            a = src[i].cpu().numpy().reshape(H, W * 4)
            want = np.empty_like(a)
            olut.apply(a, W * 4, want, W * 4, W, H, "RGBA")
            _same(dst[i].cpu().numpy().reshape(H, W * 4), want, f"colorlut frame {i}")
        return f"frames 0 and {nb - 1} of the first {nb}-frame step, every byte, against oracle/colorlut_oracle.c"
    leg.verify = verify
    leg.streams_leg = streams_leg
    leg.lane_leg = lane_leg
    leg.noise_sweep = noise_sweep
    return leg


def make_leg_hsv3(w, args, which):
    """The 3-byte formats the headline ignores (hsvfilter/imp.rs:328-371, hsvdetector/imp.rs:422-707): args.batch streams of
    3840x2160 per launch; which = "filter": hsvfilter RGB in place (3 R + 3 W B/px); "detector": hsvdetector RGB -> RGBA (3 R + 4 W)."""
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    W, H, nb = W4K, H4K, args.batch
    stride3 = (W * 3 + 3) & ~3
    pool = max(2, 48 // nb)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0600 + w.rank)
    rgba = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
    fill_frames(torch, dev, gen, rgba, "videotestsrc", W, H, first_frame=w.rank * pool * nb)
    src = rgba.view(pool * nb, H, W, 4)[..., :3].contiguous().view(pool * nb, H * stride3)  # the same smpte frames, packed RGB
    del rgba
    fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, stride3, "RGB") for i in range(nb)]) for b in range(pool)]
    host = [src[k].cpu().numpy().reshape(H, stride3).copy() for k in range(3)] if w.rank == 0 and w.world == 1 else None
    fs = vfx.HsvFilterSettings(*SETTINGS)
    ds = vfx.HsvDetectorSettings(*DETECT_SETTINGS)
    if which == "filter":
        def step(i):  # a filter alone: its output is not read again soon (MVFX_OPT_NONTEMPORAL, as the headline)
            lib.mvfx_thread_set_options(SIDE_NT[0] and vfx.OPT_NONTEMPORAL)
            vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(fi[i % pool], nb, ctypes.byref(fs), sptr))
            lib.mvfx_thread_set_options(0)

        def cpu(seconds):
            import numpy as np
            from tests import oracle_binding as orc

            def make_unit(t):
                work = host[0].copy()
                k = [t]

                def prepare():
                    np.copyto(work, host[k[0] % len(host)])
                    k[0] += 1
                return prepare, lambda: orc.hsvfilter(work, W, stride3, "RGB", SETTINGS)
            return cpu_baseline_of(make_unit, seconds, "frames/s", "3840x2160 RGB frame through orc_hsvfilter_transform_frame_ip (oracle/hsv_oracle.c)")
        leg = Leg("hsvfilter_rgb", "hsvfilter_4k_rgb_frames_per_sec", "frames/s", nb, nb * 2 * W * H * 3, "f32",
                  "synthetic videotestsrc pattern=smpte frames packed to RGB, device-resident",
                  f"hsvfilter 3840x2160 RGB (3 B/px) in place, {nb} streams per launch; 3 + 3 algorithmic B/px", step, ["hsvfilter3_typed_kernel"],
                  cpu=cpu if host else None)
        leg.keep = (src, fi)

        def verify():
            from tests import oracle_binding as orc
            picks = (0, nb - 1)
            before = {i: src[i].cpu().numpy().reshape(H, stride3).copy() for i in picks}
            step(0)
            w.sync()
            for i in picks:
                orc.hsvfilter(before[i], W, stride3, "RGB", SETTINGS)
                _same(src[i].cpu().numpy().reshape(H, stride3), before[i], f"hsvfilter RGB frame {i}")
            return f"frames 0 and {nb - 1} of the first {nb}-frame step, every byte, against oracle/hsv_oracle.c"
        leg.verify = verify
        return leg
    dst = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
    fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]

    def step(i):
        lib.mvfx_thread_set_options(SIDE_NT[0] and vfx.OPT_NONTEMPORAL)
        vfx.check(lib.mvfx_hsvdetector_transform_frames(fi[i % pool], fo[i % pool], nb, ctypes.byref(ds), sptr))
        lib.mvfx_thread_set_options(0)

    def cpu(seconds):
        import numpy as np
        from tests import oracle_binding as orc

        def make_unit(t):
            out = np.empty((H, W * 4), dtype=np.uint8)
            k = [t]

            def unit():
                f = host[k[0] % len(host)]
                k[0] += 1
                orc.hsvdetector(f, stride3, "RGB", out, W * 4, "RGBA", W, DETECT_SETTINGS)
            return (lambda: None), unit
        return cpu_baseline_of(make_unit, seconds, "frames/s", "3840x2160 RGB frame through orc_hsvdetector_transform_frame (oracle/hsv_oracle.c)")
    leg = Leg("hsvdetector_rgb", "hsvdetector_4k_rgb_in_frames_per_sec", "frames/s", nb, nb * W * H * 7, "f32",
              "synthetic videotestsrc pattern=smpte frames packed to RGB, device-resident",
              f"hsvdetector 3840x2160 RGB -> RGBA, {nb} streams per launch; 3 + 4 algorithmic B/px", step, ["hsvdetector3_typed_kernel"],
              cpu=cpu if host else None)
    leg.keep = (src, dst, fi, fo)

    def verify():
        import numpy as np
        from tests import oracle_binding as orc
        step(0)
        w.sync()
        for i in (0, nb - 1):
            a = src[i].cpu().numpy().reshape(H, stride3)
            want = np.empty((H, W * 4), dtype=np.uint8)
            orc.hsvdetector(a, stride3, "RGB", want, W * 4, "RGBA", W, DETECT_SETTINGS)
            _same(dst[i].cpu().numpy().reshape(H, W * 4), want, f"hsvdetector RGB->RGBA frame {i}")
        return f"frames 0 and {nb - 1} of the first {nb}-frame step, every byte, against oracle/hsv_oracle.c"
    leg.verify = verify
    return leg


def make_leg_videofx(w, args):
    """config 4: one 4K stream per GPU: roundedcorners I420 -> A420 compose with the r=100 mask + colordetect on the RGBA twin."""
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    W, H, pool = W4K, H4K, 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0400 + w.rank)
    i420 = torch.randint(0, 256, (pool, W * H * 3 // 2), dtype=torch.uint8, device=dev, generator=gen)
    a420 = torch.empty((pool, W * H * 5 // 2), dtype=torch.uint8, device=dev)
    rgba = torch.empty((pool, FRAME_BYTES), dtype=torch.uint8, device=dev)
    fill_frames(torch, dev, gen, rgba, "natural", W, H, first_frame=w.rank * pool)
    mask = torch.empty(W * H, dtype=torch.uint8, device=dev)
    vfx.check(lib.mvfx_roundedcorners_mask(ctypes.c_void_p(mask.data_ptr()), W, H, W, 100, sptr))
    hist = torch.zeros(32768 + 8, dtype=torch.int32, device=dev)
    offs = [0, W * H, W * H * 5 // 4, W * H * 3 // 2]
    planes = []
    for k in range(pool):
        a, b = vfx.PlanarFrame(), vfx.PlanarFrame()
        for p_ in range(3):
            a.data[p_] = i420[k].data_ptr() + offs[p_]
            b.data[p_] = a420[k].data_ptr() + offs[p_]
            a.stride[p_] = b.stride[p_] = W if p_ == 0 else W // 2
        b.data[3] = a420[k].data_ptr() + offs[3]
        b.stride[3] = W
        a.width = b.width = W
        a.height = b.height = H
        a.format, b.format = vfx.FORMATS["I420"], vfx.FORMATS["A420"]
        planes.append((a, b))
    fr = [vfx.make_frame(rgba[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(pool)]
    host = [rgba[k].cpu().numpy().copy() for k in range(4)] if w.rank == 0 and w.world == 1 else None
    # --element-streams 2: the two elements on their own HIP streams, as with a `queue` between them (two streaming threads:
    # frame k's colordetect overlaps frame k+1's compose); 1: both on one stream, one after the other (one streaming thread)
    # The two streams are the library's (mvfx_thread_stream_n(0 / 1): created one behind the other, so the runtime binds them to different
    # hardware queues) -- what two streaming threads get.  Round 3 found that the torch stream used here before shared a hardware queue
    # with the launch stream: the two elements' kernels never overlapped and the leg measured their SUM instead of the slower one.
    second = args.element_streams == 2
    s_a = ctypes.c_void_p(lib.mvfx_thread_stream_n(0)) if second else sptr
    s_b = ctypes.c_void_p(lib.mvfx_thread_stream_n(1)) if second else sptr

    def step(i):
        k = i % pool
        vfx.check(lib.mvfx_roundedcorners_compose_a420(ctypes.byref(planes[k][0]), ctypes.c_void_p(mask.data_ptr()), W,
                                                       ctypes.byref(planes[k][1]), s_a))
        vfx.check(lib.mvfx_colordetect_histogram(ctypes.byref(fr[k]), 10, 0, vfx.ALL_SAMPLES, ctypes.c_void_p(hist.data_ptr()),
                                                 ctypes.c_void_p(hist.data_ptr() + 32768 * 4), s_b))
    leg = Leg("videofx", "videofx_frames_per_sec", "frames/s", 1, W * H * 4 + FRAME_BYTES, "u8",
              "synthetic: uniform-random I420 planes (compose), natural-like RGBA frames (colordetect), device-resident",
              "roundedcorners I420->A420 compose (r=100; 1.5 R + 2.5 W B/px) + colordetect histogram (quality=10; 4 touched B/px), one "
              "3840x2160 stream per GPU, " + ("both elements on one HIP stream (one streaming thread)" if not second else
                                               "the two elements on their own HIP streams (a queue between them: two streaming threads)"),
              step, ["copy_planes_kernel", "colordetect_hist_kernel"],
              cpu=(lambda s: cpu_baseline_videofx(s, host)) if host else None)
    leg.keep = (i420, a420, rgba, mask, hist, planes, fr)

    def verify():
        import numpy as np
        from tests import oracle_binding as orc
        hist.zero_()
        a420[0].zero_()
        step(0)
        w.sync()
        torch.cuda.synchronize()
        out = a420[0].cpu().numpy()
        _same(out[:offs[3]], i420[0].cpu().numpy(), "roundedcorners A420 planes 0-2 (the I420 planes, copied)")
        _same(out[offs[3]:], mask.cpu().numpy(), "roundedcorners A420 plane 3 (the r=100 mask: libcairo's, tests/golden)")
        rc, want, mm, n = orc.colordetect_histogram(rgba[0].cpu().numpy(), "RGBA", 10)
        got = hist.cpu().numpy()
        _same(got[:32768].view(np.uint32), want.astype(np.uint32), "colordetect histogram (32 768 bins)")
        _same(got[32768:32774].view(np.uint32).tolist(), mm, "colordetect min/max")
        return "step 0: the A420 frame byte for byte (planes = input, alpha = mask) and the 32 768-bin histogram + min/max against oracle/videofx_oracle.c"
    leg.verify = verify
    if second:
        ev_a, ev_b = ctypes.c_void_p(), ctypes.c_void_p()
        vfx.check(lib.mvfx_event_create(ctypes.byref(ev_a)))
        vfx.check(lib.mvfx_event_create(ctypes.byref(ev_b)))

        def join():  # the launch stream waits for what the two element streams have been given so far
            vfx.check(lib.mvfx_event_record(ev_a, s_a))
            vfx.check(lib.mvfx_event_record(ev_b, s_b))
            vfx.check(lib.mvfx_stream_wait_event(sptr, ev_a))
            vfx.check(lib.mvfx_stream_wait_event(sptr, ev_b))
        leg.join = join
        leg.note = ("two HIP streams: avg_step_ms spans K steps with ONE join of the two streams at the end (throughput); step_us joins them after "
                    "every step (latency of one frame through both elements, no overlap between frames)")
    return leg


def make_leg_videocompare(w, args, algo):
    """config 5 on one GPU holding both whole 7680x4320 RGBA frames of every pair."""
    torch, vfx, lib, dev, sptr, stream = w.torch, w.vfx, w.lib, w.dev, w.sptr, w.stream
    W, H = 7680, 4320
    pool = 4 if algo == "blockhash" else 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0500)
    full = torch.empty((pool, 2, H * W * 4), dtype=torch.uint8, device=dev)
    for k in range(pool):
        full[k, 0].random_(0, 256, generator=gen)
    full[:, 1] = full[:, 0]
    full[:, 1, ::97] ^= 0x10          # B = A with ~1 % of the bytes perturbed (SURVEY 8d iv)
    fr = [(vfx.Frame * 2)(*[vfx.make_frame(full[k, p].data_ptr(), W, H, W * 4, "RGBA") for p in range(2)]) for k in range(pool)]
    dist_out = ctypes.c_double()
    last = [0.0]
    host_pair = None
    if w.rank == 0 and w.world == 1:
        if algo == "blockhash":
            host_pair = [full[0, p].cpu().numpy().reshape(H, W * 4).copy() for p in range(2)]
        else:
            cw, ch = 960, 540
            host_pair = [full[0, p].view(H, W * 4)[:ch, :cw * 4].contiguous().cpu().numpy().copy() for p in range(2)]
    bytes_per_pair = 2 * W * H * 4
    if algo == "dssim":
        def step(i):
            vfx.check(lib.mvfx_ssim_distance(ctypes.byref(fr[i % pool][0]), ctypes.byref(fr[i % pool][1]), ctypes.byref(dist_out), sptr))
            last[0] = dist_out.value
        leg = Leg("videocompare_dssim", "videocompare_dssim_8k_rgba_pairs_per_sec", "pairs/s", 1, bytes_per_pair, "f32",
                  "synthetic uniform-random u8 RGBA 8K pairs (B = A with 1 % of the bytes perturbed), device-resident",
                  "videocompare hash-algo=dssim (multi-scale SSIM) on 7680x4320 RGBA pairs, one GPU, synchronous mvfx_ssim_distance per pair; "
                  "4 + 4 compulsory B/px-pair (the planes of the pyramid are implementation traffic)",
                  step, ["ssim_*"], cpu=(lambda s: cpu_baseline_dssim(s, host_pair, 960, 540)) if host_pair else None,
                  fixed_settle=20, extra={"last_distance": last},
                  note="compulsory input bytes against HBM peak; the pyramid planes are extra traffic, see traffic_over_algorithmic")
    else:
        depth = max(1, args.pairs_in_flight)
        ring_host = torch.zeros((depth, 2, 64), dtype=torch.int32).pin_memory()
        # the sums of a pair land in page-locked host memory straight from the reduce kernel's stores, and the pair's fence rides on that
        # kernel (mvfx_thread_set_completion_event): no D2H copy packet and no event record behind it (2.6 us of device time each,
        # tools/probes/event_cost.hip)
        ring_busy = [False] * depth
        ring_evt = []
        for _ in range(depth):
            e = ctypes.c_void_p()
            vfx.check(lib.mvfx_event_create(ctypes.byref(e)))
            ring_evt.append(e)
        def bits(s):
            arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
            out = ctypes.c_uint64()
            vfx.check(lib.mvfx_blockhash_bits(arr, W, H, ctypes.byref(out)))
            return out.value

        def finish(slot):
            vfx.check(lib.mvfx_event_synchronize(ring_evt[slot]))
            h = [bits([int(v) & 0xFFFFFFFF for v in ring_host[slot, p].tolist()]) for p in range(2)]
            ring_busy[slot] = False
            last[0] = float(bin(h[0] ^ h[1]).count("1"))

        def step(i):
            if depth == 1:
                # HasherEngine::hash_image x2 + compare in the C ABI (one launch for both pads, one 512-byte D2H, host bit
                # derivation), exactly what the element does per aggregate
                vfx.check(lib.mvfx_videocompare_distance(ctypes.byref(fr[i % pool][0]), ctypes.byref(fr[i % pool][1]),
                                                         ctypes.byref(dist_out), sptr))
                last[0] = dist_out.value
                return
            # `depth` pairs in flight: the block sums of pair i travel to pinned host memory asynchronously and become hashes
            # / the distance while the kernel of pair i+1 .. i+depth-1 runs
            slot = i % depth
            if ring_busy[slot]:
                finish(slot)
            vfx.check(lib.mvfx_thread_set_completion_event(ring_evt[slot]))
            vfx.check(lib.mvfx_blockhash_sums_pads(fr[i % pool], 2, H, 0, ctypes.c_void_p(ring_host[slot].data_ptr()), sptr))
            if lib.mvfx_thread_clear_completion_event() <= 0:
                vfx.check(lib.mvfx_event_record(ring_evt[slot], sptr))
            ring_busy[slot] = True
        leg = Leg("videocompare_blockhash", "videocompare_blockhash_8k_rgba_pairs_per_sec", "pairs/s", 1, bytes_per_pair, "u32",
                  "synthetic uniform-random u8 RGBA 8K pairs (B = A with 1 % of the bytes perturbed), device-resident",
                  f"videocompare hash-algo=blockhash on 7680x4320 RGBA pairs, one GPU, whole frames, {depth} pair(s) in flight "
                  "(1 = the synchronous mvfx_videocompare_distance the element calls per aggregate); 4 + 4 B/px-pair",
                  step, ["blockhash_sums_kernel", "blockhash_reduce_kernel"],
                  cpu=(lambda s: cpu_baseline_blockhash(s, host_pair)) if host_pair else None, fixed_settle=400,
                  extra={"last_distance": last, "pairs_in_flight": depth},
                  note="per pair incl. the 128 block sums landing in page-locked host memory (the reduce kernel's own stores) and the host bit derivation")
        leg.drain = lambda: [finish(s) for s in range(depth) if ring_busy[s]]
    leg.keep = (full, fr)
    if host_pair is not None and algo == "blockhash":
        def verify():
            from tests import oracle_binding as orc
            step(0)
            leg.drain()
            w.sync()
            _, ha = orc.blockhash(host_pair[0], W, H, W * 4, "RGBA")
            _, hb = orc.blockhash(host_pair[1], W, H, W * 4, "RGBA")
            want = float(orc.hamming(ha, hb))
            if last[0] != want:
                raise AssertionError(f"Hamming distance of pair 0: {last[0]} from the device path, {want} from the oracle")
            out = ctypes.c_uint64()
            for p_, hw in ((0, ha), (1, hb)):  # and the two 64-bit hashes themselves, through the synchronous entry
                vfx.check(lib.mvfx_blockhash(ctypes.byref(fr[0][p_]), ctypes.byref(out), sptr))
                if out.value != hw:
                    raise AssertionError(f"blockhash of frame {p_} of pair 0: {out.value:#018x} from the device path, {hw:#018x} from the oracle")
            return "pair 0: both 64-bit hashes and the Hamming distance of the step's own path against oracle/videofx_oracle.c (bit-exact)"
        leg.verify = verify
    elif host_pair is not None:
        def verify():
            from tests import oracle_binding as orc
            cw, ch = 960, 540
            crop = [full[0, p_].view(H, W * 4)[:ch, :cw * 4].contiguous() for p_ in range(2)]
            fa, fb = (vfx.make_frame(crop[p_].data_ptr(), cw, ch, cw * 4, "RGBA") for p_ in range(2))
            d = ctypes.c_double()
            vfx.check(lib.mvfx_ssim_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), sptr))
            rc, want, _ = orc.ssim_distance(host_pair[0], host_pair[1], cw, ch, cw * 4, cw * 4, "RGBA")
            if rc != 0 or abs(d.value - want) > 1e-5 * abs(want) + 2e-9:
                raise AssertionError(f"dssim of the 960x540 crop of pair 0: {d.value!r} from the device path, {want!r} from the f64 oracle (rc {rc})")
            return ("the 960x540 crop of pair 0 through the same C entry against the f64 restatement oracle/ssim_oracle.c, 1e-5 relative (the tests' tolerance; "
                    "a whole 8K pair is ~18 s of oracle time); the restatement itself is NOT pinned against dssim-core")
        leg.verify = verify
    return leg


def other_config_legs(w, args):
    """BASELINE configs 2-5 measured in the same run as the headline: each becomes its own small JSON line printed before the final
    line (compact_sub) and a full entry of bench_out/last_run.json: value, roofline fractions over the wall clock and over HIP events,
    p10/p50/p90 per step, committed PMC traffic / bound and a bounded CPU-port baseline."""
    torch = w.torch
    out = {}
    import copy
    args_vfx = copy.copy(args)
    args_vfx.element_streams = 2  # config 4 in the driver's run: the two elements on their own streaming threads (HIP streams)
    makers = [("hsv1080p", lambda: make_leg_hsv1080p(w, args)),
              ("colorlut_natural", lambda: make_leg_colorlut(w, args, "natural")),
              ("colorlut_random", lambda: make_leg_colorlut(w, args, "random")),
              ("videofx", lambda: make_leg_videofx(w, args_vfx)),
              ("videocompare_blockhash", lambda: make_leg_videocompare(w, args, "blockhash")),
              ("videocompare_dssim", lambda: make_leg_videocompare(w, args, "dssim")),
              ("hsvfilter_rgb", lambda: make_leg_hsv3(w, args, "filter")),
              ("hsvdetector_rgb", lambda: make_leg_hsv3(w, args, "detector"))]
    steps = {"hsv1080p": 200, "colorlut_natural": 100, "colorlut_random": 40, "videofx": 400, "videocompare_blockhash": 400,
             "videocompare_dssim": 30, "hsvfilter_rgb": 60, "hsvdetector_rgb": 60}
    only = [k for k in args.only_configs.split(",") if k]
    for key, make in makers:
        if only and key not in only:
            continue
        t0 = time.perf_counter()
        try:
            leg = make()
            t_made = time.perf_counter() - t0
            k = steps[key]
            r = measure_leg(w, leg, k, max(5, k // 10), args.other_settle_seconds, 200 if key != "videocompare_dssim" else 40,
                            args.other_cpu_seconds)
            if hasattr(leg, "drain"):
                leg.drain()
            if key == "colorlut_natural" and args.stream_threads > 0:
                r["config"]["other_launch_model"] = leg.streams_leg()
                r["config"]["element_model"] = r["config"]["other_launch_model"]["value"]
            if key == "colorlut_natural" and args.stream_threads > 0:
                try:
                    one = leg.lane_leg()
                    r["config"]["one_frame_per_call"] = one
                    r.setdefault("sub_extra", {})["one_frame_per_call_fps"] = {k: round(one[k]["value"]) for k in ("two_streams", "lane_in_order", "lane")}
                except Exception as e:  # noqa: BLE001
                    r["config"]["one_frame_per_call"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            if key == "colorlut_natural" and args.noise_sweep:
                r.setdefault("sub_extra", {})["noise_fps"] = leg.noise_sweep()
            if "last_distance" in r["config"]:
                r["config"]["last_distance"] = r["config"]["last_distance"][0]
            r["measure_seconds"]["make"] = t_made
            out[key] = r
            del leg
        except Exception as e:  # a failing side leg must not cost the headline line
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out[key]["leg_seconds"] = time.perf_counter() - t0
    return out


# ------------------------------------------------------------------------------------------------ config 5, sharded (N > 1)

def videocompare_main(args):
    """BASELINE config 5: distance of 7680x4320 RGBA frame pairs.  N == 1: whole frames on the one GPU.  N > 1: inputs are
    pre-sharded, rank r holds block-row band r of both frames; one all-reduce of 2x64 sums per pair."""
    w = Worker(args)
    # MVFX_BENCH_FORCE_SHARDED_LEG=1: the band-sharded leg with ONE rank (the library's RCCL communicator of world size 1): the code the
    # driver's N > 1 run executes, on the one-GPU box
    if w.world == 1 and os.environ.get("MVFX_BENCH_FORCE_SHARDED_LEG") != "1":
        leg = make_leg_videocompare(w, args, args.hash_algo)
        r = measure_leg(w, leg, args.steps, args.warmup, args.settle_seconds, args.pct_steps, 0 if args.no_cpu_baseline else args.other_cpu_seconds)
        if hasattr(leg, "drain"):
            leg.drain()
        r["config"]["last_distance"] = r["config"]["last_distance"][0]
        r.update({"n_gpus": 1, "higher_is_better": True, "vs_baseline": None})
        r["config"].update({"parallelism": "one GPU, whole frames", "rccl_ranks": w.rccl_ranks})
        emit(r, full=bool(args.full))
        w.finish()
        return
    out = videocompare_sharded_leg(w, args, args.hash_algo, args.steps, args.warmup)
    if w.rank == 0:
        emit(out, full=bool(args.full))
    w.finish()


def videocompare_sharded_leg(w, args, algo, steps, warmup):
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    from gst_plugin_rs_amd import distributed as D
    rank, world = w.rank, w.world
    W, H = 7680, 4320
    r0, r1 = D.band_rows(H, rank, world)
    rows = r1 - r0
    gen = torch.Generator(device=dev)
    if algo == "dssim":
        # every rank holds both full frames (a band's 5-level pyramid needs a halo of up to 64 rows) and maps only its band;
        # two all-reduces of 10 f64 per pair (distributed.ssim_sharded)
        pool = 2
        gen.manual_seed(0x5EED0002)
        full = torch.randint(0, 256, (pool, 2, H * W * 4), dtype=torch.uint8, device=dev, generator=gen)
        full[:, 1] = full[:, 0]
        full[:, 1, ::97] ^= 0x10
        y0, y1 = D.ssim_band_rows(H, rank, world)
        fr = [[vfx.make_frame(full[k, p].data_ptr(), W, H, W * 4, "RGBA") for p in range(2)] for k in range(pool)]

        comm = D.make_comm(vfx, rank, world)  # the library's own RCCL communicator (its id travels over the torch group)

        def step(i):
            k = i % pool
            # band maps -> ncclAllReduce(10 f64) -> band deviations -> ncclAllReduce(5 f64) -> combine, all inside the C entry
            return [vfx.videocompare_sharded_dssim(comm, fr[k][0], fr[k][1], y0, y1, sptr)]
    else:
        pool = 4
        gen.manual_seed(0x5EED0001)  # same seed on every rank: band r of the same virtual frames
        pairs = torch.randint(0, 256, (pool, 2, rows * W * 4), dtype=torch.uint8, device=dev, generator=gen)
        bands = [(vfx.Frame * 2)(*[vfx.make_frame(pairs[k, p].data_ptr(), W, rows, W * 4, "RGBA") for p in range(2)])
                 for k in range(pool)]
        comm = D.make_comm(vfx, rank, world)  # the library's own RCCL communicator (its id travels over the torch group)

        def step(i):
            # band kernel -> ncclAllReduce(2 x 64 u32) -> bits + Hamming on the device -> 4-byte D2H, all inside the C entry
            return vfx.videocompare_sharded_distances(comm, bands[i % pool], H, r0, sptr)

    settle(step, args.settle_seconds, w.sync, fixed_steps=200 if algo == "blockhash" else 10)  # all-reduce inside the step
    for i in range(warmup):
        step(i)
    result = [None]

    def timed_step(i):
        result[0] = step(i)
    elapsed, _ = w.timed(timed_step, steps)
    (elapsed,) = w.max_over_ranks(elapsed)
    # the collective alone: the same 2x64 u32 all-reduce with nothing around it (latency-bound on xGMI), through the library
    t = torch.zeros((2, 64), dtype=torch.int32, device=dev)
    t64 = torch.zeros(10, dtype=torch.float64, device=dev)
    if algo == "blockhash":
        one_ar = lambda i: comm.allreduce(t.data_ptr(), 128, vfx.DTYPE_U32, vfx.REDUCE_SUM, sptr)
    else:
        one_ar = lambda i: comm.allreduce(t64.data_ptr(), 10, vfx.DTYPE_F64, vfx.REDUCE_SUM, sptr)
    for i in range(20):
        one_ar(i)
    ar_s, _ = w.timed(one_ar, 200)
    (ar_s,) = w.max_over_ranks(ar_s)
    w.sync()
    comm.destroy()
    bytes_per_pair = 2 * W * H * 4
    achieved = bytes_per_pair * steps / elapsed / 1e9
    return {
        "metric": "videocompare_8k_rgba_pairs_per_sec", "value": steps / elapsed, "unit": "pairs/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32" if algo == "blockhash" else "f32",
        "data": "synthetic uniform-random u8 RGBA, device-resident, rows pre-sharded by block-row band",
        "config": {"workload": "videocompare blockhash 7680x4320 RGBA pair, band-sharded + all-reduce(2x64 u32)" if algo == "blockhash"
                   else "videocompare dssim (multi-scale SSIM) 7680x4320 RGBA pair, row bands + 2 all-reduces of 10 f64",
                   "parallelism": f"{world} row bands, RCCL all-reduce per pair", "last_distance": result[0][0],
                   "rccl_ranks": w.rccl_ranks, "allreduce_us": ar_s / 200 * 1e6,
                   "allreduce_note": "RCCL all-reduce(sum) of 2x64 u32 (blockhash) / 10 f64 (dssim) alone (mvfx_comm_allreduce: ncclAllReduce inside "
                                     "libmi355vfx), back to back on the launch stream, wall clock / 200",
                   "collective": "ncclAllReduce inside mvfx_videocompare_sharded_distances (library-owned communicator), hash bits + Hamming "
                                 "distance on the device, one 4-byte D2H per pair" if algo == "blockhash" else
                                 "two ncclAllReduce (10 + 5 f64) inside mvfx_videocompare_sharded_dssim (library-owned communicator)"},
        "roofline": {"bound": "hbm" if algo == "blockhash" else "valu", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                     "note": "end-to-end per pair incl. the all-reduce, the D2H of the block sums, the synchronisation and host bit derivation"}}


# ------------------------------------------------------------------------------------------------ configs 2-4

def config_main(args):
    """One of BASELINE configs 2-4 as its own line (device-resident per-GPU stream workloads, no data-path collective)."""
    w = Worker(args)
    if args.workload == "hsv1080p":
        leg = make_leg_hsv1080p(w, args)
    elif args.workload in ("hsvfilter_rgb", "hsvdetector_rgb"):
        leg = make_leg_hsv3(w, args, "filter" if args.workload == "hsvfilter_rgb" else "detector")
    elif args.workload == "colorlut":
        leg = make_leg_colorlut(w, args, args.content)
    else:
        leg = make_leg_videofx(w, args)
    r = measure_leg(w, leg, args.steps, args.warmup, args.settle_seconds, args.pct_steps, 0 if args.no_cpu_baseline else args.other_cpu_seconds)
    if args.workload == "colorlut" and args.stream_threads > 0:
        r["config"]["other_launch_model"] = leg.streams_leg()
    r.update({"n_gpus": w.world, "higher_is_better": True, "vs_baseline": None})
    r["config"].update({"parallelism": f"{w.world} independent streams", "rccl_ranks": w.rccl_ranks})
    if w.rank == 0:
        emit(r, full=bool(args.full))
    w.finish()


# ------------------------------------------------------------------------------------------------ headline

def hsvfilter_main(args):
    t_start = time.perf_counter()
    timing = {}
    w = Worker(args)
    timing["worker_init"] = time.perf_counter() - t_start
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    rank, world = w.rank, w.world
    opts = vfx.options(variant=args.variant, nontemporal=bool(args.streaming), typed=bool(args.typed_loads)).word
    vfx.check(lib.mvfx_thread_set_options(opts))

    # ---- resident frame pool: (pool + 2 scratch) x batch distinct 4K RGBA frames of --frame-content ---------
    pool = max(1, args.pool)
    n_scratch = 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0100 + rank)
    frames = torch.empty((pool + n_scratch, args.batch, FRAME_BYTES), dtype=torch.uint8, device=dev)
    flat_frames = frames.view(-1, FRAME_BYTES)
    # rank r's shard = its own streams: with videotestsrc content every rank starts further down the snow sequence
    fill_frames(torch, dev, gen, flat_frames, args.frame_content, W4K, H4K, first_frame=rank * flat_frames.shape[0])
    settings = vfx.HsvFilterSettings(*SETTINGS)
    frame_arrays = []
    for b in range(pool + n_scratch):
        arr = (vfx.Frame * args.batch)(*[
            vfx.make_frame(frames[b, i].data_ptr(), W4K, H4K, W4K * 4, "RGBA") for i in range(args.batch)])
        frame_arrays.append(arr)

    n_launches = [0]

    def launch(batch_index):
        n_launches[0] += 1
        rc = lib.mvfx_hsvfilter_transform_frames_ip(frame_arrays[batch_index], args.batch, ctypes.byref(settings), sptr)
        if rc != 0:
            raise RuntimeError(f"mvfx status {rc}: {vfx.last_error()}")

    def scratch_step(i):   # settle + warm-up: only the scratch batches are filtered (2 x batch x 33 MB >> Infinity Cache)
        launch(pool + (i % n_scratch))

    def step(i):           # timed: batch i of the untouched pool
        launch(i % pool)

    def batch_leg():
        """settle + W warm-up launches on the scratch batches, then K timed launches on untouched pool batches."""
        n_settle = settle(scratch_step, args.settle_seconds, w.sync)
        if args.converged_data:  # A/B only: filter the timed pool a few times first (what round 1 timed without saying so)
            for k in range(args.converged_data):
                for b in range(pool):
                    launch(b)
            w.sync()
        for i in range(args.warmup):
            scratch_step(i)
        secs, k_ms = w.timed(step, args.steps, events=True)
        secs, k_ms = w.max_over_ranks(secs, k_ms)
        return n_settle, secs, k_ms

    def verify():
        """one launch of the timed entry on the first scratch batch, two of its frames compared byte for byte with the oracle"""
        from tests import oracle_binding as orc
        picks = (0, args.batch - 1)
        before = {i: frames[pool, i].cpu().numpy().reshape(H4K, W4K * 4).copy() for i in picks}
        launch(pool)
        w.sync()
        for i in picks:
            orc.hsvfilter(before[i], W4K, W4K * 4, "RGBA", SETTINGS)
            _same(frames[pool, i].cpu().numpy().reshape(H4K, W4K * 4), before[i], f"hsvfilter frame {i}")
        return (f"frames 0 and {args.batch - 1} of one {args.batch}-frame launch of the timed entry (a scratch batch, before the settle), every byte, "
                "against oracle/hsv_oracle.c")

    class _Headline:
        pass
    _Headline.verify = staticmethod(verify)
    verified = run_verify(w, _Headline) if not args.no_verify else {"verified": None}
    n_launches[0] = 0
    settle_steps, elapsed, kernel_ms = batch_leg()
    batch_fps_rank = w.gather(args.steps * args.batch / elapsed)
    batch_fps = args.steps * args.batch * world / elapsed
    # per-launch spread (median + p10/p90 over >= 200 iterations): the launches continue through the pool with a HIP
    # event between every two; launches beyond the first `pool` re-filter frames (videotestsrc bars stay bars)
    launch_pct = percentiles(w.event_times(step, args.pct_steps, first_index=args.steps)) if args.pct_steps > 0 else None

    # ---- the element's launch model: --batch host threads x own HIP stream x single-frame calls -----------
    streams = combined = single_stream = None
    if args.stream_threads > 0:
        hb = bench_harness()
        nthr = args.stream_threads
        fpt = max(2, (pool * args.batch) // nthr)          # frames per thread, all from the resident pool
        flat = (vfx.Frame * (nthr * fpt))(*[
            vfx.make_frame(frames[(k // args.batch) % pool, k % args.batch].data_ptr(), W4K, H4K, W4K * 4, "RGBA")
            for k in range(nthr * fpt)])
        # warm-up on the scratch batches (as the batch leg): the timed launches start on pool frames that have been through the
        # filter as often as the batch leg left them, not ~70 more times
        wfpt = max(1, (n_scratch * args.batch) // nthr)
        warm = (vfx.Frame * (nthr * wfpt))(*[
            vfx.make_frame(frames[pool + (k // args.batch) % n_scratch, k % args.batch].data_ptr(), W4K, H4K, W4K * 4, "RGBA")
            for k in range(nthr * wfpt)])
        # at least 200 launches per thread: the K x batch frames of the batch leg (320 at the driver's K=20) would be ~20
        # launches per thread = 5 ms, dominated by thread wake-up skew
        launches = max(200, args.steps * args.batch // nthr)
        # the threads create their streams first (GPU idle for several ms -> clocks drop): own ~0.4 s warm-up on the threads
        stream_warmup = max(20, args.warmup, int(args.settle_seconds / 0.6 * 28000) // nthr)
        reps = 5  # the median of five back-to-back repetitions: a single 40 ms window is at the mercy of one descheduled thread

        def threads_leg(batch_arg, what):
            secs = (ctypes.c_double * reps)()
            per = (ctypes.c_double * nthr)()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_hsvfilter_streams_warm(w.device_index, nthr, stream_warmup, launches, reps, flat, fpt, batch_arg, warm, wfpt,
                                                     ctypes.byref(settings), opts, secs, per)
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            rep_secs = sorted(secs)
            (s_elapsed,) = w.max_over_ranks(rep_secs[reps // 2])
            s_fps = nthr * launches * world / s_elapsed
            return {"launch_model": what, "value": s_fps, "unit": "frames/s", "frames": nthr * launches, "launches_per_thread": launches,
                    "warmup_launches_per_thread": stream_warmup, "seconds": s_elapsed, "statistic": "median of 5 repetitions",
                    "achieved_GBs": s_fps / world * 2 * FRAME_BYTES / 1e9, "frac_wall": s_fps / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS,
                    "repetitions_frames_per_sec": [round(nthr * launches / t) for t in secs],
                    "per_rank_frames_per_sec": w.gather(nthr * launches / rep_secs[reps // 2])}

        streams = threads_leg(1, f"{nthr} threads x 1 frame (own HIP stream each, single-frame mvfx_hsvfilter_transform_frame_ip, "
                                 "no sync between launches; warm-up on scratch frames)")
        # ONE video stream: one host thread making single-frame calls, on one private HIP stream and alternating between two (what the
        # GStreamer elements do per buffer since round 3, MVFX_ELEMENT_STREAMS = 2: the next frame's head overlaps the previous one's tail).
        # It runs AFTER the 16-thread leg on purpose: its thread then gets a pair of streams that an exited thread left in the library's pool,
        # and a pair that shares a hardware queue would not overlap (capi_common.hip, StreamBundle)
        def one_thread_leg(streams_per_thread):
            n1 = 2000
            secs = (ctypes.c_double * reps)()
            per = (ctypes.c_double * 1)()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_hsvfilter_streams_rot(w.device_index, 1, streams_per_thread, 200, n1, reps, flat, nthr * fpt, warm, nthr * wfpt,
                                                    ctypes.byref(settings), opts, secs, per)
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fps1 = n1 * world / med
            return {"launch_model": f"1 thread x single-frame calls alternating between {streams_per_thread} private HIP stream(s)", "value": fps1,
                    "unit": "frames/s", "launches": n1, "statistic": "median of 5 repetitions",
                    "frac_wall": fps1 / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
        single_stream = {"one_stream": one_thread_leg(1), "two_streams": one_thread_leg(2)}

        def pairs_leg():
            """what the hsvfilter ELEMENT does since round 4: one thread, two consecutive frames per launch, pairs alternating between
            two streams (the call per buffer returns at once; host/gst/gsthsv.cpp)"""
            n1 = 2000
            secs = (ctypes.c_double * reps)()
            per = (ctypes.c_double * 1)()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_hsvfilter_streams_rot_batched(w.device_index, 1, 2, 100, n1 // 2, reps, flat, (nthr * fpt) // 2 * 2, 2,
                                                            ctypes.byref(settings), opts, secs, per)
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fps1 = n1 * world / med
            return {"launch_model": "1 thread x 2 frames per launch, launches alternating between 2 private HIP streams (the element's model)",
                    "value": fps1, "unit": "frames/s", "frames": n1, "statistic": "median of 5 repetitions",
                    "frac_wall": fps1 / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
        single_stream["pairs_two_streams"] = pairs_leg()

        def threads_pairs_leg():
            """the same model on all streams at once: --stream-threads elements, each launching pairs on its two streams"""
            n1 = max(200, args.steps * args.batch // nthr) // 2 * 2
            secs = (ctypes.c_double * reps)()
            per = (ctypes.c_double * nthr)()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_hsvfilter_streams_rot_batched(w.device_index, nthr, 2, 100, n1 // 2, reps, flat, fpt // 2 * 2, 2,
                                                            ctypes.byref(settings), opts, secs, per)
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fpsn = nthr * n1 * world / med
            return {"launch_model": f"{nthr} threads x 2 frames per launch, each alternating between 2 private HIP streams", "value": fpsn,
                    "unit": "frames/s", "statistic": "median of 5 repetitions", "frac_wall": fpsn / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
        def direct_leg(options):
            """the element's contract through the direct-dispatch lane (round 6, MVFX_OPT_DIRECT_DISPATCH): one thread, one single-frame call per
            buffer, a fence per frame, the kernels on the library's own two HSA queues as AQL packets without a release fence (write-through stores)"""
            n1 = 2000
            secs = (ctypes.c_double * reps)()
            took = ctypes.c_uint64()
            w.sync()
            w.barrier()
            # (its own settle: the leg starts a thread, builds the lane on first use and creates its events -- milliseconds of idle GPU during which the
            # clock governor steps down, and 200 launches are 2.4 ms)
            rc = hb.mvfxbench_hsvfilter_direct(w.device_index, 12000, n1, reps, flat, nthr * fpt, ctypes.byref(settings), options, secs, ctypes.byref(took))
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fps1 = n1 * world / med
            return {"launch_model": "1 thread x single-frame calls through the direct-dispatch lane (own HSA queues, AQL packets with acquire = agent, release = none, "
                                    "write-through stores; a completion signal per frame as its fence)", "value": fps1, "unit": "frames/s", "launches": n1,
                    "statistic": "median of 5 repetitions", "share_through_the_lane": took.value / float(reps * n1),
                    "frac_wall": fps1 / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
        try:
            hb.mvfxbench_hsvfilter_direct.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                                      ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
            single_stream["direct_lane"] = direct_leg(opts)                       # the cache policy of the other legs (non-temporal loads by default)
            single_stream["direct_lane_cached"] = direct_leg(opts & ~vfx.OPT_NONTEMPORAL)  # what the element uses (the next element reads the frame)
        except Exception as e:  # noqa: BLE001  (a box without the lane still prints its line)
            single_stream["direct_lane"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # the lane's two hardware queues go again: idle ones beside HIP's four slow kernels on busy HIP streams down (csrc/direct_dispatch.h, "PARKING"),
        # and the legs that follow run on streams
        lib.mvfx_direct_lane_park()
        if fpt >= 2:
            single_stream["threads_pairs"] = threads_pairs_leg()
        if args.combiner_legs:  # (0: a profiling run -- rocprofv3's queue interceptor crashes on this leg's cross-stream event waits)
            # the launch combiner: the same threads make the same single-frame calls, the library coalesces them into batched launches
            nb, nf = ctypes.c_uint64(), ctypes.c_uint64()
            lib.mvfx_combiner_stats(w.device_index, ctypes.byref(nb), ctypes.byref(nf))
            combined = threads_leg(0, f"{nthr} threads x 1 frame through the launch combiner (mvfx_hsvfilter_transform_frame_ip_combined: one call "
                                      "per buffer, frames of all threads coalesced into batched launches by the library)")
            nb2, nf2 = ctypes.c_uint64(), ctypes.c_uint64()
            lib.mvfx_combiner_stats(w.device_index, ctypes.byref(nb2), ctypes.byref(nf2))
            combined["frames_per_combined_launch"] = (nf2.value - nf.value) / max(nb2.value - nb.value, 1)
            # ... and its fenced entry (what the element uses with MVFX_COMBINE=2): no caller streams, the frames' fences in and out, all
            # combined launches on one library-owned stream
            fenced = threads_leg(0xFFFFFFFF, f"{nthr} threads x 1 frame through the launch combiner's fenced entry (mvfx_hsvfilter_transform_frame_ip_fenced: one "
                                              "call per buffer, the buffer's fence in and out, batched launches on one library-owned stream)")
            nb3, nf3 = ctypes.c_uint64(), ctypes.c_uint64()
            lib.mvfx_combiner_stats(w.device_index, ctypes.byref(nb3), ctypes.byref(nf3))
            fenced["frames_per_combined_launch"] = (nf3.value - nf2.value) / max(nb3.value - nb2.value, 1)
            combined["fenced_entry"] = fenced

    # ---- the same batch leg on the other frame contents: the kernel has no data-dependent branch, but the chip is power
    # limited on this kernel and the bytes decide how much the data paths toggle (tools/exp_content_power.py) -------------
    sweep = {}
    if args.content_sweep:
        for kind in FRAME_CONTENTS:
            if kind == args.frame_content:
                continue
            fill_frames(torch, dev, gen, flat_frames, kind, W4K, H4K, first_frame=rank * flat_frames.shape[0])
            _, sw_secs, sw_ms = batch_leg()
            sweep[kind] = {"value": args.steps * args.batch * world / sw_secs, "unit": "frames/s", "avg_launch_ms": sw_ms,
                           "frac_kernel": args.batch * 2 * FRAME_BYTES / (sw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "frac_wall": args.steps * args.batch * 2 * FRAME_BYTES / sw_secs / 1e9 / HBM_PEAK_GBS, "data": CONTENT_TEXT[kind]}
    # measured after the timed legs: a burst of other kernels between settle and the timed steps leaves the governor in another
    # power state (the timed kernels then ran 3-4 % slower: profiles/r2/ab_fresh_vs_converged_data.txt)
    settle(scratch_step, min(args.settle_seconds, 0.3), w.sync)
    ceiling = measured_rmw_ceiling(w, frames.data_ptr(), args.batch * FRAME_BYTES, pool + n_scratch)
    bytes_per_launch = args.batch * 2 * FRAME_BYTES  # 4 B read + 4 B written per pixel
    batch_model = {"launch_model": f"1 launch x {args.batch} frames (mvfx_hsvfilter_transform_frames_ip, blockIdx.z = stream)",
                   "value": batch_fps, "unit": "frames/s", "per_rank_frames_per_sec": batch_fps_rank}
    use_streams = args.launch_model == "streams" and streams is not None
    head, other = (streams, batch_model) if use_streams else (batch_model, streams)
    total_frames = args.steps * args.batch * world
    ceil_gbs = max(ceiling["in_place_nt"]["GBs"], ceiling["in_place_write_through"]["GBs"])  # the better of the two streaming shapes
    kernel_name = "hsvfilter4_typed_kernel" if args.typed_loads else "hsvfilter4_kernel<RGBA, vec4>"
    # wall seconds per launch of the model `value` reports (per GPU: every rank runs its own launches)
    wall_per_launch = args.batch * world / head["value"]
    roof = roofline_of("hsvfilter", args.batch, bytes_per_launch, wall_per_launch, kernel_ms * 1e-3, kernel_name, launch_pct)
    # `bound`: the roofline `achieved` / `peak` are priced against -- HBM; this path has no MFMA work.  What the counters of the committed rocprofv3 passes
    # say beside it (the kernel sits on the VALU/HBM ridge: VALUBusy above 100 % AND 0.94 of what its memory shape reaches with trivial arithmetic)
    # stays in `bound_counters`.
    roof["bound_counters"] = roof["bound"]
    roof["bound"] = "hbm"
    roof.update({"avg_launch_ms": kernel_ms, "launch_us": launch_pct, "bytes_per_launch": bytes_per_launch,
                 "read_side_GBs": roof["achieved_kernel"] / 2, "ceiling_measured_GBs": ceil_gbs,
                 "frac_of_measured_ceiling": roof["achieved_kernel"] / ceil_gbs if ceil_gbs else None, "ceilings": ceiling,
                 "ceiling_note": "in-tree RMW probe (gst-plugin-rs_amd/bench/probe_rmw.hip): the kernel's own memory shape -- one 16-byte "
                                 "load + store per lane, in place, trivial arithmetic -- over the same resident pool, timed with HIP events in this "
                                 "run after the timed legs; the better of non-temporal load + store (rounds 1-5) and cached load + write-through "
                                 "store (what the kernel does since round 6); in_place_cached / out_of_place_nt are the shapes of the other filters",
                 "launch_model": batch_model["launch_model"]})
    # what ONE drop-in hsvfilter element does (one call per buffer, hsvfilter/imp.rs:322-326), beside the batched entry `value` reports
    element_path = None
    if streams is not None:
        element_path = {"threads16_fps": _r(streams["value"], 5), "threads16_frac": _r(streams["frac_wall"], 4)}
        if single_stream:
            element_path.update({"one_thread_fps": _r(single_stream["two_streams"]["value"], 5),
                                 "one_thread_frac": _r(single_stream["two_streams"]["frac_wall"], 4)})
            if "pairs_two_streams" in single_stream:
                element_path.update({"one_thread_pairs_fps": _r(single_stream["pairs_two_streams"]["value"], 5),
                                     "one_thread_pairs_frac": _r(single_stream["pairs_two_streams"]["frac_wall"], 4)})
            if single_stream.get("direct_lane_cached", {}).get("value"):
                element_path.update({"one_thread_direct_fps": _r(single_stream["direct_lane_cached"]["value"], 5),
                                     "one_thread_direct_frac": _r(single_stream["direct_lane_cached"]["frac_wall"], 4)})
            if "threads_pairs" in single_stream:
                element_path.update({"threads16_pairs_fps": _r(single_stream["threads_pairs"]["value"], 5),
                                     "threads16_pairs_frac": _r(single_stream["threads_pairs"]["frac_wall"], 4)})
    out = {
        "metric": "hsvfilter_4k_rgba_frames_per_sec",
        "value": head["value"],
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": (total_frames / head["value"]) / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": (f"synthetic {args.frame_content} 4K RGBA frames, device-resident; timed steps start on frames no kernel has touched"
                 if not args.converged_data else
                 f"A/B: {args.frame_content} frames filtered {args.converged_data}x before the timed steps (converged, low-entropy)"),
        "data_detail": CONTENT_TEXT[args.frame_content],
        "config": {"workload": "hsvfilter 3840x2160 RGBA in place, hue-shift=90 saturation-mul=1.25 "
                               "saturation-off=-0.05 value-mul=0.9 value-off=0.02",
                   "frame_content": args.frame_content, "other_frame_contents": sweep,
                   "launch_model": head["launch_model"], "other_launch_model": other, "combined_launch_model": combined,
                   "one_video_stream_launch_models": single_stream, "element_path": element_path,
                   "value_p50": args.batch * world / (launch_pct["p50"] * 1e-6) if launch_pct else None,
                   "frames_per_step_per_gpu": args.batch, "resident_batches": pool, "steps_executed": n_launches[0],
                   "settle_seconds_before_warmup": args.settle_seconds, "settle_steps": settle_steps,
                   "parallelism": f"{world} independent stream shards, no data-path collective",
                   "rccl_ranks": w.rccl_ranks, "rendezvous_backend": w.backend if world > 1 else None,
                   "per_rank_frames_per_sec": head["per_rank_frames_per_sec"],
                   "kernel_variant": {0: "auto", 1: "literal", 2: "strength-reduced"}[args.variant],
                   "cache_policy": "non-temporal (MVFX_OPT_NONTEMPORAL)" if args.streaming else "default",
                   "u8_to_unit_float": "typed buffer loads (texture-unit UNORM8, exact)" if args.typed_loads else "VALU (cvt + mul + fmac)"},
        "roofline": roof,
    }
    out.update(verified)
    timing["headline_gpu_legs"] = time.perf_counter() - t_start
    del frames, flat_frames, frame_arrays
    torch.cuda.empty_cache()
    subs = []
    if args.other_configs and world == 1:
        vfx.check(lib.mvfx_thread_set_options(0))
        t1 = time.perf_counter()
        ALL_CORES_SECONDS[0] = args.other_cpu_all_seconds
        others = other_config_legs(w, args)
        out["config"]["other_configs"] = others
        subs = list(others.items())
        timing["other_configs"] = time.perf_counter() - t1
        if args.gst_pipeline:
            t1 = time.perf_counter()
            subs.append(("gst_element_pipeline", gst_pipeline_leg(args)))
            out["config"]["gst_element_pipeline"] = subs[-1][1]
            timing["gst_pipeline"] = time.perf_counter() - t1
    elif args.other_configs and world > 1:
        # the one workload with a data-path collective, on the real xGMI fabric: 8K pairs band-sharded over the ranks, the all-reduce
        # inside libmi355vfx (its own RCCL communicator).  It runs under a watchdog: this leg has never seen more than one GPU before
        # the driver's scaling run, and a rank stuck in a rendezvous must not cost the headline line.
        vfx.check(lib.mvfx_thread_set_options(0))
        import threading
        box = {}

        def side_leg():
            if os.environ.get("MVFX_BENCH_TEST_STUCK_RANK") == str(rank):  # tests/test_distributed_gpu.py: a rank that never comes back
                time.sleep(10 ** 6)
            if w.backend != "nccl":  # the two-ranks-on-one-GPU test mode: RCCL does not run two ranks on one device, nothing to measure
                box["r"] = box["d"] = {"error": "not run: the ranks share one GPU (MVFX_BENCH_TEST_SHARED_GPU), RCCL needs a device per rank"}
                return
            try:
                torch.cuda.set_device(w.device_index)
                vfx.check(lib.mvfx_set_device(w.device_index))
                box["r"] = videocompare_sharded_leg(w, args, "blockhash", 200, 20)
            except Exception as e:  # noqa: BLE001
                box["r"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # BASELINE config 5 proper: the SSIM distance of 8K pairs, rows shared out over the ranks
                box["d"] = videocompare_sharded_leg(w, args, "dssim", 30, 5)
            except Exception as e:  # noqa: BLE001
                box["d"] = {"error": f"{type(e).__name__}: {e}"}

        th = threading.Thread(target=side_leg, daemon=True)
        # every rank gets here within the skew of the headline's closing barrier: the common deadline for the "who is stuck" flags is this
        # instant + the watchdog + slack (advisor r5: 60 s per missing rank was shorter than the 180 s watchdog of the other ranks)
        flags_deadline = time.monotonic() + args.side_leg_timeout + SIDE_LEG_FLAG_SLACK_S
        th.start()
        th.join(timeout=args.side_leg_timeout)
        late = {"error": f"no result within {args.side_leg_timeout} s (watchdog)"}
        sides = {"videocompare_blockhash_sharded": box.get("r", late), "videocompare_dssim_sharded": box.get("d", late)}
        out["config"]["other_configs"] = sides
        subs = list(sides.items())
        # A rank stuck in the collective is a FAILURE of the run: the ranks agree over the rendezvous store (plain TCP, no GPU
        # collective -- the device queue of a stuck rank may never drain), rank 0 still prints the headline (with an `error` entry),
        # and every rank leaves with a non-zero code so that the launcher / the spawning parent reports the run as failed.
        stuck_ranks = agree_on_stuck(w, th.is_alive(), args.side_leg_timeout, deadline=flags_deadline)
        if stuck_ranks:
            if rank == 0:
                out["config"]["error"] = f"side leg stuck in its collective on rank(s) {stuck_ranks}; exit code {EXIT_SIDE_LEG_STUCK}"
                out["wall_s"] = time.perf_counter() - t_start
                emit(out, subs, full=bool(args.full))
                mark_emitted(w)
            else:  # rank 0 prints once ITS flag loop is through, which is the common deadline at the latest
                wait_emitted(w, max(20.0, flags_deadline - time.monotonic() + 20.0))
            os._exit(EXIT_SIDE_LEG_STUCK)  # the stuck thread holds the communicator: no orderly teardown; never exec from here
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t1 = time.perf_counter()
        ALL_CORES_SECONDS[0] = args.cpu_all_seconds
        out["cpu_baseline"] = cpu_baseline_hsvfilter(args.cpu_seconds, args.frame_content)
        timing["cpu_baseline"] = time.perf_counter() - t1
    if world > 1:  # every rank empties its C stdio first (RCCL's banner), so that rank 0's line is the last thing on the launcher's stdout
        _flush_c_stdio()
        sys.stdout.flush()
        w.barrier()
    if rank == 0:
        out["wall_s"] = time.perf_counter() - t_start
        out["timing_s"] = timing
        emit(out, subs, full=bool(args.full))
    w.finish()


EXIT_SIDE_LEG_STUCK = 5
SIDE_LEG_FLAG_SLACK_S = 30.0  # beyond --side-leg-timeout: how long a rank waits for the other ranks' "stuck / not stuck" flags


def _store(w):
    from torch.distributed import distributed_c10d
    return distributed_c10d._get_default_store()


def agree_on_stuck(w, stuck, patience_s, store=None, deadline=None, clock=time.monotonic):
    """Every rank publishes whether its side-leg thread is still alive; returns the sorted list of stuck ranks as every rank sees it.
    A rank that never publishes (dead, or hung before this point) counts as stuck.  Keys live in the rendezvous store: no collective.

    ONE deadline is shared by the whole loop (`deadline`, on `clock`; default now + patience_s): the ranks' watchdogs run independently, so a
    rank whose leg fails at once publishes up to --side-leg-timeout before the ranks that sit out their watchdog in the rendezvous -- it has to
    wait for THEIR flags that long, or it would count them as never arrived, leave early and (a launcher ends every rank at the first non-zero
    exit) take rank 0 down before the headline line.  With a common deadline every rank that publishes in time is seen by all, so the lists agree."""
    import datetime
    store = store or _store(w)
    if deadline is None:
        deadline = clock() + max(5.0, patience_s)
    store.set(f"mvfx_side_stuck_{w.rank}", "1" if stuck else "0")
    bad = []
    for r in range(w.world):
        try:
            store.wait([f"mvfx_side_stuck_{r}"], datetime.timedelta(seconds=max(1.0, deadline - clock())))
            if store.get(f"mvfx_side_stuck_{r}") != b"0":
                bad.append(r)
        except Exception:  # noqa: BLE001  (timeout: the rank never arrived)
            bad.append(r)
    return bad


def mark_emitted(w, store=None):
    try:
        (store or _store(w)).set("mvfx_side_emitted", "1")
    except Exception:  # noqa: BLE001
        pass


def wait_emitted(w, seconds, store=None):
    """ranks other than 0 hold their non-zero exit until rank 0 has printed: a launcher ends every rank at the first failure"""
    import datetime
    try:
        (store or _store(w)).wait(["mvfx_side_emitted"], datetime.timedelta(seconds=seconds))
    except Exception:  # noqa: BLE001
        pass


def gst_pipeline_leg(args):
    """The real element, in a child process: `gst-launch-1.0 hiptestsrc refresh=false ! video/x-raw(memory:HIPMemory),RGBA,3840x2160 !
    hsvfilter ! fakesink` (tools/bench_gst_pipeline.py --honest 1) -- one launch per buffer, as shipped -- on a rotation of twelve 33 MB
    blocks (398 MB: larger than the 256 MB Infinity Cache, every buffer comes from HBM): `value`, `cache_resident: false`, and the only
    figure that gets a roofline fraction.  Beside it: the same pipeline on the pool's usual four blocks (cache resident, no fraction) and
    with a device consumer behind the filter (`! hsvdetector ! fakesink`).  The rate is taken inside the process by hiptestsrc between
    buffer N1 and the last one, at two instants where a recycled block is back with all downstream work on it finished.  `bound` is the
    headline kernel's (the committed counter passes): the same kernel runs."""
    gst_dir = os.path.join(ROOT, "gst-plugin-rs_amd", "gst-plugins")
    if not os.path.isdir(gst_dir) or not os.path.exists("/opt/conda/bin/gst-launch-1.0"):
        return {"error": "no GStreamer on this box (the element layer is an optional build target)"}
    import shutil
    import tempfile
    dump = tempfile.mkdtemp(prefix="mvfx_gst_verify_") if not NO_VERIFY[0] else ""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_gst_pipeline.py"), "--honest", "1", "--repeats", "2",
           "--n1", str(args.gst_n1), "--n2", str(args.gst_n2)] + (["--dump-dir", dump] if dump else [])
    verified = {"verified": None}
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("MVFX_ELEMENT_PAIR", "MVFX_HIP_POOL_MIN")}
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.gst_timeout, env=env)
        if r.returncode != 0:
            raise RuntimeError(f"bench_gst_pipeline rc {r.returncode}: {r.stderr[-200:]}")
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        if d.get("dump"):
            verified = verify_gst_dump(d.pop("dump"))
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        if dump:
            shutil.rmtree(dump, ignore_errors=True)
    fps = d.get("hsvfilter_hbm_resident_fps")
    committed = committed_counters("hsvfilter", 1)
    return {"metric": "gst_hsvfilter_element_4k_rgba_frames_per_sec", "value": fps, "unit": "frames/s",
            "ms_per_step": 1e3 / fps if fps else None,
            "roofline": {"bound": committed.get("bound"), "frac": fps * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS if fps else None},
            "sub_extra": {"pipeline": "gst-launch-1.0 hiptestsrc refresh=false ! video/x-raw(memory:HIPMemory),format=RGBA,3840x2160 ! hsvfilter ! fakesink, MVFX_HIP_POOL_MIN=12",
                          "cache_resident": False, "rotation_MB": round(12 * FRAME_BYTES / 1e6), "launch_model": "one launch per buffer (the default)",
                          "runs_fps": d.get("hsvfilter_hbm_resident_runs"), "buffers": [d.get("n1"), d.get("n2")],
                          "cache_resident_fps_4_blocks": d.get("hsvfilter_cache_resident_fps"),
                          "with_device_consumer_fps": d.get("hsvfilter_then_hsvdetector_hbm_resident_fps"),
                          "with_device_consumer_frac": d.get("hsvfilter_then_hsvdetector_frac_of_8TBs")},
            "detail": d, **verified}


def verify_gst_dump(dump):
    """three buffers out of `hiptestsrc ! (memory:HIPMemory) ! hsvfilter <bench settings> ! hipdownload` against the oracle on the source's frame"""
    try:
        import numpy as np
        from tests import oracle_binding as orc
        W, H = dump["width"], dump["height"]
        want = np.fromfile(dump["in"], dtype=np.uint8).reshape(H, W * 4).copy()
        orc.hsvfilter(want, W, W * 4, "RGBA", SETTINGS)
        got = np.fromfile(dump["out"], dtype=np.uint8)
        if got.size != dump["frames_out"] * want.size:
            raise AssertionError(f"the element pipeline wrote {got.size} bytes, expected {dump['frames_out']} frames of {want.size}")
        for k in range(dump["frames_out"]):
            _same(got[k * want.size:(k + 1) * want.size].reshape(H, W * 4), want, f"hsvfilter element, buffer {k}")
        return {"verified": True, "verified_what": "three buffers of the timed element (same caps and properties) downloaded and compared byte for byte with "
                                                   "oracle/hsv_oracle.c on the source's frame"}
    except Exception as e:  # noqa: BLE001
        return {"verified": False, "verified_what": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-seconds", type=float, default=0.6,
                    help="untimed run of the same step before the W warmup steps: the clock governor of the MI355X needs "
                         "~0.2 s of sustained load to leave its low-power state (profiles/r1/exp_ramp_launch_series.txt: "
                         "305 us/launch for the first 100 launches, 199 us after 0.2 s); 0 disables")
    ap.add_argument("--batch", type=int, default=16, help="4K frames (streams) per step per GPU")
    ap.add_argument("--pool", type=int, default=24, help="distinct batches resident in HBM (timed steps beyond it re-filter frames)")
    ap.add_argument("--launch-model", default="batch", choices=["batch", "streams"],
                    help="which launch model `value` reports: batch = --batch frames in one launch; streams = --stream-threads host "
                         "threads x own HIP stream x single-frame calls (the element's model); the other one is reported in config")
    ap.add_argument("--stream-threads", type=int, default=16, help="host threads of the streams model (0 = skip that leg)")
    ap.add_argument("--converged-data", type=int, default=0,
                    help="A/B only: filter every frame of the timed pool this many times BEFORE the timed steps (frames that have "
                         "been through hsvfilter repeatedly converge to low-entropy colours; the chip then draws less power and "
                         "clocks higher: profiles/r2/ab_fresh_vs_converged_data.txt). Default 0 = fresh uniform-random frames")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU baseline budget per leg (1 thread, then nproc threads)")
    ap.add_argument("--cpu-all-seconds", type=float, default=2.5, help="headline CPU baseline on nproc threads (0 = skip)")
    ap.add_argument("--other-cpu-all-seconds", type=float, default=0.0,
                    help="configs 2-5: CPU baseline on nproc threads (0 = skip: the driver's command has to fit in a minute; the sub-lines quote the "
                         "one-thread rate, which is what the reference's one streaming thread does)")
    ap.add_argument("--full", type=int, default=0, choices=[0, 1],
                    help="1: the final line is the whole document (profiling tools); 0: the compact line (<= 3000 bytes) -- the whole document "
                         "is always written to bench_out/last_run.json")
    ap.add_argument("--only-configs", default="", help="comma-separated subset of the configs 2-5 legs (default: all)")
    ap.add_argument("--noise-sweep", type=int, default=1, choices=[0, 1], help="colorlut: frames/s at +-0/3/5/8/16 codes of noise (sub-line field)")
    ap.add_argument("--gst-pipeline", type=int, default=1, choices=[0, 1],
                    help="N = 1: also time the real GStreamer element (gst-launch-1.0 hiptestsrc ! hsvfilter ! fakesink, 4K, child processes)")
    ap.add_argument("--gst-n1", type=int, default=10000)
    ap.add_argument("--gst-n2", type=int, default=110000,
                    help="the element sub-line: the rate is taken inside ONE gst-launch run of n2 buffers, between buffer n1 and the last one")
    ap.add_argument("--gst-timeout", type=float, default=150.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of one step of every leg (profiling runs)")
    ap.add_argument("--other-configs", type=int, default=1, choices=[0, 1],
                    help="hsvfilter workload: after the headline legs also measure BASELINE configs 2-5 in this run "
                         "(config.other_configs); N > 1: the band-sharded videocompare leg with its RCCL all-reduce")
    ap.add_argument("--combiner-legs", type=int, default=0, choices=[0, 1],
                    help="0 skips the two launch-combiner legs: rocprofiler-sdk 7.2 (rocprofv3 --kernel-trace / --pmc) crashes inside its HSA queue "
                         "interceptor on the cross-stream hipStreamWaitEvent traffic they generate (profiles/r3/rocprofv3_crash_in_queue_interceptor.txt)")
    ap.add_argument("--side-leg-timeout", type=float, default=180.0, help="N > 1: watchdog of the band-sharded videocompare leg, seconds")
    ap.add_argument("--other-cpu-seconds", type=float, default=0.7, help="CPU baseline budget per leg of configs 2-5 (1 thread, then nproc threads)")
    ap.add_argument("--other-settle-seconds", type=float, default=0.3, help="untimed run before each leg of configs 2-5")
    ap.add_argument("--pct-steps", type=int, default=200,
                    help="extra steps with a HIP event between every two, for the p10/p50/p90 of the per-step time (0 = skip)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 literal kernel, 2 strength-reduced")
    ap.add_argument("--streaming", type=int, default=1,
                    help="MVFX_OPT_NONTEMPORAL: 1 = the frames of this workload are not read again on the GPU (standalone filter): "
                         "write-through stores with the non-temporal hint (round 6, csrc/device_store.hpp), 0 = ordinary cached stores (element chains)")
    ap.add_argument("--frame-content", default="videotestsrc", choices=list(FRAME_CONTENTS),
                    help="hsvfilter workload: what the frames hold. videotestsrc = pattern=smpte frames exactly as GStreamer's "
                         "videotestsrc renders them (the buffers BASELINE.json's workload names); natural = smooth gradients + "
                         "noise; random = uniform-random bytes (the most power-hungry input: the chip clocks ~8 %% lower on it)")
    ap.add_argument("--content-sweep", type=int, default=1, choices=[0, 1],
                    help="hsvfilter workload: also time the batch leg on the other two frame contents (reported in config)")
    ap.add_argument("--content", default="natural", choices=["natural", "random", "smpte"],
                    help="colorlut workload: frame content. The LUT gathers are content dependent: smooth gradients with +-3 "
                         "codes of noise (default), uniform-random colours (worst case: every pixel another LUT cell), or flat "
                         "videotestsrc-smpte-like bars (best case)")
    ap.add_argument("--noise", type=int, default=3, help="colorlut workload, content natural: uniform noise of +-N codes on the gradients")
    ap.add_argument("--typed-loads", type=int, default=1, choices=[0, 1],
                    help="hsvfilter: u8/255 by typed buffer loads (texture-unit UNORM conversion) instead of VALU")
    ap.add_argument("--element-streams", type=int, default=1, choices=[1, 2],
                    help="videofx workload: 1 = roundedcorners and colordetect on one HIP stream (one streaming thread), 2 = on their "
                         "own streams (a queue between the elements)")
    ap.add_argument("--pairs-in-flight", type=int, default=2,
                    help="videocompare blockhash on one GPU: pairs whose host round trip overlaps the next pair's kernel (1 = the "
                         "synchronous mvfx_videocompare_distance call the element makes per aggregate)")
    ap.add_argument("--hash-algo", default="blockhash", choices=["blockhash", "dssim"],
                    help="videocompare workload: blockhash (the element's default) or the SSIM-family distance")
    ap.add_argument("--workload", default="hsvfilter",
                    choices=["hsvfilter", "hsv1080p", "colorlut", "videofx", "videocompare", "hsvfilter_rgb", "hsvdetector_rgb"],
                    help="hsvfilter = the headline metric (default, BASELINE metric); hsv1080p = config 2 "
                         "(hsvfilter + hsvdetector 1920x1080); colorlut = config 3 (33^3 cube, 4K); videofx = config 4 "
                         "(roundedcorners compose + colordetect, one 4K stream per GPU); videocompare = config 5")
    args = ap.parse_args()
    if args.no_cpu_baseline:
        args.other_cpu_seconds = 0.0
    NO_VERIFY[0] = bool(args.no_verify)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args, sys.argv[1:])  # the parent never touches the GPU
    if args.workload == "videocompare":
        return videocompare_main(args)
    if args.workload != "hsvfilter":
        return config_main(args)
    return hsvfilter_main(args)


if __name__ == "__main__":
    main()
