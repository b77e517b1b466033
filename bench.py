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
launch stream, plus the d2d-copy ceiling measured in the same run) and, at N=1, `cpu_baseline` (the oracle's loop on
one host thread -- what the reference does on its one streaming thread -- and on nproc threads, bounded sample).
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
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.rccl_ranks = 1
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="nccl", device_id=self.dev)  # "nccl" IS RCCL on ROCm
            one = torch.ones(1, dtype=torch.int32, device=self.dev)
            dist.all_reduce(one)  # an actual collective: how many ranks RCCL sees
            self.rccl_ranks = int(one[0])
        self.vfx.check(self.lib.mvfx_set_device(self.local_rank))
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
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    def gather(self, val):
        if self.world == 1:
            return [val]
        t = self.torch.tensor([val], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(x[0]) for x in out]

    def timed(self, step, steps, first_index=0, events=False):
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
            ev1.record(self.stream)
        self.sync()
        self.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        return elapsed, (ev0.elapsed_time(ev1) / max(steps, 1) if events else None)

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


def measured_copy_ceiling(w):
    """Device-to-device copy of 1 GiB (2 GiB of HBM traffic) with torch, same run, same clocks: the practical HBM ceiling
    SURVEY 8d asks to quote beside the 8 TB/s spec."""
    torch = w.torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=w.dev)
    b = torch.empty(n, dtype=torch.uint8, device=w.dev)
    a.random_(0, 256)
    for _ in range(5):
        b.copy_(a)
    w.sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record(w.stream)
    for _ in range(reps):
        b.copy_(a)
    ev1.record(w.stream)
    w.sync()
    return 2.0 * n * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


# ------------------------------------------------------------------------------------------------ CPU baseline

# ------------------------------------------------------------------------------------------------ frame contents

FRAME_CONTENTS = ("videotestsrc", "natural", "random")
CONTENT_TEXT = {
    "videotestsrc": "videotestsrc pattern=smpte RGBA frames (colour bars + LCG snow, byte-identical to GStreamer's generator: "
                    "tests/test_videotestsrc_frames_cpu.py; consecutive frames of the stream, i.e. distinct snow in each)",
    "natural": "smooth colour gradients (another phase per frame) + uniform noise of +-3 codes (natural-like)",
    "random": "uniform-random u8 RGBA (torch.randint)",
}


def natural_frame(torch, dev, gen, k, W, H):
    """One natural-like RGBA frame as a flat u8 tensor: smooth 2-D colour gradients (phase k) + noise of +-3 codes."""
    x = torch.linspace(0, 1, W, device=dev).view(1, W)
    y = torch.linspace(0, 1, H, device=dev).view(H, 1)
    ph = 0.37 * k
    img = torch.stack([(0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)).expand(H, W),
                       (0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)).expand(H, W),
                       (0.5 + 0.45 * torch.cos(4.0 * x * y + ph)).expand(H, W),
                       torch.ones((H, W), device=dev)], dim=-1) * 255.0
    noise = torch.randint(-3, 4, img.shape, device=dev, generator=gen).float()
    noise[..., 3] = 0
    return (img + noise).clamp(0, 255).to(torch.uint8).view(-1)


def fill_frames(torch, dev, gen, flat, kind, W, H, first_frame=0):
    """Fill flat[n, W*H*4] (device, u8) with `kind` frames; frame j is frame first_frame + j of its stream."""
    n = flat.shape[0]
    if kind == "random":
        flat.random_(0, 256, generator=gen)
    elif kind == "natural":
        for j in range(n):
            flat[j] = natural_frame(torch, dev, gen, first_frame + j, W, H)
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


def cpu_baseline(seconds: float, content: str = "videotestsrc"):
    """Oracle (port of hsvfilter/imp.rs:76-120, gcc -O3 -ffp-contract=off like profile.release) on the GPU box's host cores:
    one thread -- what the reference does, its transform_frame_ip runs on ONE streaming thread per element -- and
    nproc threads each filtering its own stream.  8 distinct frames of the same content as the GPU legs rotate so
    the 33 MB input is not cache resident.  Bounded: ~seconds per leg."""
    import threading
    import numpy as np
    from tests import frames
    from tests import oracle_binding as orc
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    distinct = 8
    if content == "videotestsrc":
        vts, _ = frames.videotestsrc_smpte(W4K, H4K, distinct)
        src = [vts[k] for k in range(distinct)]
        what = "videotestsrc pattern=smpte, 8 consecutive frames rotating"
    else:
        src = [frames.random_frame(0x5EED0001 + k, W4K, H4K) for k in range(distinct)]
        what = "uniform random, 8 distinct frames rotating, seeds 0x5EED0001..8"

    def run(n_threads, budget):
        counts = [0] * n_threads
        spans = [0.0] * n_threads
        go = threading.Event()

        def body(t):
            work = [s.copy() for s in src[:2]]  # this thread's private pair of destination buffers
            orc.hsvfilter(work[0], W4K, W4K * 4, "RGBA", SETTINGS)  # warm (page faults, code)
            go.wait()
            n, dt = 0, 0.0
            while dt < budget:
                buf = work[n & 1]
                np.copyto(buf, src[(t + n) % distinct])  # fresh input each time; the copy is not timed
                t1 = time.perf_counter()
                orc.hsvfilter(buf, W4K, W4K * 4, "RGBA", SETTINGS)  # ctypes releases the GIL for the call
                dt += time.perf_counter() - t1
                n += 1
            counts[t], spans[t] = n, dt

        ths = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
        for th in ths:
            th.start()
        go.set()
        for th in ths:
            th.join()
        return sum(c / s for c, s in zip(counts, spans)), sum(counts), max(spans)

    one_fps, one_n, one_dt = run(1, seconds)
    all_fps, all_n, all_dt = run(nproc, seconds) if nproc > 1 else (one_fps, one_n, one_dt)
    return {"value": one_fps, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{one_n} frames of 3840x2160 RGBA ({what}), "
                      f"oracle/hsv_oracle.c gcc -O3 -ffp-contract=off, 1 thread, {one_dt:.1f} s",
            "all_cores": {"value": all_fps, "unit": "frames/s", "cores": nproc, "nproc": nproc,
                          "sample": f"{all_n} frames, {nproc} threads x own stream of frames, {all_dt:.1f} s per thread"}}


# ------------------------------------------------------------------------------------------------ config 5

def videocompare_main(args):
    """BASELINE config 5: blockhash distance of 7680x4320 RGBA frame pairs.  Inputs are pre-sharded:
    rank r holds block-row band r of both frames; one all-reduce of 2x64 sums."""
    w = Worker(args)
    torch, vfx, lib, dev, sptr, stream = w.torch, w.vfx, w.lib, w.dev, w.sptr, w.stream
    from gst_plugin_rs_amd import distributed as D
    rank, world = w.rank, w.world
    W, H = 7680, 4320
    r0, r1 = D.band_rows(H, rank, world)
    rows = r1 - r0
    pool = 4
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0001)  # same seed on every rank: band r of the same virtual frames
    pairs = torch.randint(0, 256, (pool, 2, rows * W * 4), dtype=torch.uint8, device=dev, generator=gen)
    sums = torch.zeros((2, 64), dtype=torch.int32, device=dev)

    def bits(s, w_, h_):
        arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
        out = ctypes.c_uint64()
        vfx.check(lib.mvfx_blockhash_bits(arr, w_, h_, ctypes.byref(out)))
        return out.value

    if args.hash_algo == "dssim":
        # every rank holds both full frames (a band's 5-level pyramid needs a halo of up to 64 rows) and maps only its band;
        # two all-reduces of 10 f64 per pair (distributed.ssim_sharded)
        del pairs
        pool = 2
        gen.manual_seed(0x5EED0002)
        full = torch.randint(0, 256, (pool, 2, H * W * 4), dtype=torch.uint8, device=dev, generator=gen)
        full[:, 1] = full[:, 0]
        full[:, 1, ::97] ^= 0x10
        y0, y1 = D.ssim_band_rows(H, rank, world)
        fr = [[vfx.make_frame(full[k, p].data_ptr(), W, H, W * 4, "RGBA") for p in range(2)] for k in range(pool)]

        def step(i):
            k = i % pool
            return [D.ssim_sharded(lambda: vfx.ssim_partial_sums(fr[k][0], fr[k][1], y0, y1, sptr),
                                   lambda mean: vfx.ssim_partial_deviation(mean, sptr), vfx.ssim_combine, dev)]
    else:
        bands = [(vfx.Frame * 2)(*[vfx.make_frame(pairs[k, p].data_ptr(), W, rows, W * 4, "RGBA") for p in range(2)])
                 for k in range(pool)]

        dist_out = ctypes.c_double()
        depth = max(1, args.pairs_in_flight)
        ring_dev = torch.zeros((depth, 2, 64), dtype=torch.int32, device=dev)
        ring_host = torch.zeros((depth, 2, 64), dtype=torch.int32).pin_memory()
        ring_evt = [torch.cuda.Event() for _ in range(depth)]
        ring_busy = [False] * depth
        last = [0.0]

        def finish(slot):
            ring_evt[slot].synchronize()
            h = [bits([int(v) & 0xFFFFFFFF for v in ring_host[slot, p].tolist()], W, H) for p in range(2)]
            ring_busy[slot] = False
            last[0] = float(bin(h[0] ^ h[1]).count("1"))

        def step(i):
            if world == 1 and depth == 1:
                # one GPU holds both whole frames: HasherEngine::hash_image x2 + compare in the C ABI (one launch for both
                # pads, one 512-byte D2H, host bit derivation), exactly what the element does per aggregate
                vfx.check(lib.mvfx_videocompare_distance(ctypes.byref(bands[i % pool][0]), ctypes.byref(bands[i % pool][1]),
                                                         ctypes.byref(dist_out), sptr))
                return [dist_out.value]
            if world == 1:
                # `depth` pairs in flight: the block sums of pair i are copied to pinned host memory asynchronously and
                # turned into hashes / the distance while the kernel of pair i+1 .. i+depth-1 runs (the host round trip of
                # a pair, ~25 us, no longer sits between two 42 us kernels)
                slot = i % depth
                if ring_busy[slot]:
                    finish(slot)
                vfx.check(lib.mvfx_blockhash_sums_pads(bands[i % pool], 2, H, 0, ctypes.c_void_p(ring_dev[slot].data_ptr()), sptr))
                ring_host[slot].copy_(ring_dev[slot], non_blocking=True)
                ring_evt[slot].record(stream)
                ring_busy[slot] = True
                return [last[0]]

            def partial():  # both pads' bands in one launch
                vfx.check(lib.mvfx_blockhash_sums_pads(bands[i % pool], 2, H, r0, ctypes.c_void_p(sums.data_ptr()), sptr))
                return sums
            return D.videocompare_sharded(partial, 2, W, H, bits, dev, all_pads=True)

    settle(step, args.settle_seconds, w.sync, fixed_steps=400 if args.hash_algo == "blockhash" else 20)  # all-reduce inside the step
    for i in range(args.warmup):
        step(i)
    result = [None]

    def timed_step(i):
        result[0] = step(i)

    def whole(i):  # the K steps + the drain of the pairs still in flight, all inside the timed region
        for k in range(args.steps):
            timed_step(k)
        if args.hash_algo == "blockhash" and world == 1 and args.pairs_in_flight > 1:
            for slot in range(len(ring_busy)):
                if ring_busy[slot]:
                    finish(slot)
            result[0] = [last[0]]

    elapsed, _ = w.timed(whole, 1)
    (elapsed,) = w.max_over_ranks(elapsed)
    d = result[0]
    per_rank = w.gather(args.steps / elapsed)
    bytes_per_pair = 2 * W * H * 4
    achieved = bytes_per_pair * args.steps / elapsed / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "videocompare_8k_rgba_pairs_per_sec", "value": args.steps / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32" if args.hash_algo == "blockhash" else "f64",
            "data": "synthetic uniform-random u8 RGBA, device-resident, rows pre-sharded by block-row band",
            "config": {"workload": "videocompare blockhash 7680x4320 RGBA pair, band-sharded + all-reduce(2x64 u32)" if args.hash_algo == "blockhash"
                       else "videocompare dssim (multi-scale SSIM, f64) 7680x4320 RGBA pair, row bands + 2 all-reduces of 10 f64",
                       "parallelism": f"{world} row bands, RCCL all-reduce per pair" if world > 1 else
                                      f"one GPU, whole frames, {args.pairs_in_flight} pair(s) in flight", "last_distance": d[0],
                       "rccl_ranks": w.rccl_ranks, "per_rank_pairs_per_sec": per_rank},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "end-to-end per pair incl. all-reduce (N > 1), D2H of the block sums, the synchronisation and host bit derivation"
                                 if args.hash_algo == "blockhash" else
                                 "the compulsory input bytes against HBM peak (SURVEY 8d); this path is f64 arithmetic: its five kernels issue "
                                 "4.74e8 wave64 VALU instructions per 8K pair (rocprofv3 SQ_INSTS_VALU, profiles/r2/ssim_counters_after.txt), see "
                                 "valu_issue_frac",
                         **({} if args.hash_algo == "blockhash" else
                            {"valu_issue_frac": 4.74e8 * 64 * (args.steps / elapsed) / world / (256 * 64 * 2.4e9),
                             "valu_issue_note": "wave64 VALU instructions per second / (256 CUs x 64 lanes x 2.4 GHz), single GPU whole frames"})}}),
              flush=True)
    w.finish()


# ------------------------------------------------------------------------------------------------ configs 2-4

def config_main(args):
    """BASELINE configs 2-4 as device-resident per-GPU stream workloads (no data-path collective)."""
    from tests import cubes
    w = Worker(args)
    torch, vfx, lib, dev, sptr = w.torch, w.vfx, w.lib, w.dev, w.sptr
    rank, world = w.rank, w.world
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0100 + rank)

    def rnd(n, nbytes):
        return torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device=dev, generator=gen)

    data = "synthetic uniform-random u8, device-resident"
    if args.workload == "hsv1080p":
        # args.batch independent 1080p streams per step: one frame of each through hsvfilter then hsvdetector,
        # two launches per step (a single 1080p frame is ~3 + ~6 us of GPU work: launch-bound one at a time)
        W, H, nb = 1920, 1080, args.batch
        pool = max(2, 96 // nb)
        src, dst = rnd(pool * nb, W * H * 4), torch.empty((pool * nb, W * H * 4), dtype=torch.uint8, device=dev)
        fs = vfx.HsvFilterSettings(*SETTINGS)
        ds = vfx.HsvDetectorSettings(120.0, 40.0, 0.6, 0.4, 0.6, 0.4)
        fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBx") for i in range(nb)]) for b in range(pool)]
        fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]

        def step(i):
            k = i % pool
            vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(fi[k], nb, ctypes.byref(fs), sptr))
            vfx.check(lib.mvfx_hsvdetector_transform_frames(fi[k], fo[k], nb, ctypes.byref(ds), sptr))
        frames_per_step = nb
        bytes_per_step, name = nb * 4 * W * H * 4, f"hsvfilter (RGBx, in place) + hsvdetector RGBx->RGBA, {nb} streams of 1920x1080 per launch"
    elif args.workload == "colorlut":
        # args.batch streams graded with the same 33^3 LUT, one frame of each per launch
        W, H, nb = W4K, H4K, args.batch
        pool = max(2, 32 // nb)
        lut = vfx.CubeLut(cubes.analytic_3d(33))
        if args.content == "random":
            src = rnd(pool * nb, FRAME_BYTES)
        elif args.content == "smpte":
            from tests import frames as _frames
            one = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
            src = one.unsqueeze(0).repeat(pool * nb, 1).contiguous()
            data = "synthetic videotestsrc-smpte-like bars, device-resident"
        else:  # smooth 2-D colour gradients (different phase per frame) + sensor-like noise of +-3 codes
            data = "synthetic smooth colour gradients + uniform noise of +-3 codes (natural-like), device-resident"
            src = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
            for k in range(pool * nb):
                src[k] = natural_frame(torch, dev, gen, k, W, H)
        dst = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
        fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
        fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]

        def step(i):
            k = i % pool
            vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[k], fo[k], nb, sptr))
        frames_per_step = nb
        bytes_per_step, name = nb * 2 * FRAME_BYTES, (f"colorlut 33^3 .cube, {nb} streams of 3840x2160 RGBA per launch, content={args.content} "
                                                      "(the LUT gathers are content dependent: random colours are the worst case, flat bars the best)")

        def streams_leg():
            """the element's launch model: --stream-threads host threads x own HIP stream x single-frame mvfx_colorlut_transform_frame"""
            hb = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
            nthr = args.stream_threads
            fpt = max(1, (pool * nb) // nthr)
            fin = (vfx.Frame * (nthr * fpt))(*[vfx.make_frame(src[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nthr * fpt)])
            fout = (vfx.Frame * (nthr * fpt))(*[vfx.make_frame(dst[k].data_ptr(), W, H, W * 4, "RGBA") for k in range(nthr * fpt)])
            launches, reps = max(100, args.steps * nb // nthr), 5
            secs, per = (ctypes.c_double * reps)(), (ctypes.c_double * nthr)()
            w.sync()
            w.barrier()
            rc = hb.mvfxbench_colorlut_streams(w.local_rank, nthr, 400, launches, reps, lut.h, fin, fout, fpt, 0, secs, per)
            if rc != 0:
                raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
            w.barrier()
            (med,) = w.max_over_ranks(sorted(secs)[reps // 2])
            fps = nthr * launches * world / med
            return {"launch_model": f"{nthr} threads x 1 frame (own HIP stream each, single-frame mvfx_colorlut_transform_frame)",
                    "value": fps, "unit": "frames/s", "launches_per_thread": launches, "statistic": "median of 5 repetitions",
                    "frac": fps / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS}
    else:  # videofx: one 4K stream per GPU: I420 -> A420 compose with the r=100 mask + colordetect on the RGBA twin
        W, H, pool = W4K, H4K, 16
        i420, a420 = rnd(pool, W * H * 3 // 2), torch.empty((pool, W * H * 5 // 2), dtype=torch.uint8, device=dev)
        rgba = rnd(pool, FRAME_BYTES)
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

        # --element-streams 2: the two elements on their own HIP streams, as with a `queue` between them (two streaming threads:
        # frame k's colordetect overlaps frame k+1's compose); 1: both on one stream, one after the other (one streaming thread)
        second = torch.cuda.Stream(device=dev) if args.element_streams == 2 else None
        sptr2 = ctypes.c_void_p(second.cuda_stream) if second is not None else sptr

        def step(i):
            k = i % pool
            vfx.check(lib.mvfx_roundedcorners_compose_a420(ctypes.byref(planes[k][0]), ctypes.c_void_p(mask.data_ptr()), W,
                                                           ctypes.byref(planes[k][1]), sptr))
            vfx.check(lib.mvfx_colordetect_histogram(ctypes.byref(fr[k]), 10, 0, vfx.ALL_SAMPLES, ctypes.c_void_p(hist.data_ptr()),
                                                     ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr2))
        frames_per_step = 1
        bytes_per_step = W * H * 4 + FRAME_BYTES
        name = ("roundedcorners I420->A420 compose (r=100) + colordetect histogram (quality=10), one 3840x2160 stream per GPU, "
                + ("both elements on one HIP stream (one streaming thread)" if second is None else
                   "the two elements on their own HIP streams (a queue between them: two streaming threads)"))

    settle(step, args.settle_seconds, w.sync)
    for i in range(args.warmup):
        step(i)
    elapsed, _ = w.timed(step, args.steps, first_index=args.warmup)
    (elapsed,) = w.max_over_ranks(elapsed)
    per_rank = w.gather(args.steps * frames_per_step / elapsed)
    achieved = bytes_per_step * args.steps / elapsed / 1e9
    other_model = streams_leg() if args.workload == "colorlut" and args.stream_threads > 0 else None
    if rank == 0:
        print(json.dumps({
            "metric": f"{args.workload}_frames_per_sec", "value": args.steps * frames_per_step * world / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.workload != "videofx" else "u8",
            "data": data, "config": {"workload": name, "parallelism": f"{world} independent streams", "rccl_ranks": w.rccl_ranks,
                                     "per_rank_frames_per_sec": per_rank, **({"other_launch_model": other_model} if other_model else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "note": "wall clock over the launches of a step (per GPU)"}}), flush=True)
    w.finish()


# ------------------------------------------------------------------------------------------------ headline

def hsvfilter_main(args):
    w = Worker(args)
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

    def launch(batch_index):
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

    settle_steps, elapsed, kernel_ms = batch_leg()
    batch_fps_rank = w.gather(args.steps * args.batch / elapsed)
    batch_fps = args.steps * args.batch * world / elapsed

    # ---- the element's launch model: --batch host threads x own HIP stream x single-frame calls -----------
    streams = None
    bench_so = os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so")
    if args.stream_threads > 0:
        hb = ctypes.CDLL(bench_so)  # raises when the harness was not built (build() builds it)
        nthr = args.stream_threads
        fpt = max(2, (pool * args.batch) // nthr)          # frames per thread, all from the resident pool
        flat = (vfx.Frame * (nthr * fpt))(*[
            vfx.make_frame(frames[(k // args.batch) % pool, k % args.batch].data_ptr(), W4K, H4K, W4K * 4, "RGBA")
            for k in range(nthr * fpt)])
        # at least 200 launches per thread: the K x batch frames of the batch leg (320 at the driver's K=20) would be ~20
        # launches per thread = 5 ms, dominated by thread wake-up skew
        launches = max(200, args.steps * args.batch // nthr)
        # the threads create their streams first (GPU idle for several ms -> clocks drop): own ~0.4 s warm-up on the threads
        stream_warmup = max(20, args.warmup, int(args.settle_seconds / 0.6 * 28000) // nthr)
        reps = 5  # the median of five back-to-back repetitions: a single 40 ms window is at the mercy of one descheduled thread
        secs = (ctypes.c_double * reps)()
        per = (ctypes.c_double * nthr)()
        w.sync()
        w.barrier()
        rc = hb.mvfxbench_hsvfilter_streams(w.local_rank, nthr, stream_warmup, launches, reps, flat, fpt, ctypes.byref(settings),
                                            opts, secs, per)
        if rc != 0:
            raise RuntimeError(f"mvfxbench status {rc}: {vfx.last_error()}")
        w.barrier()
        rep_secs = sorted(secs)
        (s_elapsed,) = w.max_over_ranks(rep_secs[reps // 2])
        s_fps = nthr * launches * world / s_elapsed
        streams = {"launch_model": f"{nthr} threads x 1 frame (own HIP stream each, single-frame mvfx_hsvfilter_transform_frame_ip, "
                                   "no sync between launches)",
                   "value": s_fps, "unit": "frames/s", "frames": nthr * launches, "launches_per_thread": launches,
                   "warmup_launches_per_thread": stream_warmup, "seconds": s_elapsed, "statistic": "median of 5 repetitions",
                   "achieved_GBs": s_fps / world * 2 * FRAME_BYTES / 1e9, "frac": s_fps / world * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS,
                   "repetitions_frames_per_sec": [round(nthr * launches / t) for t in secs],
                   "per_rank_frames_per_sec": w.gather(nthr * launches / rep_secs[reps // 2])}

    # measured after the timed legs: a burst of plain copies between settle and the timed steps leaves the governor in another
    # power state (the timed kernels then ran 3-4 % slower: profiles/r2/ab_fresh_vs_converged_data.txt)
    ceiling = measured_copy_ceiling(w)
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
                           "frac": args.batch * 2 * FRAME_BYTES / (sw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "data": CONTENT_TEXT[kind]}
    # HBM traffic per launch from the committed rocprofv3 PMC passes (cannot be collected live)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hsvfilter_traffic.json")) as f:
            t = json.load(f)
        traffic = t["hbm_bytes_per_launch"] * args.batch / t["frames_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    bytes_per_launch = args.batch * 2 * FRAME_BYTES  # 4 B read + 4 B written per pixel
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    batch_model = {"launch_model": f"1 launch x {args.batch} frames (mvfx_hsvfilter_transform_frames_ip, blockIdx.z = stream)",
                   "value": batch_fps, "unit": "frames/s", "per_rank_frames_per_sec": batch_fps_rank}
    use_streams = args.launch_model == "streams" and streams is not None
    head, other = (streams, batch_model) if use_streams else (batch_model, streams)
    total_frames = args.steps * args.batch * world
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
        "data": (f"synthetic {CONTENT_TEXT[args.frame_content]}, device-resident; the timed steps start on frames no kernel has "
                 "touched (settle + warm-up run on scratch batches)") if not args.converged_data else
                f"A/B: {args.frame_content} frames filtered {args.converged_data}x before the timed steps (converged, low-entropy)",
        "config": {"workload": "hsvfilter 3840x2160 RGBA in place, hue-shift=90 saturation-mul=1.25 "
                               "saturation-off=-0.05 value-mul=0.9 value-off=0.02",
                   "frame_content": args.frame_content, "other_frame_contents": sweep,
                   "launch_model": head["launch_model"], "other_launch_model": other,
                   "frames_per_step_per_gpu": args.batch, "resident_batches": pool,
                   "settle_seconds_before_warmup": args.settle_seconds, "settle_steps": settle_steps,
                   "parallelism": f"{world} independent stream shards, no data-path collective",
                   "rccl_ranks": w.rccl_ranks, "per_rank_frames_per_sec": head["per_rank_frames_per_sec"],
                   "kernel_variant": {0: "auto", 1: "literal", 2: "strength-reduced"}[args.variant],
                   "cache_policy": "non-temporal (MVFX_OPT_NONTEMPORAL)" if args.streaming else "default",
                   "u8_to_unit_float": "typed buffer loads (texture-unit UNORM8, exact)" if args.typed_loads else "VALU (cvt + mul + fmac)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "hsvfilter4_typed_kernel" if args.typed_loads else "hsvfilter4_kernel<RGBA, vec4>", "bytes_per_launch": bytes_per_launch,
                     "avg_launch_ms": kernel_ms, "read_side_GBs": achieved / 2,
                     "ceiling_measured_GBs": ceiling, "frac_of_measured_ceiling": achieved / ceiling if ceiling else None,
                     "ceiling_note": "torch device-to-device copy of 1 GiB (read + write bytes) timed with HIP events in this run",
                     "launch_model": batch_model["launch_model"]},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.frame_content)
    if rank == 0:
        print(json.dumps(out), flush=True)
    w.finish()


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
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline budget per leg (1 thread, then nproc threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 literal kernel, 2 strength-reduced")
    ap.add_argument("--streaming", type=int, default=1,
                    help="MVFX_OPT_NONTEMPORAL: 1 = non-temporal loads/stores (the frames of this workload are not "
                         "read again on the GPU: standalone filter), 0 = normal caching (element chains)")
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
                    choices=["hsvfilter", "hsv1080p", "colorlut", "videofx", "videocompare"],
                    help="hsvfilter = the headline metric (default, BASELINE metric); hsv1080p = config 2 "
                         "(hsvfilter + hsvdetector 1920x1080); colorlut = config 3 (33^3 cube, 4K); videofx = config 4 "
                         "(roundedcorners compose + colordetect, one 4K stream per GPU); videocompare = config 5")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args, sys.argv[1:])  # the parent never touches the GPU
    if args.workload == "videocompare":
        return videocompare_main(args)
    if args.workload != "hsvfilter":
        return config_main(args)
    return hsvfilter_main(args)


if __name__ == "__main__":
    main()
