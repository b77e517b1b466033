#!/usr/bin/env python3
"""bench.py -- hsvfilter on 3840x2160 RGBA frames, device-resident, on N MI355X of one node.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (mvfx_hsvfilter_transform_frames_ip, in place) over one
batch of `--batch` synthetic 4K RGBA frames (one frame from each of `--batch` independent
streams) in ONE launch.  Frames are independent, so ranks shard streams with no data-path
collective ("weak" scaling: every GPU gets its own `--batch` streams); the only collectives
are the timing barrier and the max-over-ranks reduction.

Before the W warmup steps the same step runs untimed for --settle-seconds (0.6 s): the MI355X
clock governor starts every process in a low-power state and needs ~0.2 s of sustained load to
reach its steady clocks (305 us/launch for the first 100 launches, 199 us afterwards); a video
stream runs in the steady state, so that is what the K timed steps measure.

Inputs are resident in HBM before the timed region.  The frame pool is much larger than the
256 MiB Infinity Cache and every step touches a different batch, so reads come from HBM.

The JSON line carries `roofline` (algorithmic bytes per launch / average launch duration
measured with HIP events on the launch stream) and, at N=1, `cpu_baseline` (the oracle's
single-threaded loop -- what the reference does on its one streaming thread -- on a bounded
sample of the same frames).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
FRAME_BYTES = W4K * H4K * 4
SETTINGS = (90.0, 1.25, -0.05, 0.9, 0.02)  # SURVEY.md 8d hsvfilter settings
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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


def cpu_baseline(seconds: float):
    """Oracle (port of hsvfilter/imp.rs:76-120) on one host core, bounded sample."""
    import numpy as np
    from tests import frames
    from tests import oracle_binding as orc
    frame = frames.random_frame(0x5EED0001, W4K, H4K)
    work = frame.copy()
    orc.hsvfilter(work, W4K, W4K * 4, "RGBA", SETTINGS)  # warm
    n = 0
    dt = 0.0
    while dt < seconds:
        np.copyto(work, frame)  # fresh input each time; the copy is not timed
        t1 = time.perf_counter()
        orc.hsvfilter(work, W4K, W4K * 4, "RGBA", SETTINGS)
        dt += time.perf_counter() - t1
        n += 1
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of 3840x2160 RGBA (uniform random, seed 0x5EED0001), "
                      f"oracle/hsv_oracle.c gcc -O2 -ffp-contract=off, 1 thread, {dt:.1f} s"}


def videocompare_main(args):
    """BASELINE config 5: blockhash distance of 7680x4320 RGBA frame pairs.  Inputs are pre-sharded:
    rank r holds block-row band r of both frames; one all-reduce of 2x64 sums."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    from gst_plugin_rs_amd import distributed as D
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    vfx.check(lib.mvfx_set_device(local_rank))
    W, H = 7680, 4320
    r0, r1 = D.band_rows(H, rank, world)
    rows = r1 - r0
    pool = 4
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0001)  # same seed on every rank: band r of the same virtual frames
    pairs = torch.randint(0, 256, (pool, 2, rows * W * 4), dtype=torch.uint8, device=dev, generator=gen)
    sums = torch.zeros((2, 64), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)

    def bits(s, w, h):
        arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
        out = ctypes.c_uint64()
        vfx.check(lib.mvfx_blockhash_bits(arr, w, h, ctypes.byref(out)))
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

    settle(step, args.settle_seconds, lambda: torch.cuda.synchronize(dev), fixed_steps=400 if args.hash_algo == "blockhash" else 20)  # all-reduce inside the step
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        d = step(i)
    if args.hash_algo == "blockhash" and world == 1 and args.pairs_in_flight > 1:
        for slot in range(len(ring_busy)):  # drain the pairs still in flight (inside the timed region)
            if ring_busy[slot]:
                finish(slot)
        d = [last[0]]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    bytes_per_pair = 2 * W * H * 4
    achieved = bytes_per_pair * args.steps / elapsed / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "videocompare_8k_rgba_pairs_per_sec", "value": args.steps / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic uniform-random u8 RGBA, device-resident, rows pre-sharded by block-row band",
            "config": {"workload": "videocompare blockhash 7680x4320 RGBA pair, band-sharded + all-reduce(2x64 u32)" if args.hash_algo == "blockhash"
                       else "videocompare dssim (multi-scale SSIM, f64) 7680x4320 RGBA pair, row bands + 2 all-reduces of 10 f64",
                       "parallelism": f"{world} row bands, RCCL all-reduce per pair" if world > 1 else
                                      f"one GPU, whole frames, {args.pairs_in_flight} pair(s) in flight", "last_distance": d[0]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "end-to-end per pair incl. all-reduce (N > 1), D2H of the block sums, the synchronisation and host bit derivation"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def config_main(args):
    """BASELINE configs 2-4 as device-resident per-GPU stream workloads (no data-path collective)."""
    import torch
    import torch.distributed as dist
    import _pkg
    from tests import cubes
    vfx = _pkg.vfx
    lib = vfx.lib()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    vfx.check(lib.mvfx_set_device(local_rank))
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0100 + rank)

    def rnd(n, nbytes):
        return torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device=dev, generator=gen)

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
        else:  # smooth 2-D colour gradients (different phase per frame) + sensor-like noise of +-3 codes
            x = torch.linspace(0, 1, W, device=dev).view(1, 1, W)
            y = torch.linspace(0, 1, H, device=dev).view(1, H, 1)
            src = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
            for k in range(pool * nb):
                ph = 0.37 * k
                img = torch.stack([(0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)).expand(1, H, W),
                                   (0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)).expand(1, H, W),
                                   (0.5 + 0.45 * torch.cos(4.0 * x * y + ph)).expand(1, H, W),
                                   torch.ones((1, H, W), device=dev)], dim=-1) * 255.0
                noise = torch.randint(-3, 4, img.shape, device=dev, generator=gen).float()
                noise[..., 3] = 0
                src[k] = (img + noise).clamp(0, 255).to(torch.uint8).view(-1)
        dst = torch.empty((pool * nb, FRAME_BYTES), dtype=torch.uint8, device=dev)
        fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
        fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]

        def step(i):
            k = i % pool
            vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[k], fo[k], nb, sptr))
        frames_per_step = nb
        bytes_per_step, name = nb * 2 * FRAME_BYTES, (f"colorlut 33^3 .cube, {nb} streams of 3840x2160 RGBA per launch, content={args.content} "
                                                      "(the LUT gathers are content dependent: random colours are the worst case, flat bars the best)")
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

        def step(i):
            k = i % pool
            vfx.check(lib.mvfx_roundedcorners_compose_a420(ctypes.byref(planes[k][0]), ctypes.c_void_p(mask.data_ptr()), W,
                                                           ctypes.byref(planes[k][1]), sptr))
            vfx.check(lib.mvfx_colordetect_histogram(ctypes.byref(fr[k]), 10, 0, vfx.ALL_SAMPLES, ctypes.c_void_p(hist.data_ptr()),
                                                     ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr))
        frames_per_step = 1
        bytes_per_step, name = W * H * 4 + FRAME_BYTES, "roundedcorners I420->A420 compose (r=100) + colordetect histogram (quality=10), one 3840x2160 stream per GPU"

    settle(step, args.settle_seconds, lambda: torch.cuda.synchronize(dev))
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    achieved = bytes_per_step * args.steps / elapsed / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": f"{args.workload}_frames_per_sec", "value": args.steps * frames_per_step * world / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.workload != "videofx" else "u8",
            "data": "synthetic uniform-random u8, device-resident", "config": {"workload": name, "parallelism": f"{world} independent streams"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "note": "wall clock over the launches of a step (per GPU)"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-seconds", type=float, default=0.6,
                    help="untimed run of the same step before the W warmup steps: the clock governor of the MI355X needs "
                         "~0.2 s of sustained load to leave its low-power state (profiles/r1/exp_ramp_launch_series.txt: "
                         "305 us/launch for the first 100 launches, 199 us after 0.2 s); 0 disables")
    ap.add_argument("--batch", type=int, default=16, help="4K frames (streams) per step per GPU")
    ap.add_argument("--pool", type=int, default=24, help="distinct batches resident in HBM")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 literal kernel, 2 strength-reduced")
    ap.add_argument("--streaming", type=int, default=1,
                    help="mvfx_hsvfilter_set_streaming: 1 = non-temporal loads/stores (the frames of this workload are not "
                         "read again on the GPU: standalone filter), 0 = normal caching (element chains)")
    ap.add_argument("--content", default="natural", choices=["natural", "random", "smpte"],
                    help="colorlut workload: frame content. The LUT gathers are content dependent: smooth gradients with +-3 "
                         "codes of noise (default), uniform-random colours (worst case: every pixel another LUT cell), or flat "
                         "videotestsrc-smpte-like bars (best case)")
    ap.add_argument("--typed-loads", type=int, default=1, choices=[0, 1],
                    help="hsvfilter: u8/255 by typed buffer loads (texture-unit UNORM conversion) instead of VALU")
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
    if args.workload == "videocompare":
        return videocompare_main(args)
    if args.workload != "hsvfilter":
        return config_main(args)

    import torch
    import torch.distributed as dist
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()  # raises if libmi355vfx.so is missing: no fallback

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                             "torch.distributed.run --nproc-per-node N\n")
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the HIP path has no CPU fallback\n")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    vfx.check(lib.mvfx_set_device(local_rank))
    vfx.check(lib.mvfx_hsvfilter_set_variant(args.variant))
    vfx.check(lib.mvfx_hsvfilter_set_streaming(args.streaming))
    vfx.check(lib.mvfx_hsvfilter_set_typed_loads(args.typed_loads))

    # ---- resident frame pool: pool x batch distinct uniform-random 4K RGBA frames -------------
    pool = max(1, args.pool)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0100 + rank)
    frames = torch.randint(0, 256, (pool, args.batch, FRAME_BYTES), dtype=torch.uint8, device=dev,
                           generator=gen)
    settings = vfx.HsvFilterSettings(*SETTINGS)
    frame_arrays = []
    for b in range(pool):
        arr = (vfx.Frame * args.batch)(*[
            vfx.make_frame(frames[b, i].data_ptr(), W4K, H4K, W4K * 4, "RGBA") for i in range(args.batch)])
        frame_arrays.append(arr)
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)

    def step(i):
        rc = lib.mvfx_hsvfilter_transform_frames_ip(frame_arrays[i % pool], args.batch,
                                                    ctypes.byref(settings), sptr)
        if rc != 0:
            raise RuntimeError(f"mvfx status {rc}: {vfx.last_error()}")

    def barrier():
        if world > 1:
            dist.barrier()

    # clock ramp, see --settle-seconds; not part of the W warmup / K timed steps
    settle_steps = settle(step, args.settle_seconds, lambda: torch.cuda.synchronize(dev))
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # average launch duration on the stream

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # HBM traffic per launch from the committed rocprofv3 PMC passes (cannot be collected live)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hsvfilter_traffic.json")) as f:
            t = json.load(f)
        traffic = t["hbm_bytes_per_launch"] * args.batch / t["frames_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    total_frames = args.steps * args.batch * world
    fps = total_frames / elapsed
    bytes_per_launch = args.batch * 2 * FRAME_BYTES  # 4 B read + 4 B written per pixel
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "hsvfilter_4k_rgba_frames_per_sec",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic uniform-random u8 RGBA (torch.randint, seed 0x5EED0100+rank), device-resident",
        "config": {"workload": "hsvfilter 3840x2160 RGBA in place, hue-shift=90 saturation-mul=1.25 "
                               "saturation-off=-0.05 value-mul=0.9 value-off=0.02",
                   "frames_per_step_per_gpu": args.batch, "resident_batches": pool,
                   "settle_seconds_before_warmup": args.settle_seconds, "settle_steps": settle_steps,
                   "parallelism": f"{world} independent stream shards, no data-path collective",
                   "kernel_variant": {0: "auto", 1: "literal", 2: "strength-reduced"}[args.variant],
                   "cache_policy": "non-temporal (mvfx_hsvfilter_set_streaming(1))" if args.streaming else "default",
                   "u8_to_unit_float": "typed buffer loads (texture-unit UNORM8, exact)" if args.typed_loads else "VALU (cvt + mul + fmac)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "hsvfilter4_typed_kernel" if args.typed_loads else "hsvfilter4_kernel<RGBA, vec4>", "bytes_per_launch": bytes_per_launch,
                     "avg_launch_ms": kernel_ms, "read_side_GBs": achieved / 2},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
