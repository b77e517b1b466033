#!/usr/bin/env python3
"""Per-stream launch model: N host threads x own HIP stream x single-frame mvfx_hsvfilter_transform_frame_ip
(libmvfxbench.so).  Prints aggregate frames/s and the fraction of 8 TB/s for N = 1, 2, 4, 8, 16, 32.

    python tools/bench_streams.py [--frames-per-thread 6] [--launches 400] [--nt 0|1]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames-per-thread", type=int, default=6)
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--threads", default="1,2,4,8,16,32")
    ap.add_argument("--nt", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H = args.width, args.height
    fb = W * H * 4
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    opts = vfx.OPT_NONTEMPORAL if args.nt else 0
    for n in [int(x) for x in args.threads.split(",")]:
        fpt = max(2, min(args.frames_per_thread, 96 // n)) if n * args.frames_per_thread > 96 else args.frames_per_thread
        pool = torch.randint(0, 256, (n * fpt, fb), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        frames = (vfx.Frame * (n * fpt))(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(n * fpt)])
        secs = (ctypes.c_double * 3)()
        per = (ctypes.c_double * n)()
        # settle the clocks with an untimed run, then the timed one
        for launches in (max(200, 3000 // n), max(50, args.launches)):
            rc = bench.mvfxbench_hsvfilter_streams(0, n, 20, launches, 3, frames, fpt, ctypes.byref(settings), opts, secs, per)
            assert rc == 0, (rc, vfx.last_error())
        fps = n * launches / sorted(secs)[1]
        print(f"threads {n:3d} x {launches} single-frame launches ({fpt} frames/thread, nt={args.nt}): {fps:9.0f} frames/s "
              f"= {fps * 2 * fb / 1e9:7.1f} GB/s = {fps * 2 * fb / 8e12:.3f} of HBM peak; slowest thread {max(per) * 1e3:.2f} ms, "
              f"fastest {min(per) * 1e3:.2f} ms", flush=True)
        del pool


if __name__ == "__main__":
    main()
