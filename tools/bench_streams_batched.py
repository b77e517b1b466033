#!/usr/bin/env python3
"""T host threads x own HIP stream x launches of B frames each (mvfx_hsvfilter_transform_frames_ip): the grid between
bench.py's two launch models (1 thread x 16 frames, 16 threads x 1 frame).

    python tools/bench_streams_batched.py [--grid 1x16,2x16,2x8,4x4,4x8,8x2,16x1] [--nt 1]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="1x16,2x16,2x8,4x4,4x8,8x2,8x4,16x1,1x32,2x32")
    ap.add_argument("--pool-frames", type=int, default=416, help="distinct resident frames shared out over the threads")
    ap.add_argument("--seconds", type=float, default=0.5)
    ap.add_argument("--nt", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H = 3840, 2160
    fb = W * H * 4
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    opts = vfx.OPT_NONTEMPORAL if args.nt else 0
    pool = torch.randint(0, 256, (args.pool_frames, fb), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for cell in args.grid.split(","):
        n, b = (int(x) for x in cell.split("x"))
        fpt = (args.pool_frames // n) // b * b
        frames = (vfx.Frame * (n * fpt))(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(n * fpt)])
        secs = (ctypes.c_double * args.reps)()
        per = (ctypes.c_double * n)()
        launches = max(20, int(args.seconds * 80000 / (n * b)))
        for _ in range(2):  # first pass settles the clocks
            rc = bench.mvfxbench_hsvfilter_streams_batched(0, n, 10, launches, args.reps, frames, fpt, b, ctypes.byref(settings), opts, secs, per)
            assert rc == 0, (rc, vfx.last_error())
        s = sorted(secs)
        fps = [n * b * launches / x for x in (s[-1], s[len(s) // 2], s[0])]
        print(f"{n:3d} threads x {b:2d} frames/launch x {launches} launches: min/median/max {fps[0]:8.0f} {fps[1]:8.0f} {fps[2]:8.0f} frames/s "
              f"= {fps[1] * 2 * fb / 8e12:.3f} of HBM peak (median)", flush=True)


if __name__ == "__main__":
    main()
