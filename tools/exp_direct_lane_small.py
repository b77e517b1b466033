#!/usr/bin/env python3
"""tools/exp_direct_lane_small.py [w h] -- launch rate of the direct-dispatch lane against two HIP streams on small frames (host / packet cost only)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    fpt = 16
    pool = torch.randint(0, 256, (fpt, W * H * 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    fr = (vfx.Frame * fpt)(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
    n = 20000
    secs = (ctypes.c_double * 5)()
    per = (ctypes.c_double * 1)()
    rc = bench.mvfxbench_hsvfilter_streams_rot(0, 1, 2, 600, n, 5, fr, fpt, None, 0, ctypes.byref(settings), 0, secs, per)
    assert rc == 0
    a = n / sorted(secs)[2]
    took = ctypes.c_uint64()
    rc = bench.mvfxbench_hsvfilter_direct(0, 600, n, 5, fr, fpt, ctypes.byref(settings), 0, secs, ctypes.byref(took))
    assert rc == 0, (rc, vfx.last_error())
    b = n / sorted(secs)[2]
    print(f"{W}x{H}: two streams {a:8.0f} launches/s ({1e6 / a:.2f} us)   direct lane {b:8.0f} launches/s ({1e6 / b:.2f} us), {took.value} through the lane")


if __name__ == "__main__":
    main()
