#!/usr/bin/env python3
"""tools/exp_single_frame_trace.py -- 2000 single-frame hsvfilter launches on 4K RGBA frames (one stream, back to back; then alternating
between two streams): wall clock per call, to be set beside the kernel durations of `rocprofv3 --kernel-trace --stats` of this script."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H, N = 3840, 2160, 16
    buf = torch.randint(0, 256, (N, W * H * 4), dtype=torch.uint8, device=dev)
    fr = [vfx.make_frame(buf[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)]
    s = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    streams = [ctypes.c_void_p(lib.mvfx_thread_stream_n(k)) for k in range(2)]
    for mode, pick in (("one stream", lambda i: streams[0]), ("two streams alternating", lambda i: streams[i & 1])):
        for i in range(400):  # (warm-up)
            vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr[i % N]), ctypes.byref(s), pick(i)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2000
        for i in range(n):
            vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr[i % N]), ctypes.byref(s), pick(i)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mode}: {dt / n * 1e6:.2f} us per call end to end ({n / dt:.0f} frames/s = {n / dt * 2 * W * H * 4 / 8e12:.3f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
