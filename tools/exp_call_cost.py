#!/usr/bin/env python3
"""tools/exp_call_cost.py -- host cost per asynchronous C-ABI call on 64 x 64 frames (the GPU work is nothing): which entry points cost
the calling thread more than a launch should."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402
from tests import cubes  # noqa: E402

vfx = _pkg.vfx
lib = vfx.lib()
vfx.check(lib.mvfx_set_device(0))
dev = torch.device("cuda", 0)
W = H = 64
a = torch.zeros((4, W * H * 4), dtype=torch.uint8, device=dev)
b = torch.zeros((4, W * H * 4), dtype=torch.uint8, device=dev)
fa = [vfx.make_frame(a[i].data_ptr(), W, H, W * 4, "RGBx") for i in range(4)]
fr = [vfx.make_frame(a[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(4)]
fb = [vfx.make_frame(b[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(4)]
hs = vfx.HsvFilterSettings(45.0, 1.0, 0.0, 1.0, 0.0)
ds = vfx.HsvDetectorSettings(120.0, 60.0, 0.6, 0.4, 0.6, 0.4)
lut = vfx.CubeLut(cubes.analytic_3d(33))
vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fr[0]), ctypes.byref(fb[0]), None))
torch.cuda.synchronize()
calls = {
    "mvfx_hsvfilter_transform_frame_ip": lambda i: lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr[i & 3]), ctypes.byref(hs), None),
    "mvfx_hsvdetector_transform_frame": lambda i: lib.mvfx_hsvdetector_transform_frame(ctypes.byref(fa[i & 3]), ctypes.byref(fb[i & 3]), ctypes.byref(ds), None),
    "mvfx_colorlut_transform_frame": lambda i: lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fr[i & 3]), ctypes.byref(fb[i & 3]), None),
    "mvfx_stream_wait_event-free baseline (mvfx_thread_stream)": lambda i: lib.mvfx_thread_stream(),
}
for name, fn in calls.items():
    for _ in range(2000):
        fn(_)
    torch.cuda.synchronize()
    n = 50000
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {(t1 - t0) / n * 1e6:6.2f} us per call on the host ({n / (t2 - t0):8.0f} calls/s with the drain)", flush=True)
