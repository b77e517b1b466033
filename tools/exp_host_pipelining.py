#!/usr/bin/env python3
"""Experiment: a system-memory frame through hsvfilter as the `_host` entry point does it (H2D, kernel, D2H in series on one
stream) against a banded version -- the frame cut into N row bands, bands alternating between two streams, so the upload of the
next band overlaps the download of the previous one (PCIe is full duplex).  Pinned vs pageable host memory."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, _pkg
vfx = _pkg.vfx; lib = vfx.lib()
dev = torch.device("cuda", 0); vfx.check(lib.mvfx_set_device(0))
W, H = 3840, 2160; FB = W * H * 4
settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
dbuf = torch.empty(FB, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
p1, p2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)

def serial(host):
    lib.mvfx_copy_to_device_async(ctypes.c_void_p(dbuf.data_ptr()), ctypes.c_void_p(host.data_ptr()), FB, p1)
    f = vfx.make_frame(dbuf.data_ptr(), W, H, W * 4, "RGBA")
    vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(settings), p1))
    lib.mvfx_copy_to_host_async(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(dbuf.data_ptr()), FB, p1)
    lib.mvfx_stream_synchronize(p1)

def banded(host, n):
    rows = (H + n - 1) // n
    for b in range(n):
        r0 = b * rows; r = min(rows, H - r0)
        if r <= 0: break
        off, nbytes = r0 * W * 4, r * W * 4
        st = p1 if b % 2 == 0 else p2
        lib.mvfx_copy_to_device_async(ctypes.c_void_p(dbuf.data_ptr() + off), ctypes.c_void_p(host.data_ptr() + off), nbytes, st)
        f = vfx.make_frame(dbuf.data_ptr() + off, W, r, W * 4, "RGBA")
        vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(settings), st))
        lib.mvfx_copy_to_host_async(ctypes.c_void_p(host.data_ptr() + off), ctypes.c_void_p(dbuf.data_ptr() + off), nbytes, st)
    lib.mvfx_stream_synchronize(p1); lib.mvfx_stream_synchronize(p2)

def timeit(fn, iters=40):
    for _ in range(5): fn()
    t = time.perf_counter()
    for _ in range(iters): fn()
    return (time.perf_counter() - t) / iters * 1e3

for kind in ("pinned", "pageable"):
    host = torch.randint(0, 256, (FB,), dtype=torch.uint8)
    if kind == "pinned": host = host.pin_memory()
    print(f"{kind:9s} serial {timeit(lambda: serial(host)):.3f} ms   " + "   ".join(f"{n} bands {timeit(lambda n=n: banded(host, n)):.3f} ms" for n in (2, 4, 8, 16)), flush=True)
