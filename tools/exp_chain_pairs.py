#!/usr/bin/env python3
"""tools/exp_chain_pairs.py -- round 4: the device chain hsvfilter -> hsvdetector -> colorlut on 4K frames from ONE host thread through the
C ABI (no GStreamer): (a) a launch per frame per stage, frame k's three kernels on stream k & 1 (what the elements do in a chain);
(b) two frames per launch per stage, pair p's three kernels on stream p & 1 (what chain-wide pair launches would reach at best).
Decides whether teaching the elements to keep PAIRS together through a chain is worth building."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402
from tests import cubes, frames as _frames  # noqa: E402

vfx = _pkg.vfx
lib = vfx.lib()
lib.mvfx_thread_stream_n.restype = ctypes.c_void_p
dev = torch.device("cuda", 0)
vfx.check(lib.mvfx_set_device(0))
W, H, N = 3840, 2160, 8
base = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
a = base.unsqueeze(0).repeat(N, 1).contiguous()
b = torch.empty_like(a)
c = torch.empty_like(a)
torch.cuda.synchronize()
mk = lambda t, fmt: (vfx.Frame * N)(*[vfx.make_frame(t[i].data_ptr(), W, H, W * 4, fmt) for i in range(N)])
fa, fb_, fc = mk(a, "RGBx"), mk(b, "RGBA"), mk(c, "RGBA")
hs = vfx.HsvFilterSettings(45.0, 1.0, 0.0, 1.0, 0.0)
ds = vfx.HsvDetectorSettings(120.0, 60.0, 0.6, 0.4, 0.6, 0.4)
lut = vfx.CubeLut(cubes.analytic_3d(33))
st = [ctypes.c_void_p(lib.mvfx_thread_stream_n(k)) for k in range(2)]
F = ctypes.sizeof(vfx.Frame)
at = lambda arr, i: ctypes.cast(ctypes.addressof(arr) + i * F, ctypes.POINTER(vfx.Frame))


def single(frames_total):
    for k in range(frames_total):
        i, s = k % N, st[k & 1]
        vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(at(fa, i), ctypes.byref(hs), s))
        vfx.check(lib.mvfx_hsvdetector_transform_frame(at(fa, i), at(fb_, i), ctypes.byref(ds), s))
        vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, at(fb_, i), at(fc, i), s))


def pairs(frames_total):
    for p in range(frames_total // 2):
        i, s = (2 * p) % N, st[p & 1]
        vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(at(fa, i), 2, ctypes.byref(hs), s))
        vfx.check(lib.mvfx_hsvdetector_transform_frames(at(fa, i), at(fb_, i), 2, ctypes.byref(ds), s))
        vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, at(fb_, i), at(fc, i), 2, s))


for name, fn in (("a launch per frame per stage, frame-parity streams", single), ("two frames per launch per stage, pair-parity streams", pairs)):
    res = []
    for rep in range(5):
        fn(400)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(4000)
        torch.cuda.synchronize()
        res.append(4000 / (time.perf_counter() - t0))
    print(f"{name}: {sorted(res)[2]:8.0f} frames/s (runs {[round(x) for x in res]})", flush=True)
