#!/usr/bin/env python3
"""tools/bench_kernels.py -- per-kernel device-resident timings (HIP events on the launch stream)
for every element kernel on the path, at the BASELINE configs.  Prints one JSON object per line.
Used for DESIGN.md's roofline table; bench.py stays the headline hsvfilter line."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import _pkg  # noqa: E402
from tests import cubes  # noqa: E402

vfx = _pkg.vfx
lib = vfx.lib()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
vfx.check(lib.mvfx_set_device(0))
stream = torch.cuda.current_stream(dev)
sptr = ctypes.c_void_p(stream.cuda_stream)
PEAK = 8000.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(iters):
        fn(i)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rand_frames(n, nbytes, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device=dev, generator=g)


def report(name, ms, bytes_per_call, frames_per_call, extra=None):
    gbs = bytes_per_call / (ms * 1e-3) / 1e9
    o = {"kernel": name, "ms_per_call": round(ms, 4), "frames_per_s": round(frames_per_call / (ms * 1e-3), 1),
         "algorithmic_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}
    if extra:
        o.update(extra)
    print(json.dumps(o), flush=True)


def smpte_like_gpu(n, w, h):
    import numpy as np
    from tests import frames
    f = torch.from_numpy(frames.smpte_like(w, h).reshape(-1)).to(dev)
    return f.unsqueeze(0).repeat(n, 1).contiguous()


def main():
    only = sys.argv[1:] if len(sys.argv) > 1 else None
    W, H = 3840, 2160
    NB = W * H * 4
    POOL = 16  # 16 x 33 MB = 531 MB > 256 MiB L3

    def want(k):
        return only is None or any(k.startswith(o) for o in only)

    if want("hsvfilter"):
        s = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
        for fmt in ("RGBA", "xBGR"):
            fr = rand_frames(POOL, NB, 1)
            ptrs = [fr[i].data_ptr() for i in range(POOL)]
            arr = (vfx.Frame * POOL)(*[vfx.make_frame(p, W, H, W * 4, fmt) for p in ptrs])
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(arr, POOL, ctypes.byref(s), sptr)))
            report(f"hsvfilter {fmt} 4K batch16", ms, POOL * 2 * NB, POOL)
            one = [(vfx.Frame * 1)(vfx.make_frame(p, W, H, W * 4, fmt)) for p in ptrs]
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(one[i % POOL], 1, ctypes.byref(s), sptr)), iters=64)
            report(f"hsvfilter {fmt} 4K single-frame launches", ms, 2 * NB, 1)
        fr = rand_frames(POOL, W * H * 3, 2)
        arr = (vfx.Frame * POOL)(*[vfx.make_frame(fr[i].data_ptr(), W, H, W * 3, "RGB") for i in range(POOL)])
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(arr, POOL, ctypes.byref(s), sptr)))
        report("hsvfilter RGB 4K batch16", ms, POOL * 2 * W * H * 3, POOL)

    if want("hsvdetector"):
        W2, H2 = 1920, 1080
        n = 64
        src = rand_frames(n, W2 * H2 * 4, 3)
        dst = torch.empty_like(src)
        s = vfx.HsvDetectorSettings(120.0, 40.0, 0.6, 0.4, 0.6, 0.4)
        fi = [vfx.make_frame(src[i].data_ptr(), W2, H2, W2 * 4, "RGBx") for i in range(n)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W2, H2, W2 * 4, "RGBA") for i in range(n)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_frame(ctypes.byref(fi[i % n]), ctypes.byref(fo[i % n]), ctypes.byref(s), sptr)), iters=128)
        report("hsvdetector RGBx->RGBA 1080p", ms, 2 * W2 * H2 * 4, 1)
        src = rand_frames(POOL, NB, 4)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBx") for i in range(POOL)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_frame(ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), ctypes.byref(s), sptr)), iters=64)
        report("hsvdetector RGBx->RGBA 4K", ms, 2 * NB, 1)

    if want("colorlut"):
        for size in (33, 17):
            lut = vfx.CubeLut(cubes.analytic_3d(size))
            for data in ("random", "smpte"):
                src = rand_frames(POOL, NB, 5) if data == "random" else smpte_like_gpu(POOL, W, H)
                dst = torch.empty_like(src)
                fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
                fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
                for placement in ((0, 1) if size <= 21 else (0,)):
                    vfx.check(lib.mvfx_colorlut_set_placement(placement))
                    ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), sptr)), iters=32)
                    report(f"colorlut 3D {size}^3 RGBA 4K {data} placement={'auto' if placement == 0 else 'global'}", ms, 2 * NB, 1)
                vfx.check(lib.mvfx_colorlut_set_placement(0))
        lut = vfx.CubeLut(cubes.curve_1d(1024))
        src = rand_frames(POOL, NB, 6)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), sptr)), iters=32)
        report("colorlut 1D 1024 RGBA 4K random", ms, 2 * NB, 1)
        lut = vfx.CubeLut(cubes.analytic_3d(33))
        src = rand_frames(8, NB * 2, 7)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % 8]), ctypes.byref(fo[i % 8]), sptr)), iters=16)
        report("colorlut 3D 33^3 RGBA64_LE 4K random", ms, 4 * NB, 1)

    # d2d copy ceiling measured on this box (SURVEY 8d asks for it next to the 8 TB/s spec)
    if want("copy"):
        a = rand_frames(POOL, NB, 8)
        b = torch.empty_like(a)
        ms = timeit(lambda i=0: b.copy_(a), iters=10)
        report("torch d2d copy 531 MB", ms, 2 * POOL * NB, POOL)


if __name__ == "__main__":
    main()
