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
if os.environ.get("MVFX_TYPED_LOADS"):
    vfx.check(lib.mvfx_thread_set_options(vfx.options(typed=bool(int(os.environ["MVFX_TYPED_LOADS"])).word)))
stream = torch.cuda.current_stream(dev)
sptr = ctypes.c_void_p(stream.cuda_stream)
PEAK = 8000.0


def timeit(fn, iters=300, warm=3, settle_s=0.4):
    """Steady clock state first (the governor needs ~0.2 s of load, profiles/r1/exp_ramp_launch_series.txt)."""
    import time
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    k = 0
    while time.perf_counter() - t < settle_s:
        for _ in range(20):
            fn(k)
            k += 1
        torch.cuda.synchronize()
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


NOISE = int(os.environ.get("MVFX_BENCH_NOISE", "3"))


def natural_like_gpu(n, w, h, seed):
    """Smooth 2-D colour gradients + sensor-like noise of +-3 codes (MVFX_BENCH_NOISE): neighbouring pixels mostly share a LUT cell."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.linspace(0, 1, w, device=dev).view(1, 1, w)
    y = torch.linspace(0, 1, h, device=dev).view(1, h, 1)
    ph = torch.rand((n, 1, 1), device=dev, generator=g) * 6.28
    r = 0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)
    gch = 0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)
    b = 0.5 + 0.45 * torch.cos(4.0 * x * y + ph)
    img = torch.stack([r.expand(n, h, w), gch.expand(n, h, w), b.expand(n, h, w), torch.ones((n, h, w), device=dev)], dim=-1) * 255.0
    img = img + torch.randint(-NOISE, NOISE + 1, img.shape, device=dev, generator=g).float() * torch.tensor([1, 1, 1, 0], device=dev)
    return img.clamp(0, 255).to(torch.uint8).view(n, -1).contiguous()


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
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(one[i % POOL], 1, ctypes.byref(s), sptr)), iters=300)
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
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_frame(ctypes.byref(fi[i % n]), ctypes.byref(fo[i % n]), ctypes.byref(s), sptr)), iters=300)
        report("hsvdetector RGBx->RGBA 1080p", ms, 2 * W2 * H2 * 4, 1)
        src = rand_frames(POOL, NB, 4)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBx") for i in range(POOL)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_frame(ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), ctypes.byref(s), sptr)), iters=300)
        report("hsvdetector RGBx->RGBA 4K", ms, 2 * NB, 1)
        fia, foa = (vfx.Frame * POOL)(*fi), (vfx.Frame * POOL)(*fo)
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_frames(fia, foa, POOL, ctypes.byref(s), sptr)), iters=60)
        report(f"hsvdetector RGBx->RGBA 4K batch{POOL} (one launch)", ms, 2 * NB * POOL, POOL)

    if want("baked"):
        # placement 6 (the LUT baked into a 64 MiB table of all 2^24 colours, one gather per pixel) against the automatic choice
        # (the interpolating tile kernel), same frames, single-frame launches and 16 frames per launch
        lut = vfx.CubeLut(cubes.analytic_3d(33))
        for data in ("natural", "smpte", "random"):
            src = rand_frames(POOL, NB, 5) if data == "random" else (smpte_like_gpu(POOL, W, H) if data == "smpte" else natural_like_gpu(POOL, W, H, 11))
            dst = torch.empty_like(src)
            fi = (vfx.Frame * POOL)(*[vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)])
            fo = (vfx.Frame * POOL)(*[vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)])
            for placement in (0, 5, 6):
                vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=placement).word))
                what = {0: "auto (x-prelerped tile kernel)", 5: "tile kernel of round 2", 6: "baked table"}[placement]
                ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), sptr)), iters=300)
                report(f"colorlut 3D 33^3 RGBA 4K {data} placement={what}", ms, 2 * NB, 1)
                ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi, fo, POOL, sptr)), iters=40 if data != "random" else 10)
                report(f"colorlut 3D 33^3 RGBA 4K {data} batch{POOL} (one launch) placement={what}", ms, 2 * NB * POOL, POOL)
            vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=0).word))
        import time as _t
        for size in (33, 65):
            l2 = vfx.CubeLut(cubes.analytic_3d(size))
            vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=6).word))
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            vfx.check(lib.mvfx_colorlut_transform_frame(l2.h, ctypes.byref(fi[0]), ctypes.byref(fo[0]), sptr))
            torch.cuda.synchronize()
            print(json.dumps({"kernel": f"colorlut baked table: first call with a new {size}^3 LUT (upload + bake 2^24 colours + one 4K frame)",
                              "ms": round((_t.perf_counter() - t0) * 1e3, 3)}), flush=True)
            vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=0).word))

    if want("colorlut"):
        for size in (33, 65, 17, 9):
            lut = vfx.CubeLut(cubes.analytic_3d(size))
            for data in ("random", "smpte", "natural"):
                src = rand_frames(POOL, NB, 5) if data == "random" else (smpte_like_gpu(POOL, W, H) if data == "smpte" else natural_like_gpu(POOL, W, H, 11))
                dst = torch.empty_like(src)
                fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
                fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
                for placement in ((0, 1, 2, 5) if size <= 21 else (0, 5)):
                    vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=placement).word))
                    ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), sptr)), iters=300)
                    report(f"colorlut 3D {size}^3 RGBA 4K {data} placement={ {0: 'auto', 1: 'global nodes', 2: 'LDS cube', 5: 'tile kernel (round 2)'}[placement]}", ms, 2 * NB, 1)
                vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=0).word))
        lut = vfx.CubeLut(cubes.analytic_3d(33))
        src = natural_like_gpu(POOL, W, H, 11)
        dst = torch.empty_like(src)
        fi = (vfx.Frame * POOL)(*[vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)])
        fo = (vfx.Frame * POOL)(*[vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)])
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi, fo, POOL, sptr)), iters=40)
        report(f"colorlut 3D 33^3 RGBA 4K natural batch{POOL} (one launch)", ms, 2 * NB * POOL, POOL)
        lut17 = vfx.CubeLut(cubes.analytic_3d(17))
        for placement in (2, 5):
            vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=placement).word))
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frames(lut17.h, fi, fo, POOL, sptr)), iters=40)
            report(f"colorlut 3D 17^3 RGBA 4K natural batch{POOL} (one launch) placement={ {2: 'LDS cube', 5: 'tile kernel'}[placement]}", ms, 2 * NB * POOL, POOL)
        vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=0).word))
        lut = vfx.CubeLut(cubes.curve_1d(1024))
        src = rand_frames(POOL, NB, 6)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), sptr)), iters=300)
        report("colorlut 1D 1024 RGBA 4K random", ms, 2 * NB, 1)
        lut = vfx.CubeLut(cubes.analytic_3d(33))
        src = rand_frames(8, NB * 2, 7)
        dst = torch.empty_like(src)
        fi = [vfx.make_frame(src[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        fo = [vfx.make_frame(dst[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % 8]), ctypes.byref(fo[i % 8]), sptr)), iters=300)
        report("colorlut 3D 33^3 RGBA64_LE 4K random", ms, 4 * NB, 1)
        # natural-like 16-bit frames: the 8-bit picture scaled by 257 plus noise in the low byte (+-3 codes of 8-bit worth)
        nat8 = natural_like_gpu(8, W, H, 21).view(8, H * W * 4).to(torch.int32)
        nat16 = (nat8 * 257 + torch.randint(-128, 129, nat8.shape, device=dev, dtype=torch.int32)).clamp(0, 65535).to(torch.int16)
        src64 = nat16.view(torch.uint8).view(8, -1).contiguous()
        dst64 = torch.empty_like(src64)
        fi = [vfx.make_frame(src64[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        fo = [vfx.make_frame(dst64[i].data_ptr(), W, H, W * 8, "RGBA64_LE") for i in range(8)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fi[i % 8]), ctypes.byref(fo[i % 8]), sptr)), iters=300)
        report("colorlut 3D 33^3 RGBA64_LE 4K natural", ms, 4 * NB, 1)

    if want("colordetect"):
        src = rand_frames(POOL, NB, 9)
        hist = torch.zeros(32768 + 8, dtype=torch.int32, device=dev)
        fr = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        for q in (10, 1):
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_colordetect_histogram(ctypes.byref(fr[i % POOL]), q, 0, vfx.ALL_SAMPLES,
                        ctypes.c_void_p(hist.data_ptr()), ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr)), iters=300)
            report(f"colordetect histogram 4K RGBA quality={q}", ms, NB, 1)
        srcn = natural_like_gpu(POOL, W, H, 13)
        frn = [vfx.make_frame(srcn[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colordetect_histogram(ctypes.byref(frn[i % POOL]), 10, 0, vfx.ALL_SAMPLES,
                    ctypes.c_void_p(hist.data_ptr()), ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr)), iters=300)
        report("colordetect histogram 4K RGBA quality=10 natural content", ms, NB, 1)
        # one frame of each of 16 streams per pair of launches (mvfx_colordetect_histogram_frames), two rotating batches
        big = rand_frames(2 * POOL, NB, 19)
        recs = torch.zeros(POOL * vfx.COLORDETECT_RECORD_WORDS, dtype=torch.int32, device=dev)
        arrs = [(vfx.Frame * POOL)(*[vfx.make_frame(big[b * POOL + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]) for b in range(2)]
        for q in (10, 1):
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_colordetect_histogram_frames(arrs[i & 1], POOL, q, ctypes.c_void_p(recs.data_ptr()), sptr)),
                        iters=100)
            report(f"colordetect histogram 4K RGBA quality={q}, 16 frames per launch", ms, POOL * NB, POOL)

    if want("blockhash"):
        W8, H8 = 7680, 4320
        src = rand_frames(4, W8 * H8 * 4, 10)
        sums = torch.zeros(64, dtype=torch.int32, device=dev)
        fr = [vfx.make_frame(src[i].data_ptr(), W8, H8, W8 * 4, "RGBA") for i in range(4)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_blockhash_sums(ctypes.byref(fr[i % 4]), 0, H8, ctypes.c_void_p(sums.data_ptr()), sptr)), iters=300)
        report("blockhash sums 8K RGBA (one frame)", ms, W8 * H8 * 4, 1)
        sums2 = torch.zeros(128, dtype=torch.int32, device=dev)
        pairs = [(vfx.Frame * 2)(fr[(2 * k) % 4], fr[(2 * k + 1) % 4]) for k in range(2)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_blockhash_sums_pads(pairs[i % 2], 2, H8, 0, ctypes.c_void_p(sums2.data_ptr()), sptr)), iters=300)
        report("blockhash sums 8K RGBA pair (one launch)", ms, 2 * W8 * H8 * 4, 2)
        src4 = rand_frames(POOL, NB, 11)
        fr = [vfx.make_frame(src4[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_blockhash_sums(ctypes.byref(fr[i % POOL]), 0, H, ctypes.c_void_p(sums.data_ptr()), sptr)), iters=300)
        report("blockhash sums 4K RGBA", ms, NB, 1)

    if want("imghash"):
        W8, H8 = 7680, 4320
        src = rand_frames(4, W8 * H8 * 4, 14)
        fr = [vfx.make_frame(src[i].data_ptr(), W8, H8, W8 * 4, "RGBA") for i in range(4)]
        hv = ctypes.c_uint64()
        for name, algo in (("mean 8x8", 0), ("gradient 9x8", 1), ("doublegradient 5x5", 3)):
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_image_hash(ctypes.byref(fr[i % 4]), algo, ctypes.byref(hv), None, sptr)), iters=50, settle_s=0.3)
            report(f"videocompare {name} hash 8K RGBA (gray + Lanczos3 resize, synchronous)", ms, W8 * H8 * 4, 1)

    if want("ssim"):
        for (w, h, tag) in ((W, H, "4K"), (7680, 4320, "8K")):
            if os.environ.get("SSIM_ONLY", tag) != tag:
                continue
            src = rand_frames(2, w * h * 4, 12)
            src[1] = src[0]
            src[1, :: 97] ^= 0x10
            fa = vfx.make_frame(src[0].data_ptr(), w, h, w * 4, "RGBA")
            fb = vfx.make_frame(src[1].data_ptr(), w, h, w * 4, "RGBA")
            d = ctypes.c_double()
            ms = timeit(lambda i=0: vfx.check(lib.mvfx_ssim_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), sptr)), iters=10, settle_s=0.5)
            report(f"videocompare dssim {tag} RGBA pair (5 scales, f32 pipeline)", ms, 2 * w * h * 4, 1, {"distance": d.value})

    if want("roundedcorners"):
        mask = torch.empty(W * H, dtype=torch.uint8, device=dev)
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_roundedcorners_mask(ctypes.c_void_p(mask.data_ptr()), W, H, W, 100, sptr)), iters=300)
        report("roundedcorners mask 4K r=100 (once per config)", ms, W * H, 1)
        i420 = rand_frames(POOL, W * H * 3 // 2, 12)
        a420 = torch.empty((POOL, W * H * 5 // 2), dtype=torch.uint8, device=dev)
        def compose(i=0):
            k = i % POOL
            fi, fo = vfx.PlanarFrame(), vfx.PlanarFrame()
            offs = [0, W * H, W * H * 5 // 4, W * H * 3 // 2]
            for p_ in range(3):
                fi.data[p_] = i420[k].data_ptr() + offs[p_]
                fo.data[p_] = a420[k].data_ptr() + offs[p_]
                fi.stride[p_] = fo.stride[p_] = W if p_ == 0 else W // 2
            fo.data[3] = a420[k].data_ptr() + offs[3]
            fo.stride[3] = W
            fi.width = fo.width = W
            fi.height = fo.height = H
            fi.format, fo.format = vfx.FORMATS["I420"], vfx.FORMATS["A420"]
            vfx.check(lib.mvfx_roundedcorners_compose_a420(ctypes.byref(fi), ctypes.c_void_p(mask.data_ptr()), W, ctypes.byref(fo), sptr))
        ms = timeit(compose, iters=300)
        report("roundedcorners I420->A420 compose 4K", ms, W * H * 4, 1)

    if want("convert"):
        ys, cs = W, W // 2
        isz = W * H * 3 // 2
        i420 = rand_frames(POOL, isz, 15)
        rgba = torch.empty((POOL, NB), dtype=torch.uint8, device=dev)
        fi = [vfx.make_i420(i420[i].data_ptr(), W, H, ys, cs, W * H, W * H * 5 // 4) for i in range(POOL)]
        fo = [vfx.make_frame(rgba[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_convert_i420_to_rgba(ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), 0, sptr)), iters=300)
        report("convert I420->RGBA 4K (videoconvert-equivalent)", ms, W * H * 11 // 2, 1)
        src = rand_frames(POOL, NB, 16)
        fr = [vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_convert_rgba_to_i420(ctypes.byref(fr[i % POOL]), ctypes.byref(fi[i % POOL]), 0, sptr)), iters=300)
        report("convert RGBA->I420 4K (videoconvert-equivalent, BT.2020 h-cosited)", ms, W * H * 11 // 2, 1)
        fia = (vfx.PlanarFrame * POOL)(*fi)
        foa = (vfx.Frame * POOL)(*fo)
        fra = (vfx.Frame * POOL)(*fr)
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_convert_i420_to_rgba_frames(fia, foa, POOL, 0, sptr)), iters=60)
        report(f"convert I420->RGBA 4K batch{POOL} (one launch)", ms, POOL * W * H * 11 // 2, POOL)
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_convert_rgba_to_i420_frames(fra, fia, POOL, 0, sptr)), iters=60)
        report(f"convert RGBA->I420 4K batch{POOL} (one launch)", ms, POOL * W * H * 11 // 2, POOL)

    if want("lut420"):
        from tests import cubes as _c
        lut = vfx.CubeLut(_c.analytic_3d(33))
        isz = W * H * 3 // 2
        # natural-like I420: convert the natural RGBA frames with our own converter
        srcn = natural_like_gpu(POOL, W, H, 17)
        i420 = torch.empty((POOL, isz), dtype=torch.uint8, device=dev)
        o420 = torch.empty((POOL, isz), dtype=torch.uint8, device=dev)
        fr = [vfx.make_frame(srcn[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        fi = [vfx.make_i420(i420[i].data_ptr(), W, H, W, W // 2, W * H, W * H * 5 // 4) for i in range(POOL)]
        fo = [vfx.make_i420(o420[i].data_ptr(), W, H, W, W // 2, W * H, W * H * 5 // 4) for i in range(POOL)]
        for i in range(POOL):
            vfx.check(lib.mvfx_convert_rgba_to_i420(ctypes.byref(fr[i]), ctypes.byref(fi[i]), 0, sptr))
        torch.cuda.synchronize()
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_i420(lut.h, ctypes.byref(fi[i % POOL]), ctypes.byref(fo[i % POOL]), 0, sptr)), iters=200)
        report("colorlut 33^3 on I420 4K natural, fused videoconvert!colorlut!videoconvert (one kernel)", ms, W * H * 3, 1)
        # one flat colour: every pixel repeats its lane's previous LUT cell -> no gathers at all: the floor of the fused kernel
        flat = torch.full((POOL, isz), 0x6B, dtype=torch.uint8, device=dev)
        ff = [vfx.make_i420(flat[i].data_ptr(), W, H, W, W // 2, W * H, W * H * 5 // 4) for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_colorlut_transform_i420(lut.h, ctypes.byref(ff[i % POOL]), ctypes.byref(fo[i % POOL]), 0, sptr)), iters=200)
        report("colorlut 33^3 on I420 4K one flat colour (no LUT gathers), fused kernel", ms, W * H * 3, 1)
        tmp_a = torch.empty((POOL, NB), dtype=torch.uint8, device=dev)
        tmp_b = torch.empty((POOL, NB), dtype=torch.uint8, device=dev)
        fa = [vfx.make_frame(tmp_a[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]
        fb = [vfx.make_frame(tmp_b[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(POOL)]

        def three(i=0):
            k = i % POOL
            vfx.check(lib.mvfx_convert_i420_to_rgba(ctypes.byref(fi[k]), ctypes.byref(fa[k]), 0, sptr))
            vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, ctypes.byref(fa[k]), ctypes.byref(fb[k]), sptr))
            vfx.check(lib.mvfx_convert_rgba_to_i420(ctypes.byref(fb[k]), ctypes.byref(fo[k]), 0, sptr))
        ms = timeit(three, iters=200)
        report("colorlut 33^3 on I420 4K natural, three launches through RGBA frames", ms, W * H * 3, 1)
        hs = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
        r420 = rand_frames(POOL, isz, 18)
        fri = [vfx.make_i420(r420[i].data_ptr(), W, H, W, W // 2, W * H, W * H * 5 // 4) for i in range(POOL)]
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvfilter_transform_i420(ctypes.byref(fri[i % POOL]), ctypes.byref(fo[i % POOL]), ctypes.byref(hs), 0, sptr)), iters=300)
        report("hsvfilter on I420 4K random, fused videoconvert!hsvfilter!videoconvert (one kernel)", ms, W * H * 3, 1)

        def three_h(i=0):
            k = i % POOL
            vfx.check(lib.mvfx_convert_i420_to_rgba(ctypes.byref(fri[k]), ctypes.byref(fa[k]), 0, sptr))
            vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fa[k]), ctypes.byref(hs), sptr))
            vfx.check(lib.mvfx_convert_rgba_to_i420(ctypes.byref(fa[k]), ctypes.byref(fo[k]), 0, sptr))
        ms = timeit(three_h, iters=300)
        report("hsvfilter on I420 4K random, three launches through an RGBA frame", ms, W * H * 3, 1)
        ds = vfx.HsvDetectorSettings(120.0, 40.0, 0.6, 0.4, 0.6, 0.4)
        ms = timeit(lambda i=0: vfx.check(lib.mvfx_hsvdetector_transform_i420(ctypes.byref(fri[i % POOL]), ctypes.byref(fa[i % POOL]), ctypes.byref(ds), 0, sptr)), iters=300)
        report("hsvdetector on I420 4K random -> RGBA, fused videoconvert!hsvdetector (one kernel)", ms, W * H * 11 // 2, 1)

    # d2d copy ceiling measured on this box (SURVEY 8d asks for it next to the 8 TB/s spec)
    if want("copy"):
        a = rand_frames(POOL, NB, 8)
        b = torch.empty_like(a)
        ms = timeit(lambda i=0: b.copy_(a), iters=300)
        report("torch d2d copy 531 MB", ms, 2 * POOL * NB, POOL)


if __name__ == "__main__":
    main()
