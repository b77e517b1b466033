#!/usr/bin/env python3
"""tools/exp_ssim32_error.py -- how far the f32 SSIM pipeline (default) and its f64 twin (MVFX_OPT_SSIM_F64) are from the f64
checker oracle/ssim_oracle.c, case by case: relative error of the distance, and the time per pair on the device."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402
from tests import frames  # noqa: E402
from tests import oracle_binding as orc  # noqa: E402

vfx = _pkg.vfx
lib = vfx.lib()
vfx.check(lib.mvfx_set_device(0))


def natural(w, h, k=0):
    x = np.linspace(0, 1, w, dtype=np.float64)[None, :]
    y = np.linspace(0, 1, h, dtype=np.float64)[:, None]
    ph = 0.37 * k
    img = np.stack([0.5 + 0.45 * np.sin(3 * x + 2 * y + ph) + 0 * y, 0.5 + 0.45 * np.sin(5 * y - 1.5 * x + 2 * ph), 0.5 + 0.45 * np.cos(4 * x * y + ph),
                    np.ones((h, w))], axis=-1) * 255.0
    rng = np.random.default_rng(77 + k)
    noise = rng.integers(-3, 4, img.shape)
    noise[..., 3] = 0
    return np.clip(img + noise, 0, 255).astype(np.uint8).reshape(h, w * 4)


def gpu_distance(a, b, w, h, f64):
    da, db = vfx.DeviceBuffer(a.nbytes).upload(a), vfx.DeviceBuffer(b.nbytes).upload(b)
    fa, fb = vfx.make_frame(da.ptr, w, h, w * 4, "RGBA"), vfx.make_frame(db.ptr, w, h, w * 4, "RGBA")
    d = ctypes.c_double()
    with vfx.options(ssim_f64=f64):
        vfx.check(lib.mvfx_ssim_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), None))
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            vfx.check(lib.mvfx_ssim_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), None))
        dt = (time.perf_counter() - t) / n
    return d.value, dt


cases = []
for (w, h) in ((320, 240), (1920, 1080), (3840, 2160)):
    a = frames.random_frame(0xE000 + w, w, h)
    for amp, every in ((1, 97), (25, 13), (120, 5)):
        b = a.copy()
        flat = b.reshape(-1)
        idx = np.arange(0, flat.size, every)
        flat[idx] = np.clip(flat[idx].astype(np.int32) + amp, 0, 255).astype(np.uint8)
        cases.append((f"random {w}x{h} +{amp} every {every}", a, b, w, h))
    n = natural(w, h)
    m = n.copy()
    m[h // 3: h // 3 + 40, 400:1200] ^= 0x08
    cases.append((f"natural {w}x{h}, a block of low-bit flips", n, m, w, h))
    n2 = natural(w, h, 1)
    cases.append((f"natural {w}x{h}, two phases", n, n2, w, h))
    flat_a = np.full((h, w * 4), 200, np.uint8)
    flat_b = flat_a.copy()
    flat_b[:, ::8] = 201
    cases.append((f"flat bright {w}x{h}, every other pixel's red +1", flat_a, flat_b, w, h))
print(f"{'case':58s} {'oracle f64':>14s} {'rel err f64 twin':>17s} {'rel err f32':>12s} {'ms f64':>8s} {'ms f32':>8s}")
for name, a, b, w, h in cases:
    rc, want, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    g64, t64 = gpu_distance(a, b, w, h, True)
    g32, t32 = gpu_distance(a, b, w, h, False)
    z, _ = gpu_distance(a, a, w, h, False)
    rel = lambda g: abs(g - want) / abs(want) if want else abs(g)
    print(f"{name:58s} {want:14.8e} {rel(g64):17.2e} {rel(g32):12.2e} {t64 * 1e3:8.3f} {t32 * 1e3:8.3f}  identical->{z}")
