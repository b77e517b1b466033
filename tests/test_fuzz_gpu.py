"""Randomised parity sweeps (GPU): the continuous settings space cannot be enumerated, so on top of
the exhaustive-RGB tests (which fix the settings) these tests draw many settings vectors, including
the edges of the strength-reduced kernels' domains, and compare HIP vs oracle on a pixel set that
contains every grey, the primaries, near-equal channels and 32K random pixels."""
import ctypes

import numpy as np
import pytest

from tests import cubes, frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu


def _pixels():
    rnd = frames.splitmix64_bytes(0xF0220001, 32768 * 4).reshape(-1, 4)
    g = np.arange(256, dtype=np.uint8)
    greys = np.stack([g, g, g, g], axis=1)
    near = np.stack([g, np.roll(g, 1), g, 255 - g], axis=1)
    prim = np.array([[255, 0, 0, 0], [0, 255, 0, 1], [0, 0, 255, 2], [255, 255, 0, 3], [0, 255, 255, 4], [255, 0, 255, 5],
                     [1, 0, 0, 6], [0, 1, 0, 7], [0, 0, 1, 8], [254, 255, 255, 9], [255, 254, 255, 10], [255, 255, 254, 11]], np.uint8)
    px = np.concatenate([rnd, greys, near, prim])
    pad = (-len(px)) % 4
    if pad:
        px = np.concatenate([px, px[:pad]])
    return np.ascontiguousarray(px).reshape(1, -1)


def _settings(rng, n):
    out = []
    shifts = [0.0, -0.0, 360.0, -360.0, 359.99997, -359.99997, 180.0, 1e-30, -1e-30, 1e-3, 60.0, 120.0, 300.0, -60.0, 0.5, 719.0, -400.0]
    for i in range(n):
        kind = rng.integers(0, 4)
        shift = shifts[i % len(shifts)] if i < 2 * len(shifts) else float(np.float32(rng.uniform(-360, 360)))
        if kind == 0:
            s = (shift, 1.0, 0.0, 1.0, 0.0)
        elif kind == 1:
            s = (shift, float(np.float32(rng.uniform(0, 3))), float(np.float32(rng.uniform(-1, 1))),
                 float(np.float32(rng.uniform(0, 3))), float(np.float32(rng.uniform(-1, 1))))
        elif kind == 2:
            s = (shift, float(np.float32(rng.uniform(-2, 2))), float(np.float32(rng.normal(0, 2))),
                 float(np.float32(rng.uniform(-2, 2))), float(np.float32(rng.normal(0, 2))))
        else:
            s = (shift, float(np.float32(10.0 ** rng.uniform(-3, 3))), float(np.float32(-10.0 ** rng.uniform(-3, 1))),
                 float(np.float32(10.0 ** rng.uniform(-3, 3))), float(np.float32(10.0 ** rng.uniform(-4, 0))))
        out.append(s)
    return out


def test_hsvfilter_random_settings(gpu):
    px = _pixels()
    n = px.size // 4
    rng = np.random.default_rng(0xF0220002)
    buf = gpu.DeviceBuffer(px.nbytes)
    for fmt in ("RGBA", "xBGR"):
        for s in _settings(rng, 120):
            exp = px.copy()
            assert orc.hsvfilter(exp, n, n * 4, fmt, s) == 0
            buf.upload(px)
            gpu.hsvfilter_device(buf.ptr, n, 1, n * 4, fmt, gpu.HsvFilterSettings(*s))
            gpu.check(gpu.lib().mvfx_stream_synchronize(None))
            got = buf.download().reshape(px.shape)
            bad = np.count_nonzero(got != exp)
            assert bad == 0, f"{fmt} settings {s}: {bad} bytes differ"


def test_hsvdetector_random_settings(gpu):
    px = _pixels()
    n = px.size // 4
    rng = np.random.default_rng(0xF0220003)
    src = gpu.DeviceBuffer(px.nbytes).upload(px)
    dst = gpu.DeviceBuffer(px.nbytes)
    fi = gpu.make_frame(src.ptr, n, 1, n * 4, "BGRx")
    fo = gpu.make_frame(dst.ptr, n, 1, n * 4, "ARGB")
    refs = [0.0, 180.0, -180.0, 540.0, 539.9999, -180.0001, 360.0, 90.0, 1e-20]
    for i in range(150):
        href = refs[i % len(refs)] if i < 2 * len(refs) else float(np.float32(rng.uniform(-400, 700)))
        s = (href, float(np.float32(rng.choice([0.0, 10.0, 180.0, rng.uniform(0, 180)]))),
             float(np.float32(rng.uniform(0, 1))), float(np.float32(rng.choice([0.0, 0.15, 1.0, rng.uniform(0, 1)]))),
             float(np.float32(rng.uniform(0, 1))), float(np.float32(rng.choice([0.0, 0.3, 1.0, rng.uniform(0, 1)]))))
        exp = np.empty_like(px)
        assert orc.hsvdetector(px, n * 4, "BGRx", exp, n * 4, "ARGB", n, s) == 0
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(gpu.HsvDetectorSettings(*s)), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        got = dst.download().reshape(px.shape)
        assert np.array_equal(got, exp), f"settings {s}"


def _random_cube(rng, size, one_d, wild):
    lines = [f"LUT_{'1' if one_d else '3'}D_SIZE {size}"]
    if wild:
        lo = rng.uniform(-0.5, 0.2, 3)
        hi = lo + rng.uniform(0.3, 2.0, 3)
        lines.append("DOMAIN_MIN %.6f %.6f %.6f" % tuple(lo))
        lines.append("DOMAIN_MAX %.6f %.6f %.6f" % tuple(hi))
    n = size if one_d else size ** 3
    vals = rng.uniform(-0.3, 1.3, (n, 3)) if wild else rng.uniform(0, 1, (n, 3))
    if wild:  # a few special nodes: exact 0 / 1, huge, tiny
        vals[rng.integers(0, n, 4)] = [[0, 1, 0.5], [1e30, -1e30, 1e-30], [1, 1, 1], [0, 0, 0]]
    lines += ["%.9g %.9g %.9g" % tuple(v) for v in vals]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("case", [(2, False, False), (5, False, True), (21, False, True), (33, False, True), (64, False, False),
                                  (2, True, True), (17, True, True), (4096, True, False), (5000, True, True)],
                         ids=lambda c: f"{'1d' if c[1] else '3d'}_{c[0]}{'_wild' if c[2] else ''}")
def test_colorlut_random_luts(gpu, case):
    size, one_d, wild = case
    rng = np.random.default_rng(0xF0220004 + size)
    text = _random_cube(rng, size, one_d, wild)
    o = orc.CubeLut(text)
    assert o.ok, o.error
    dev = gpu.CubeLut(text)
    px = _pixels()
    n = px.size // 4
    for fmt, bpp in (("RGBA", 4), ("RGBA64_LE", 8), ("RGBA64_BE", 8)):
        w = n if bpp == 4 else n // 2
        exp = np.empty_like(px)
        assert o.apply(px, w * bpp, exp, w * bpp, w, 1, fmt) == 0
        got = np.empty_like(px)
        dev.apply_host(px.reshape(-1), w * bpp, got.reshape(-1), w * bpp, w, 1, fmt)
        bad = np.count_nonzero(got != exp)
        assert bad == 0, f"{fmt}: {bad} bytes differ"
