"""GPU parity tests for hsvfilter / hsvdetector: HIP path (through the C ABI) vs the oracle.

Bit-exact u8 is the bar.  The exhaustive frame (all 2^24 RGB triples) makes each settings
vector a complete proof for the per-pixel function.
"""
import ctypes

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu

BENCH_SETTINGS = (90.0, 1.25, -0.05, 0.9, 0.02)
FILTER_SETTINGS = [
    (0.0, 1.0, 0.0, 1.0, 0.0),          # defaults (not an identity: SURVEY F5)
    BENCH_SETTINGS,
    (-123.4, 0.5, 0.3, 1.7, -0.2),
    (360.0, 1.0, 0.0, 1.0, 0.0),        # h + 360 can reach exactly 360 / wrap
    (-360.0, 2.0, -0.5, 0.25, 0.5),
    (359.99997, 3.0, -1.0, 3.0, -1.0),
    (1e-3, 1.0, 1e-3, 1.0, -1e-3),
]
GENERAL_ONLY_SETTINGS = [               # outside the strength-reduced kernel's domain
    (720.5, 1.0, 0.0, 1.0, 0.0),
    (-1e6, 1.1, 0.0, 0.9, 0.0),
    (float("nan"), 1.0, 0.0, 1.0, 0.0),
    (10.0, float("inf"), 0.0, float("nan"), 0.0),
    (float("inf"), 1.0, float("-inf"), 1.0, 0.0),
    (1e-35, 1.0, 0.0, 1.0, 0.0),
]


def _device_filter(vfx, host_frame, w, h, stride, fmt, settings, variant=0, batch=False, nontemporal=False):
    buf = vfx.DeviceBuffer(host_frame.nbytes).upload(host_frame)
    vfx.check(vfx.lib().mvfx_thread_set_options(vfx.options(variant=variant, nontemporal=nontemporal).word))
    try:
        s = vfx.HsvFilterSettings(*settings)
        if batch:
            vfx.hsvfilter_device_batch([buf.ptr], w, h, stride, fmt, s)
        else:
            vfx.hsvfilter_device(buf.ptr, w, h, stride, fmt, s)
        vfx.check(vfx.lib().mvfx_stream_synchronize(None))
    finally:
        vfx.lib().mvfx_thread_set_options(vfx.options(variant=0).word)
    return buf.download().reshape(host_frame.shape)


@pytest.fixture(scope="module")
def exhaustive():
    return frames.exhaustive_rgbx()


@pytest.mark.parametrize("settings", FILTER_SETTINGS)
@pytest.mark.parametrize("variant", [2, 1], ids=["fast", "general"])
def test_hsvfilter_exhaustive_rgba(gpu, exhaustive, settings, variant):
    """All 2^24 triples, RGBA, both kernel variants, bit-exact vs oracle; alpha untouched."""
    expect = exhaustive.copy()
    assert orc.hsvfilter(expect, 4096, 4096 * 4, "RGBA", settings) == 0
    got = _device_filter(gpu, exhaustive, 4096, 4096, 4096 * 4, "RGBA", settings, variant)
    bad = np.count_nonzero(got != expect)
    assert bad == 0, f"{bad} bytes differ for settings {settings}"


@pytest.mark.parametrize("settings", GENERAL_ONLY_SETTINGS)
def test_hsvfilter_exhaustive_general_domain(gpu, exhaustive, settings):
    """NaN / inf / huge / denormal-range settings: auto mode must pick the literal kernel."""
    expect = exhaustive.copy()
    assert orc.hsvfilter(expect, 4096, 4096 * 4, "RGBA", settings) == 0
    got = _device_filter(gpu, exhaustive, 4096, 4096, 4096 * 4, "RGBA", settings, 0)
    assert np.array_equal(got, expect)
    # and the fast kernel must refuse rather than silently produce something
    buf = gpu.DeviceBuffer(16)
    gpu.lib().mvfx_thread_set_options(gpu.options(variant=2).word)
    try:
        f = gpu.make_frame(buf.ptr, 2, 2, 8, "RGBA")
        rc = gpu.lib().mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(gpu.HsvFilterSettings(*settings)), None)
        assert rc == gpu.ERR_INVALID_ARGUMENT
    finally:
        gpu.lib().mvfx_thread_set_options(gpu.options(variant=0).word)


@pytest.mark.parametrize("variant", [2, 1], ids=["fast", "general"])
def test_from_rgb_f32_exhaustive(gpu, exhaustive, variant):
    """f32 HSV of every RGB triple: bit-identical floats (stronger than the 1-ULP bar)."""
    vfx = gpu
    src = vfx.DeviceBuffer(exhaustive.nbytes).upload(exhaustive)
    out = vfx.DeviceBuffer((1 << 24) * 3 * 4)
    vfx.lib().mvfx_thread_set_options(vfx.options(variant=variant).word)
    try:
        f = vfx.make_frame(src.ptr, 4096, 4096, 4096 * 4, "RGBx")
        vfx.check(vfx.lib().mvfx_hsv_from_frame(ctypes.byref(f), ctypes.c_void_p(out.ptr), None))
        vfx.check(vfx.lib().mvfx_stream_synchronize(None))
    finally:
        vfx.lib().mvfx_thread_set_options(vfx.options(variant=0).word)
    got = out.download(dtype=np.float32).reshape(-1, 3)
    expect = orc.hsv_from_rgbx(exhaustive)
    diff = got.view(np.uint32) != expect.view(np.uint32)
    # +0.0 vs -0.0 would be a bit difference without being a value difference
    diff &= ~((got == 0) & (expect == 0))
    assert np.count_nonzero(diff) == 0, f"{np.count_nonzero(diff)} f32 values differ"


ALL_FILTER_FORMATS = ["RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR", "RGB", "BGR"]


@pytest.mark.parametrize("fmt", ALL_FILTER_FORMATS)
@pytest.mark.parametrize("geom", [(64, 48, 0), (641, 37, 0), (67, 33, 12), (1, 1, 0), (3, 5, 4), (1918, 9, 0)],
                         ids=lambda g: f"{g[0]}x{g[1]}+pad{g[2]}")
def test_hsvfilter_formats_and_strides(gpu, fmt, geom):
    """Every format, odd widths, row padding (must stay untouched), host entry point."""
    w, h, pad = geom
    bpp = 3 if fmt in ("RGB", "BGR") else 4
    stride = (w * bpp + 3) // 4 * 4 + pad
    if (stride * h) % bpp != 0:
        pytest.skip("reference asserts on this size (covered by test_reference_panic_sizes)")
    frame = frames.random_frame(0x5EED0100 + w * 131 + h, w, h, bpp, stride)
    for settings in (BENCH_SETTINGS, (0.0, 1.0, 0.0, 1.0, 0.0), (720.5, 0.7, 0.1, 1.2, -0.1)):
        expect = frame.copy()
        assert orc.hsvfilter(expect, w, stride, fmt, settings) == 0
        got = frame.copy()
        gpu.hsvfilter_host(got.reshape(-1), w, h, stride, fmt, gpu.HsvFilterSettings(*settings))
        assert np.array_equal(got, expect), f"{fmt} {geom} {settings}"


@pytest.mark.parametrize("fmt", ["RGBA", "BGR"])
def test_hsvfilter_unaligned_base(gpu, fmt):
    """Plane pointer not 16-byte (or even 4-byte) aligned: dword / byte kernels, same result."""
    w, h = 124, 17
    bpp = 3 if fmt == "BGR" else 4
    stride = w * bpp  # 496 / 372: multiple of 4, and stride*h is a multiple of bpp
    frame = frames.random_frame(0x5EED0777, w, h, bpp, stride)
    expect = frame.copy()
    assert orc.hsvfilter(expect, w, stride, fmt, BENCH_SETTINGS) == 0
    for offset in (4, 1):
        buf = gpu.DeviceBuffer(frame.nbytes + 64)
        host = np.zeros(frame.nbytes + 64, dtype=np.uint8)
        host[offset:offset + frame.nbytes] = frame.reshape(-1)
        buf.upload(host)
        gpu.hsvfilter_device(buf.ptr + offset, w, h, stride, fmt, gpu.HsvFilterSettings(*BENCH_SETTINGS))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        out = buf.download()
        assert np.array_equal(out[offset:offset + frame.nbytes], expect.reshape(-1))
        assert not out[:offset].any() and not out[offset + frame.nbytes:].any()


@pytest.mark.parametrize("case", [("RGB", 5, 4, 15), ("BGR", 7, 3, 23), ("RGBA", 3, 5, 13), ("xRGB", 9, 2, 38)],
                         ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}_stride{c[3]}")
def test_hsvfilter_strides_that_are_not_multiples_of_four(gpu, case):
    """A GstVideoMeta may carry any stride: rows that are not dword aligned take the byte kernels."""
    fmt, w, h, stride = case
    bpp = 3 if fmt in ("RGB", "BGR") else 4
    if (stride * h) % bpp:
        h = bpp * h  # keep the reference's plane-size assert satisfied
    frame = frames.random_frame(0x5EED0999, w, h, bpp, stride)
    expect = frame.copy()
    assert orc.hsvfilter(expect, w, stride, fmt, BENCH_SETTINGS) == 0
    got = frame.copy()
    gpu.hsvfilter_host(got.reshape(-1), w, h, stride, fmt, gpu.HsvFilterSettings(*BENCH_SETTINGS))
    assert np.array_equal(got, expect)


def test_hsvfilter_8k_full_size(gpu):
    """largest BASELINE shape (7680x4320 RGBA, 132.7 MB): full-frame compare on uniform-random data"""
    w, h = 7680, 4320
    frame = frames.random_frame(0x5EED0002, w, h)
    expect = frame.copy()
    assert orc.hsvfilter(expect, w, w * 4, "RGBA", (-77.0, 0.8, 0.1, 1.1, -0.03)) == 0
    got = _device_filter(gpu, frame, w, h, w * 4, "RGBA", (-77.0, 0.8, 0.1, 1.1, -0.03), 0)
    assert np.array_equal(got, expect)


def test_hsvfilter_batch_matches_single(gpu):
    """The batched entry point == N single-frame calls (33 frames crosses the 32-frame launch split)."""
    w, h = 256, 64
    n = 33
    host = [frames.random_frame(0x5EED0100 + k, w, h) for k in range(n)]
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in host]
    gpu.hsvfilter_device_batch([b.ptr for b in bufs], w, h, w * 4, "BGRA", gpu.HsvFilterSettings(*BENCH_SETTINGS))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for k in range(n):
        expect = host[k].copy()
        orc.hsvfilter(expect, w, w * 4, "BGRA", BENCH_SETTINGS)
        assert np.array_equal(bufs[k].download().reshape(h, w * 4), expect), f"frame {k}"


@pytest.mark.parametrize("w,h", [(3840, 2160), (1021, 7), (8, 1), (2047, 3)])
def test_hsvfilter_streaming_policy_same_bytes(gpu, w, h):
    """MVFX_OPT_NONTEMPORAL (write-through stores since round 6, csrc/device_store.hpp) changes no byte; sizes with partial tiles / 1-3 pixel tails
    included.  (Until round 6 the helper reset the option word before the call: the option was not what this test ran.)"""
    frame = frames.random_frame(0x5EED0400 + w, w, h)
    expect = frame.copy()
    assert orc.hsvfilter(expect, w, w * 4, "RGBA", BENCH_SETTINGS) == 0
    for batch in (False, True):
        got = _device_filter(gpu, frame, w, h, w * 4, "RGBA", BENCH_SETTINGS, 0, batch=batch, nontemporal=True)
        assert np.array_equal(got, expect), batch


def test_hsvfilter_4k_full_size(gpu):
    """BASELINE config: 3840x2160 RGBA, uniform random + smpte-like, full-frame compare."""
    w, h = 3840, 2160
    for frame in (frames.random_frame(0x5EED0001, w, h), frames.smpte_like(w, h)):
        expect = frame.copy()
        assert orc.hsvfilter(expect, w, w * 4, "RGBA", BENCH_SETTINGS) == 0
        got = _device_filter(gpu, frame, w, h, w * 4, "RGBA", BENCH_SETTINGS, 0, batch=True)
        assert np.array_equal(got, expect)


def test_hsvfilter_4k_videotestsrc_frames(gpu):
    """The frames bench.py times (videotestsrc pattern=smpte, consecutive frames = different snow): full-frame compare."""
    w, h = 3840, 2160
    vts, _ = frames.videotestsrc_smpte(w, h, 2)
    for frame in vts:
        expect = frame.copy()
        assert orc.hsvfilter(expect, w, w * 4, "RGBA", BENCH_SETTINGS) == 0
        got = _device_filter(gpu, frame, w, h, w * 4, "RGBA", BENCH_SETTINGS, 0, batch=True)
        assert np.array_equal(got, expect)


def test_reference_panic_sizes(gpu):
    """642x481 RGB: stride 1928 * 481 % 3 == 2 -> the reference's assert_eq! fires."""
    w, h, stride = 642, 481, 1928
    frame = frames.random_frame(1, w, h, 3, stride)
    assert orc.hsvfilter(frame.copy(), w, stride, "RGB", BENCH_SETTINGS) == -1
    f = gpu.make_frame(frame.ctypes.data, w, h, stride, "RGB")
    rc = gpu.lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(gpu.HsvFilterSettings(*BENCH_SETTINGS)))
    assert rc == gpu.ERR_REFERENCE_PANIC
    assert "hsvfilter/imp.rs:92" in gpu.last_error()


def test_hsvfilter_rejects_bad_arguments(gpu):
    s = gpu.HsvFilterSettings.default()
    buf = np.zeros(128, dtype=np.uint8)
    f = gpu.make_frame(buf.ctypes.data, 4, 4, 32, "RGBA64_LE")
    assert gpu.lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(s)) == gpu.ERR_UNSUPPORTED_FORMAT
    f = gpu.make_frame(buf.ctypes.data, 8, 4, 16, "RGBA")  # row longer than stride
    assert gpu.lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(s)) == gpu.ERR_INVALID_ARGUMENT
    f = gpu.make_frame(0, 4, 4, 16, "RGBA")
    assert gpu.lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(s)) == gpu.ERR_INVALID_ARGUMENT
    f = gpu.make_frame(buf.ctypes.data, 0, 0, 16, "RGBA")  # empty frame is a no-op
    assert gpu.lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(s)) == gpu.OK


# ---------------------------------------------------------------- hsvdetector

DETECT_SETTINGS = [
    (540.0, 0.0, 0.5, 0.5, 0.5, 0.5),           # ref_off = -360 (domain edge), hue_var = 0: exact-equality hits only
    (-180.0, 180.0, 0.25, 0.0, 1.0, 0.0),       # ref_off = +360 (other edge), zero sat/val tolerance
    (0.0, 10.0, 0.0, 0.15, 0.0, 0.3),           # defaults
    (120.0, 40.0, 0.6, 0.4, 0.6, 0.4),          # bench settings
    (350.0, 25.0, 0.5, 0.5, 0.5, 0.5),          # wraps through 0
    (-200.0, 180.0, 1.0, 1.0, 1.0, 1.0),        # everything matches
    (1e6, 90.0, 0.5, 0.3, 0.5, 0.3),            # huge hue_ref -> general fmod
    (float("nan"), 10.0, 0.0, 0.15, 0.0, 0.3),  # nothing matches
]
DET_IN = ["RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR"]
DET_OUT = ["RGBA", "ARGB", "BGRA", "ABGR"]


@pytest.mark.parametrize("in_fmt", DET_IN)
@pytest.mark.parametrize("out_fmt", DET_OUT)
def test_hsvdetector_all_24_pairs(gpu, in_fmt, out_fmt):
    for (w, h, pad) in ((64, 16, 0), (45, 7, 8)):
        bpp = 3 if in_fmt in ("RGB", "BGR") else 4
        in_stride = (w * bpp + 3) // 4 * 4 + pad
        while (in_stride * h) % bpp:
            in_stride += 4
        out_stride = w * 4 + pad
        src = frames.random_frame(0x5EED0200 + w, w, h, bpp, in_stride)
        for settings in DETECT_SETTINGS[:5]:
            sentinel = np.full((h, out_stride), 0xA5, dtype=np.uint8)
            expect = sentinel.copy()
            assert orc.hsvdetector(src, in_stride, in_fmt, expect, out_stride, out_fmt, w, settings) == 0
            got = sentinel.copy()
            gpu.hsvdetector_host(src.reshape(-1), in_stride, in_fmt, got.reshape(-1), out_stride, out_fmt, w, h,
                                 gpu.HsvDetectorSettings(*settings))
            assert np.array_equal(got, expect), f"{in_fmt}->{out_fmt} {w}x{h} {settings}"


@pytest.mark.parametrize("settings", DETECT_SETTINGS)
@pytest.mark.parametrize("variant", [0, 1, 2], ids=["auto", "general", "fast"])
def test_hsvdetector_exhaustive(gpu, exhaustive, settings, variant):
    """All 2^24 triples RGBx -> RGBA: colour copied, alpha 0/255 exactly as the reference."""
    if variant == 2 and not (np.isfinite(settings[0]) and abs(180.0 - settings[0]) <= 360.0):
        pytest.skip("outside the strength-reduced hue test's domain (auto mode covers it with the literal test)")
    expect = np.empty_like(exhaustive)
    assert orc.hsvdetector(exhaustive, 4096 * 4, "RGBx", expect, 4096 * 4, "RGBA", 4096, settings) == 0
    src = gpu.DeviceBuffer(exhaustive.nbytes).upload(exhaustive)
    dst = gpu.DeviceBuffer(exhaustive.nbytes)
    fi = gpu.make_frame(src.ptr, 4096, 4096, 4096 * 4, "RGBx")
    fo = gpu.make_frame(dst.ptr, 4096, 4096, 4096 * 4, "RGBA")
    gpu.lib().mvfx_thread_set_options(gpu.options(variant=variant).word)
    try:
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo),
                                                             ctypes.byref(gpu.HsvDetectorSettings(*settings)), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    finally:
        gpu.lib().mvfx_thread_set_options(gpu.options(variant=0).word)
    got = dst.download().reshape(exhaustive.shape)
    assert np.array_equal(got, expect)
    if not np.isnan(settings[0]) and settings[1] > 0 and settings[3] > 0:
        assert got[:, 3::4].max() == 255 or settings[0] > 1e5  # (a hue_ref far below -180 never matches: one +360 only)


@pytest.mark.parametrize("in_fmt,bpp,out_fmt,w,h,stride_pad", [("RGBx", 4, "RGBA", 256, 64, 0), ("BGR", 3, "ABGR", 101, 37, 6),
                                                               ("xBGR", 4, "BGRA", 320, 48, 16)])
def test_hsvdetector_batch_matches_single(gpu, in_fmt, bpp, out_fmt, w, h, stride_pad):
    """mvfx_hsvdetector_transform_frames == N single-frame calls (33 pairs cross the 32-pair launch split);
    padded / unaligned layouts take the byte path, the row padding of the outputs stays untouched."""
    n = 33
    istride, ostride = w * bpp + stride_pad, w * 4 + stride_pad
    ins = [frames.random_frame(0x5EED0200 + k, w, h, bpp, istride) for k in range(n)]
    outs0 = [frames.random_frame(0x5EED0300 + k, w, h, 4, ostride) for k in range(n)]
    din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in ins]
    dout = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in outs0]
    fi = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, istride, in_fmt) for b in din])
    fo = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, ostride, out_fmt) for b in dout])
    s = DETECT_SETTINGS[3]
    gpu.check(gpu.lib().mvfx_hsvdetector_transform_frames(fi, fo, n, ctypes.byref(gpu.HsvDetectorSettings(*s)), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for k in range(n):
        expect = outs0[k].copy()
        assert orc.hsvdetector(ins[k], istride, in_fmt, expect, ostride, out_fmt, w, s) == 0
        assert np.array_equal(dout[k].download().reshape(h, ostride), expect), f"pair {k}"
    # mixed geometry inside one batch is refused
    fo[1].width = w - 1
    assert gpu.lib().mvfx_hsvdetector_transform_frames(fi, fo, 2, ctypes.byref(gpu.HsvDetectorSettings(*s)), None) == gpu.ERR_INVALID_ARGUMENT
    assert gpu.lib().mvfx_hsvdetector_transform_frames(fi, fo, 0, ctypes.byref(gpu.HsvDetectorSettings(*s)), None) == gpu.ERR_INVALID_ARGUMENT


def test_hsvdetector_1080p_after_hsvfilter(gpu):
    """BASELINE config 2: hsvfilter (RGBx, in place) then hsvdetector RGBx->RGBA at 1920x1080."""
    w, h = 1920, 1080
    frame = frames.random_frame(0x5EED0002, w, h)
    expect_mid = frame.copy()
    orc.hsvfilter(expect_mid, w, w * 4, "RGBx", BENCH_SETTINGS)
    expect = np.empty_like(frame)
    orc.hsvdetector(expect_mid, w * 4, "RGBx", expect, w * 4, "RGBA", w, DETECT_SETTINGS[3])
    src = gpu.DeviceBuffer(frame.nbytes).upload(frame)
    dst = gpu.DeviceBuffer(frame.nbytes)
    gpu.hsvfilter_device(src.ptr, w, h, w * 4, "RGBx", gpu.HsvFilterSettings(*BENCH_SETTINGS))
    fi = gpu.make_frame(src.ptr, w, h, w * 4, "RGBx")
    fo = gpu.make_frame(dst.ptr, w, h, w * 4, "RGBA")
    gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo),
                                                         ctypes.byref(gpu.HsvDetectorSettings(*DETECT_SETTINGS[3])), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    assert np.array_equal(src.download().reshape(h, w * 4), expect_mid)
    assert np.array_equal(dst.download().reshape(h, w * 4), expect)


def test_hsvdetector_size_mismatch(gpu):
    a = np.zeros(64, dtype=np.uint8)
    fi = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBx")
    fo = gpu.make_frame(a.ctypes.data, 4, 2, 16, "RGBA")
    rc = gpu.lib().mvfx_hsvdetector_transform_frame_host(ctypes.byref(fi), ctypes.byref(fo),
                                                         ctypes.byref(gpu.HsvDetectorSettings.default()))
    assert rc == gpu.ERR_NOT_NEGOTIATED
    fo = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBx")  # RGBx is not an output format
    rc = gpu.lib().mvfx_hsvdetector_transform_frame_host(ctypes.byref(fi), ctypes.byref(fo),
                                                         ctypes.byref(gpu.HsvDetectorSettings.default()))
    assert rc == gpu.ERR_UNSUPPORTED_FORMAT


def test_typed_loads_and_valu_paths_agree_on_all_triples(gpu):
    """hsvfilter's u8/255 through typed buffer loads (texture-unit UNORM8 conversion, the default) and through the VALU
    kernels: both bit-exact against the oracle on all 2^24 (R,G,B) triples, every 4-byte layout, both hue-shift signs"""
    ex = frames.exhaustive_rgbx()
    try:
        for fmt in ("RGBA", "xBGR", "BGRx", "ARGB"):
            for st in ((90.0, 1.25, -0.05, 0.9, 0.02), (-45.0, 0.8, 0.1, 1.1, -0.03)):
                want = ex.copy()
                assert orc.hsvfilter(want, 4096, 4096 * 4, fmt, st) == 0
                for typed in (1, 0):
                    gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(typed=bool(typed)).word))
                    buf = gpu.DeviceBuffer(ex.nbytes).upload(ex)
                    f = gpu.make_frame(buf.ptr, 4096, 4096, 4096 * 4, fmt)
                    s = gpu.HsvFilterSettings(*st)
                    gpu.check(gpu.lib().mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(s), None))
                    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
                    got = buf.download().reshape(want.shape)
                    assert np.array_equal(got, want), (fmt, st, typed, int(np.count_nonzero(got != want)))
    finally:
        gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(typed=bool(1)).word))


def test_typed_loads_hsvdetector_all_triples(gpu):
    """the detector's typed-load path (default) and its VALU path against the oracle on all 2^24 triples, 4 input x 2 output layouts"""
    ex = frames.exhaustive_rgbx()
    st = (120.0, 40.0, 0.6, 0.4, 0.6, 0.4)
    try:
        for in_fmt, out_fmt in (("RGBx", "RGBA"), ("xBGR", "ARGB"), ("BGRx", "BGRA"), ("xRGB", "ABGR")):
            want = np.empty_like(ex)
            assert orc.hsvdetector(ex, 4096 * 4, in_fmt, want, 4096 * 4, out_fmt, 4096, st) == 0
            for typed in (1, 0):
                gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(typed=bool(typed)).word))
                src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
                dst = gpu.DeviceBuffer(ex.nbytes)
                fi = gpu.make_frame(src.ptr, 4096, 4096, 4096 * 4, in_fmt)
                fo = gpu.make_frame(dst.ptr, 4096, 4096, 4096 * 4, out_fmt)
                s = gpu.HsvDetectorSettings(*st)
                gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(s), None))
                gpu.check(gpu.lib().mvfx_stream_synchronize(None))
                got = dst.download().reshape(want.shape)
                assert np.array_equal(got, want), (in_fmt, out_fmt, typed, int(np.count_nonzero(got != want)))
    finally:
        gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(typed=bool(1)).word))


# ---------------------------------------------------------------- RGB / BGR through the typed 3-byte kernels (round 5's default path)
# hsvfilter3_typed_kernel / hsvdetector3_typed_kernel take every 3-byte frame whose width is a multiple of four (hsv_kernels.hip,
# hsvfilter_impl / hsvdetector_impl).  Their proofs, like the 4-byte kernels': all 2^24 colours against the oracle.

def exhaustive_rgb3() -> np.ndarray:
    """4096 x 4096 packed 3-byte frame (stride 12 288) holding all 2^24 (c0, c1, c2) triples, in the pixel order of frames.exhaustive_rgbx()"""
    return np.ascontiguousarray(frames.exhaustive_rgbx().reshape(-1, 4)[:, :3]).reshape(4096, 4096 * 3)


@pytest.fixture(scope="module")
def exhaustive3():
    return exhaustive_rgb3()


def test_typed_unorm8_loads_at_unaligned_addresses(gpu):
    """tools/probes/typed_unaligned.hip as a test: buffer_load_format_xyz of 8_8_8_8 UNORM at byte addresses 4 k + 0 / 1 / 2 / 3, through
    both descriptors and the immediate offsets 0 / 3 / 6 / 8 of the 3-byte kernels: RN(byte / 255) for all 256 values, RGB and BGR order"""
    checked, bad = ctypes.c_uint32(), ctypes.c_uint32(0xFFFFFFFF)
    gpu.check(gpu.lib().mvfx_selftest_typed_unorm8(ctypes.byref(checked), ctypes.byref(bad)))
    assert checked.value == 2 * 1024 * 12 and bad.value == 0, (checked.value, bad.value)


RGB3_FILTER_SETTINGS = [BENCH_SETTINGS, (-123.4, 0.5, 0.3, 1.7, -0.2), (0.0, 1.0, 0.0, 1.0, 0.0)]


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("settings", RGB3_FILTER_SETTINGS, ids=["bench", "negative-shift", "defaults"])
@pytest.mark.parametrize("nontemporal", [False, True], ids=["cached", "nt"])
def test_hsvfilter_rgb3_typed_exhaustive(gpu, exhaustive3, fmt, settings, nontemporal):
    """All 2^24 triples packed as RGB / BGR through mvfx_hsvfilter_transform_frame_ip: the whole 4096 x 4096 frame takes the two-groups-per-lane
    instantiation (TILE = 2: >= 3840 x 2160 / 4 groups), its four 4096 x 1024 bands TILE = 1; positive and negative hue-shift (kFast / kFastNeg);
    cached and non-temporal; bit-exact against the oracle."""
    W, H, stride = 4096, 4096, 4096 * 3
    expect = exhaustive3.copy()
    assert orc.hsvfilter(expect, W, stride, fmt, settings) == 0
    s = gpu.HsvFilterSettings(*settings)
    gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(nontemporal=nontemporal).word))
    try:
        buf = gpu.DeviceBuffer(exhaustive3.nbytes).upload(exhaustive3)
        gpu.hsvfilter_device(buf.ptr, W, H, stride, fmt, s)
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        got = buf.download().reshape(H, stride)
        assert np.array_equal(got, expect), f"whole frame: {int(np.count_nonzero(got != expect))} bytes differ"
        buf.upload(exhaustive3)
        for band in range(4):
            gpu.hsvfilter_device(buf.ptr + band * 1024 * stride, W, 1024, stride, fmt, s)
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        got = buf.download().reshape(H, stride)
        assert np.array_equal(got, expect), f"bands: {int(np.count_nonzero(got != expect))} bytes differ"
    finally:
        gpu.check(gpu.lib().mvfx_thread_set_options(0))


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
def test_hsvfilter_rgb3_typed_and_valu_paths_agree(gpu, exhaustive3, fmt):
    """the VALU kernel hsvfilter3_kernel (MVFX_OPT_HSV_VALU_UNORM; also what widths that are not multiples of four take) on the same
    2^24 triples, and a 256 x 64 frame (TILE = 1, rows = 1 flat walk) through both"""
    W, H, stride = 4096, 4096, 4096 * 3
    for settings in (BENCH_SETTINGS, (-45.0, 0.8, 0.1, 1.1, -0.03)):
        expect = exhaustive3.copy()
        assert orc.hsvfilter(expect, W, stride, fmt, settings) == 0
        small = frames.random_frame(0x5EED0A00, 256, 64, 3)
        small_expect = small.copy()
        assert orc.hsvfilter(small_expect, 256, 256 * 3, fmt, settings) == 0
        try:
            for typed in (False, True):
                gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(typed=typed).word))
                buf = gpu.DeviceBuffer(exhaustive3.nbytes).upload(exhaustive3)
                gpu.hsvfilter_device(buf.ptr, W, H, stride, fmt, gpu.HsvFilterSettings(*settings))
                sb = gpu.DeviceBuffer(small.nbytes).upload(small)
                gpu.hsvfilter_device(sb.ptr, 256, 64, 256 * 3, fmt, gpu.HsvFilterSettings(*settings))
                gpu.check(gpu.lib().mvfx_stream_synchronize(None))
                assert np.array_equal(buf.download().reshape(H, stride), expect), (fmt, settings, typed)
                assert np.array_equal(sb.download().reshape(64, 256 * 3), small_expect), (fmt, settings, typed)
        finally:
            gpu.check(gpu.lib().mvfx_thread_set_options(0))


RGB3_DETECT_SETTINGS = [DETECT_SETTINGS[3], DETECT_SETTINGS[0], DETECT_SETTINGS[1], DETECT_SETTINGS[4]]


@pytest.mark.parametrize("in_fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("out_fmt", DET_OUT)
def test_hsvdetector_rgb3_typed_exhaustive(gpu, exhaustive3, in_fmt, out_fmt):
    """All 2^24 triples packed as RGB / BGR -> every output layout through mvfx_hsvdetector_transform_frame (hsvdetector3_typed_kernel);
    bench settings + both edges of the hue test's domain + a wrap through 0"""
    W, H, istride, ostride = 4096, 4096, 4096 * 3, 4096 * 4
    src = gpu.DeviceBuffer(exhaustive3.nbytes).upload(exhaustive3)
    dst = gpu.DeviceBuffer(H * ostride)
    fi = gpu.make_frame(src.ptr, W, H, istride, in_fmt)
    fo = gpu.make_frame(dst.ptr, W, H, ostride, out_fmt)
    for settings in RGB3_DETECT_SETTINGS:
        expect = np.empty((H, ostride), dtype=np.uint8)
        assert orc.hsvdetector(exhaustive3, istride, in_fmt, expect, ostride, out_fmt, W, settings) == 0
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(gpu.HsvDetectorSettings(*settings)), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        got = dst.download().reshape(H, ostride)
        assert np.array_equal(got, expect), (in_fmt, out_fmt, settings, int(np.count_nonzero(got != expect)))


def _vts_rgb3(n):
    """n consecutive videotestsrc pattern=smpte 3840 x 2160 frames packed to RGB (what bench.py's hsvfilter_rgb / hsvdetector_rgb legs time)"""
    w, h = 3840, 2160
    vts, _ = frames.videotestsrc_smpte(w, h, n)
    return [np.ascontiguousarray(vts[k].reshape(h, w, 4)[..., :3]).reshape(h, w * 3) for k in range(n)]


@pytest.mark.parametrize("nontemporal", [True, False], ids=["nt", "cached"])
def test_hsv_rgb3_bench_legs_full_compare(gpu, nontemporal):
    """bench.py's two RGB legs themselves: 16 x 3840 x 2160 RGB videotestsrc frames through mvfx_hsvfilter_transform_frames_ip (in place) and
    mvfx_hsvdetector_transform_frames (-> RGBA), every byte of every frame against the oracle"""
    w, h, n = 3840, 2160, 16
    stride3 = w * 3
    host = _vts_rgb3(n)
    din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in host]
    dout = [gpu.DeviceBuffer(h * w * 4) for _ in range(n)]
    fi = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, stride3, "RGB") for b in din])
    fo = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in dout])
    ds = DETECT_SETTINGS[3]
    gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(nontemporal=nontemporal).word))
    try:
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_frames(fi, fo, n, ctypes.byref(gpu.HsvDetectorSettings(*ds)), None))
        gpu.check(gpu.lib().mvfx_hsvfilter_transform_frames_ip(fi, n, ctypes.byref(gpu.HsvFilterSettings(*BENCH_SETTINGS)), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    finally:
        gpu.check(gpu.lib().mvfx_thread_set_options(0))
    for k in range(n):
        expect_det = np.empty((h, w * 4), dtype=np.uint8)
        assert orc.hsvdetector(host[k], stride3, "RGB", expect_det, w * 4, "RGBA", w, ds) == 0
        assert np.array_equal(dout[k].download().reshape(h, w * 4), expect_det), f"detector frame {k}"
        expect = host[k].copy()
        assert orc.hsvfilter(expect, w, stride3, "RGB", BENCH_SETTINGS) == 0
        assert np.array_equal(din[k].download().reshape(h, stride3), expect), f"filter frame {k}"


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("geom", [(256, 64, 16), (1920, 1080, 0), (3840, 2160, 32), (8, 3, 12), (4, 1, 0)], ids=lambda g: f"{g[0]}x{g[1]}+pad{g[2]}")
def test_hsv_rgb3_typed_never_touches_a_byte_outside_the_frame(gpu, fmt, geom):
    """The typed 3-byte kernels lean on the buffer descriptor's bounds (num_records = stride x height) and read a lane's fourth pixel through a
    second descriptor: the frame sits in the middle of a larger allocation filled with a sentinel, the row padding is non-zero random bytes, and
    (a) the result equals the oracle's, (b) padding and sentinel are untouched, (c) the detector's output allocation likewise -- also with the
    frame ending exactly at the last byte of its allocation (the last lane's loads end at num_records)."""
    w, h, pad = geom
    stride = w * 3 + pad
    while (stride * h) % 3 or stride % 4:
        stride += 4
    frame = frames.random_frame(0x5EED0B00 + w, w, h, 3, stride)
    expect = frame.copy()
    assert orc.hsvfilter(expect, w, stride, fmt, BENCH_SETTINGS) == 0
    ostride = w * 4 + 16
    expect_det = np.full((h, ostride), 0xC3, dtype=np.uint8)
    assert orc.hsvdetector(frame, stride, fmt, expect_det, ostride, "ARGB", w, DETECT_SETTINGS[3]) == 0
    for lead, trail in ((256, 256), (4096, 0)):
        total = lead + frame.nbytes + trail
        host = np.full(total, 0x5A, dtype=np.uint8)
        host[lead:lead + frame.nbytes] = frame.reshape(-1)
        buf = gpu.DeviceBuffer(total).upload(host)
        ototal = lead + h * ostride + trail
        obuf = gpu.DeviceBuffer(ototal).upload(np.full(ototal, 0xC3, dtype=np.uint8))
        fi = gpu.make_frame(buf.ptr + lead, w, h, stride, fmt)
        fo = gpu.make_frame(obuf.ptr + lead, w, h, ostride, "ARGB")
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(gpu.HsvDetectorSettings(*DETECT_SETTINGS[3])), None))
        gpu.hsvfilter_device(buf.ptr + lead, w, h, stride, fmt, gpu.HsvFilterSettings(*BENCH_SETTINGS))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        out = buf.download()
        assert np.array_equal(out[lead:lead + frame.nbytes].reshape(h, stride), expect)
        assert (out[:lead] == 0x5A).all() and (out[lead + frame.nbytes:] == 0x5A).all()
        oout = obuf.download()
        assert np.array_equal(oout[lead:lead + h * ostride].reshape(h, ostride), expect_det)
        assert (oout[:lead] == 0xC3).all() and (oout[lead + h * ostride:] == 0xC3).all()


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("w,h", [(854, 480), (1366, 768), (2001, 1501), (5, 3), (2, 4), (7, 1)])
def test_hsvfilter_rgb3_widths_that_are_not_multiples_of_four(gpu, fmt, w, h):
    """Round 6: 3-byte frames whose width is not a multiple of four take the typed kernel too -- whole groups of four pixels as ever,
    a row's last 1..3 pixels by one lane each (a typed load of the pixel's own bytes + the first padding byte, three byte stores).  Device entry,
    GStreamer's stride (round_up_4(3 w)), random padding bytes that must stay; both signs of hue-shift; also as a 3-frame batch."""
    stride = (3 * w + 3) // 4 * 4
    while (stride * h) % 3:
        stride += 4
    f = frames.random_frame(0x5EED0E50 + w, w, h, 3, stride)
    for settings in (BENCH_SETTINGS, (-45.0, 0.8, 0.1, 1.1, -0.03)):
        want = f.copy()
        assert orc.hsvfilter(want, w, stride, fmt, settings) == 0
        lead = 64
        host = np.full(lead + f.nbytes + 64, 0x5A, dtype=np.uint8)
        host[lead:lead + f.nbytes] = f.reshape(-1)
        buf = gpu.DeviceBuffer(host.nbytes).upload(host)
        gpu.hsvfilter_device(buf.ptr + lead, w, h, stride, fmt, gpu.HsvFilterSettings(*settings))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        out = buf.download()
        assert np.array_equal(out[lead:lead + f.nbytes].reshape(h, stride), want), (fmt, w, h, settings)
        assert (out[:lead] == 0x5A).all() and (out[lead + f.nbytes:] == 0x5A).all()
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for _ in range(3)]
    gpu.hsvfilter_device_batch([b.ptr for b in bufs], w, h, stride, fmt, gpu.HsvFilterSettings(*BENCH_SETTINGS))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    want = f.copy()
    assert orc.hsvfilter(want, w, stride, fmt, BENCH_SETTINGS) == 0
    for b in bufs:
        assert np.array_equal(b.download().reshape(h, stride), want)


# ---------------------------------------------------------------- MVFX_OPT_NONTEMPORAL = write-through stores (round 6, csrc/device_store.hpp)
# With the option the typed kernels store `global_store_dwordx4 / x3 ... sc0 sc1 nt` through inline asm (the hsvdetector kernels gained the
# instantiation in round 6).  New store instructions => their own proofs: all 2^24 colours, every kernel that has the form.

@pytest.mark.parametrize("fmt", ["RGBA", "xBGR"])
@pytest.mark.parametrize("settings", [BENCH_SETTINGS, (-123.4, 0.5, 0.3, 1.7, -0.2)], ids=["bench", "negative"])
def test_hsvfilter_streaming_stores_exhaustive(gpu, exhaustive, fmt, settings):
    want = exhaustive.copy()
    assert orc.hsvfilter(want, 4096, 4096 * 4, fmt, settings) == 0
    s = gpu.HsvFilterSettings(*settings)
    try:
        gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(nontemporal=True).word))
        for batch in (False, True):
            buf = gpu.DeviceBuffer(exhaustive.nbytes).upload(exhaustive)
            if batch:
                gpu.hsvfilter_device_batch([buf.ptr], 4096, 4096, 4096 * 4, fmt, s)
            else:
                gpu.hsvfilter_device(buf.ptr, 4096, 4096, 4096 * 4, fmt, s)
            gpu.check(gpu.lib().mvfx_stream_synchronize(None))
            got = buf.download().reshape(exhaustive.shape)
            assert np.array_equal(got, want), (batch, int(np.count_nonzero(got != want)))
    finally:
        gpu.check(gpu.lib().mvfx_thread_set_options(0))


@pytest.mark.parametrize("in_fmt,out_fmt", [("RGBx", "RGBA"), ("xBGR", "ARGB"), ("RGB", "BGRA"), ("BGR", "ABGR")])
def test_hsvdetector_streaming_stores_exhaustive(gpu, exhaustive, exhaustive3, in_fmt, out_fmt):
    """hsvdetector_typed_kernel<true> / hsvdetector3_typed_kernel<true>: all 2^24 colours, two settings, and a frame with row padding on the output
    (kept) whose last groups are partial tiles"""
    L = gpu.lib()
    three = in_fmt in ("RGB", "BGR")
    ex = exhaustive3 if three else exhaustive
    istride = 4096 * (3 if three else 4)
    src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
    dst = gpu.DeviceBuffer(4096 * 4096 * 4)
    fi, fo = gpu.make_frame(src.ptr, 4096, 4096, istride, in_fmt), gpu.make_frame(dst.ptr, 4096, 4096, 4096 * 4, out_fmt)
    try:
        gpu.check(L.mvfx_thread_set_options(gpu.options(nontemporal=True).word))
        for settings in (DETECT_SETTINGS[0], DETECT_SETTINGS[3]):
            want = np.empty((4096, 4096 * 4), np.uint8)
            assert orc.hsvdetector(ex, istride, in_fmt, want, 4096 * 4, out_fmt, 4096, settings) == 0
            gpu.check(L.mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(gpu.HsvDetectorSettings(*settings)), None))
            gpu.check(L.mvfx_stream_synchronize(None))
            got = dst.download().reshape(want.shape)
            assert np.array_equal(got, want), (settings, int(np.count_nonzero(got != want)))
        w, h, bpp = 1004, 37, 3 if three else 4
        istr, ostr = w * bpp + (12 if three else 16), w * 4 + 16  # (four pixels of padding on both sides: what the reference's row zip accepts)
        f = frames.random_frame(0x5EED1200 + bpp, w, h, bpp, istr)
        fill = np.full((h, ostr), 0x6B, np.uint8)
        want = fill.copy()
        assert orc.hsvdetector(f, istr, in_fmt, want, ostr, out_fmt, w, DETECT_SETTINGS[0]) == 0
        a, b = gpu.DeviceBuffer(f.nbytes).upload(f), gpu.DeviceBuffer(fill.nbytes).upload(fill)
        fa, fb = gpu.make_frame(a.ptr, w, h, istr, in_fmt), gpu.make_frame(b.ptr, w, h, ostr, out_fmt)
        gpu.check(L.mvfx_hsvdetector_transform_frame(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(gpu.HsvDetectorSettings(*DETECT_SETTINGS[0])), None))
        gpu.check(L.mvfx_stream_synchronize(None))
        assert np.array_equal(b.download().reshape(h, ostr), want)
    finally:
        gpu.check(L.mvfx_thread_set_options(0))
