"""GPU parity tests for colorlut: HIP path (C ABI) vs the oracle, bit-exact u8 / u16."""
import ctypes

import numpy as np
import pytest

from tests import cubes, frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu


def _cases():
    return {
        "identity17": cubes.identity_3d(17),
        "analytic9": cubes.analytic_3d(9),
        "analytic21": cubes.analytic_3d(21),   # largest cube that is staged in LDS
        "analytic33": cubes.analytic_3d(33),   # BASELINE config 3 (L2-resident)
        "analytic65": cubes.analytic_3d(65),   # largest cell-packed cube: the 32 x 16 block shape of the window kernel
        "analytic3": cubes.analytic_3d(3),     # smallest cube the window kernel takes (window = the whole cube)
        # 3-D with a DOMAIN: scaled / offset coordinates, channels that never reach the upper / lower cells
        "domain33": cubes.analytic_3d(33).replace("DOMAIN_MIN 0.0 0.0 0.0", "DOMAIN_MIN -0.25 0.0 0.1").replace("DOMAIN_MAX 1.0 1.0 1.0", "DOMAIN_MAX 1.5 1.0 0.9"),
        "curve1d_256": cubes.curve_1d(256),
        "curve1d_2": cubes.curve_1d(2),
        "curve1d_4096": cubes.curve_1d(4096),
        "curve1d_65536": cubes.curve_1d(65536),
        "curve1d_domain": cubes.curve_1d(64, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))),
        "nan_nodes": "LUT_3D_SIZE 2\n" + "nan 0.5 inf\n" * 4 + "0.25 -inf 2\n" * 4,
        "nan_domain": "LUT_1D_SIZE 2\nDOMAIN_MIN nan 0 0\n0 0.1 0.2\n1 0.9 0.8\n",
    }


@pytest.fixture(scope="module")
def luts(gpu):
    out = {}
    for name, text in _cases().items():
        o = orc.CubeLut(text)
        assert o.ok, o.error
        out[name] = (gpu.CubeLut(text), o)
    return out


@pytest.mark.parametrize("name", sorted(_cases()))
@pytest.mark.parametrize("fmt", ["RGBA", "RGBA64_LE", "RGBA64_BE"])
@pytest.mark.parametrize("placement", [0, 1, 3, 4], ids=["auto", "global", "cells", "literal"])
def test_colorlut_random_frames(gpu, luts, name, fmt, placement):
    """ragged width, row padding (kept), host entry point, every LUT placement / kernel family"""
    dev, o = luts[name]
    if placement == 3 and not dev.is_3d:
        pytest.skip("cell-packed layout is 3-D only")
    gpu.check(gpu.lib().mvfx_thread_set_options(gpu.options(placement=placement).word))
    try:
        for (w, h, pad) in ((257, 9, 16), (64, 8, 0), (1, 1, 0), (1023, 3, 0)):
            bpp = 8 if fmt != "RGBA" else 4
            stride = w * bpp + pad
            src = frames.random_frame(0x5EED0500 + w, w, h, bpp, stride)
            exp = np.full((h, stride), 0xC3, np.uint8)
            assert o.apply(src, stride, exp, stride, w, h, fmt) == 0
            got = np.full((h, stride), 0xC3, np.uint8)
            dev.apply_host(src.reshape(-1), stride, got.reshape(-1), stride, w, h, fmt)
            bad = np.count_nonzero(got != exp)
            assert bad == 0, f"{name} {fmt} {w}x{h}: {bad} bytes differ"
    finally:
        gpu.lib().mvfx_thread_set_options(gpu.options(placement=0).word)


@pytest.mark.parametrize("name", ["analytic33", "analytic21", "identity17", "curve1d_256", "nan_nodes", "nan_domain", "analytic65", "analytic9",
                                  "analytic3", "domain33"])
def test_colorlut_exhaustive_rgba8(gpu, luts, name):
    """all 2^24 RGB triples through the LUT (RGBA8), device entry point, automatic kernel choice"""
    dev, o = luts[name]
    ex = frames.exhaustive_rgbx()
    exp = np.empty_like(ex)
    assert o.apply(ex, 4096 * 4, exp, 4096 * 4, 4096, 4096, "RGBA") == 0
    src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
    dst = gpu.DeviceBuffer(ex.nbytes)
    dev.apply_device(src.ptr, 4096 * 4, dst.ptr, 4096 * 4, 4096, 4096, "RGBA")
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = dst.download().reshape(ex.shape)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("name", ["analytic33", "domain33", "analytic65", "curve1d_256", "nan_nodes", "nan_domain"])
def test_colorlut_baked_table(gpu, luts, name):
    """placement 6: the LUT baked into a table of all 2^24 colours (built on first use by the interpolating kernels) and applied with one
    gather per pixel.  All 2^24 triples with a non-zero alpha byte, then a batch of padded frames: the oracle's bytes, alpha copied,
    padding untouched; frames the kernel does not take (unaligned rows) are refused, RGBA64 falls through to the automatic choice."""
    dev, o = luts[name]
    L = gpu.lib()
    gpu.check(L.mvfx_thread_set_options(gpu.options(placement=6).word))
    try:
        ex = frames.exhaustive_rgbx()  # byte 3 is a hash of the pixel index: it must come through untouched
        exp = np.empty_like(ex)
        assert o.apply(ex, 4096 * 4, exp, 4096 * 4, 4096, 4096, "RGBA") == 0
        src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
        dst = gpu.DeviceBuffer(ex.nbytes)
        dev.apply_device(src.ptr, 4096 * 4, dst.ptr, 4096 * 4, 4096, 4096, "RGBA")
        gpu.check(L.mvfx_stream_synchronize(None))
        assert np.array_equal(dst.download().reshape(ex.shape), exp)
        # a batch of frames with row padding (stride a multiple of 16, width a multiple of 4)
        n, w, h, stride = 5, 252, 31, 252 * 4 + 16
        ins = [frames.random_frame(0x5EED0B00 + k, w, h, 4, stride) for k in range(n)]
        din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in ins]
        fill = np.full((h, stride), 0x5A, np.uint8)
        dout = [gpu.DeviceBuffer(fill.nbytes).upload(fill) for _ in range(n)]
        fi = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, stride, "RGBA") for b in din])
        fo = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, stride, "RGBA") for b in dout])
        gpu.check(L.mvfx_colorlut_transform_frames(dev.h, fi, fo, n, None))
        gpu.check(L.mvfx_stream_synchronize(None))
        for k in range(n):
            e = fill.copy()
            assert o.apply(ins[k], stride, e, stride, w, h, "RGBA") == 0
            assert np.array_equal(dout[k].download().reshape(h, stride), e), f"frame {k}"
        # unaligned rows: refused, not silently routed elsewhere
        odd_in = (gpu.Frame * 1)(gpu.make_frame(din[0].ptr + 4, 8, 2, stride, "RGBA"))
        odd_out = (gpu.Frame * 1)(gpu.make_frame(dout[0].ptr, 8, 2, stride, "RGBA"))
        assert L.mvfx_colorlut_transform_frames(dev.h, odd_in, odd_out, 1, None) == gpu.ERR_INVALID_ARGUMENT
        # RGBA64 has no table (2^48 colours): the automatic choice runs
        w16 = frames.random_frame(0x5EED0B10, 16, 4, 8, 16 * 8)
        e16 = np.empty_like(w16)
        assert o.apply(w16, 16 * 8, e16, 16 * 8, 16, 4, "RGBA64_LE") == 0
        s16, d16 = gpu.DeviceBuffer(w16.nbytes).upload(w16), gpu.DeviceBuffer(w16.nbytes)
        dev.apply_device(s16.ptr, 16 * 8, d16.ptr, 16 * 8, 16, 4, "RGBA64_LE")
        gpu.check(L.mvfx_stream_synchronize(None))
        assert np.array_equal(d16.download().reshape(w16.shape), e16)
    finally:
        L.mvfx_thread_set_options(gpu.options(placement=0).word)


@pytest.mark.parametrize("name,fmt,w,h,pad", [("analytic33", "RGBA", 256, 33, 0), ("analytic21", "RGBA", 127, 9, 8),
                                               ("curve1d_4096", "RGBA64_LE", 64, 17, 0), ("analytic9", "RGBA64_BE", 33, 5, 16)])
def test_colorlut_batch_matches_single(gpu, luts, name, fmt, w, h, pad):
    """mvfx_colorlut_transform_frames == N single calls (33 pairs cross the 32-pair launch split); LDS-staged, cell-packed,
    1-D and byte-path layouts; output row padding is left alone."""
    dev, o = luts[name]
    n = 33
    bpp = 8 if fmt != "RGBA" else 4
    stride = w * bpp + pad
    ins = [frames.random_frame(0x5EED0600 + k, w, h, bpp, stride) for k in range(n)]
    din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in ins]
    fill = np.full((h, stride), 0x5A, np.uint8)
    dout = [gpu.DeviceBuffer(fill.nbytes).upload(fill) for _ in range(n)]
    fi = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, stride, fmt) for b in din])
    fo = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, stride, fmt) for b in dout])
    gpu.check(gpu.lib().mvfx_colorlut_transform_frames(dev.h, fi, fo, n, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for k in range(n):
        exp = fill.copy()
        assert o.apply(ins[k], stride, exp, stride, w, h, fmt) == 0
        assert np.array_equal(dout[k].download().reshape(h, stride), exp), f"pair {k}"
    fo[1].height = h - 1
    assert gpu.lib().mvfx_colorlut_transform_frames(dev.h, fi, fo, 2, None) == gpu.ERR_INVALID_ARGUMENT
    assert gpu.lib().mvfx_colorlut_transform_frames(dev.h, fi, fo, 0, None) == gpu.ERR_INVALID_ARGUMENT
    assert gpu.lib().mvfx_colorlut_transform_frames(None, fi, fo, 1, None) == gpu.ERR_NO_LUT


def test_colorlut_4k_rgba_33(gpu, luts):
    """BASELINE config 3: 33^3 cube, 3840x2160 (RGBx in BASELINE == RGBA here)"""
    dev, o = luts["analytic33"]
    w, h = 3840, 2160
    for frame in (frames.random_frame(0x5EED0001, w, h), frames.smpte_like(w, h)):
        exp = np.empty_like(frame)
        assert o.apply(frame, w * 4, exp, w * 4, w, h, "RGBA") == 0
        src = gpu.DeviceBuffer(frame.nbytes).upload(frame)
        dst = gpu.DeviceBuffer(frame.nbytes)
        dev.apply_device(src.ptr, w * 4, dst.ptr, w * 4, w, h, "RGBA")
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        assert np.array_equal(dst.download().reshape(h, w * 4), exp)


@pytest.mark.parametrize("name", ["analytic33", "domain33", "analytic65", "analytic9", "nan_nodes4"])
@pytest.mark.parametrize("placement", ["wg", 7, 5], ids=["workgroup-window", "per-wave-windows", "tile-round2"])
def test_colorlut_window_kernels(gpu, luts, name, placement):
    """The three window kernels on the frames that exercise their windows: smooth gradients + noise (most pixels inside the window, some
    outside), flat bars (everything inside), uniform-random (everything outside: the per-lane global path), gradients with +-8 and +-16
    codes of noise (round 5: what the workgroup window is for), at sizes that are not multiples of the wave block or of the workgroup's
    128 x 40 block (partial blocks, a last workgroup whose trailing waves lie outside the frame), batched.
    "wg" = colorlut_xwg_kernel (MVFX_OPT_LUT_WG_WINDOW: no content probe in the way; cubes below 5 points fall through to the per-wave
    kernel), 7 = colorlut_xtile_kernel (x-prelerped table, per-wave windows), 5 = colorlut_tile_kernel (3 x 3 x 3 cell window): the
    oracle's bytes from all of them."""
    if name == "nan_nodes4":
        text = "LUT_3D_SIZE 4\n" + "".join(("nan 0.5 inf\n" if (i * 7) % 5 == 0 else ("0.25 -inf 2\n" if i % 3 == 0 else f"{i / 64:.6f} {1 - i / 64:.6f} 0.5\n"))
                                            for i in range(64))
        o = orc.CubeLut(text)
        assert o.ok, o.error
        dev = gpu.CubeLut(text)
    else:
        dev, o = luts[name]
    L = gpu.lib()
    gpu.check(L.mvfx_thread_set_options((gpu.options(wg_window=True) if placement == "wg" else gpu.options(placement=placement)).word))
    try:
        for (w, h) in ((3840, 2160), (1000, 250), (68, 20), (132, 44)):
            n = 2 if w > 2000 else 5
            srcs = []
            for k in range(n):
                kind = (k + (w // 4)) % 3 if k < 3 else k
                if kind in (0, 3, 4): # gradients + noise of the generator's own, then +-8 / +-16 codes on top
                    f = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED0C00 + k)).reshape(h, w, 4)
                    if kind >= 3:
                        amp = 8 if kind == 3 else 16
                        nz = np.random.default_rng(0x5EED0C20 + k).integers(-amp, amp + 1, (h, w, 3))
                        f = f.copy()
                        f[..., :3] = np.clip(f[..., :3].astype(np.int32) + nz, 0, 255).astype(np.uint8)
                else:
                    f = frames.smpte_like(w, h) if kind == 1 else frames.random_frame(0x5EED0C10 + k, w, h)
                srcs.append(np.ascontiguousarray(f).reshape(h, w * 4))
            din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in srcs]
            dout = [gpu.DeviceBuffer(f.nbytes) for f in srcs]
            fi = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in din])
            fo = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in dout])
            gpu.check(L.mvfx_colorlut_transform_frames(dev.h, fi, fo, n, None))
            gpu.check(L.mvfx_stream_synchronize(None))
            for k in range(n):
                exp = np.empty_like(srcs[k])
                assert o.apply(srcs[k], w * 4, exp, w * 4, w, h, "RGBA") == 0
                got = dout[k].download().reshape(h, w * 4)
                assert np.array_equal(got, exp), f"{name} {w}x{h} frame {k}: {np.count_nonzero(got != exp)} bytes differ"
    finally:
        L.mvfx_thread_set_options(gpu.options(placement=0).word)


def test_colorlut_rgba64_wide_values(gpu, luts):
    """RGBA64: every 16-bit value on each channel at least once (65536 px ramp + random)"""
    dev, o = luts["analytic33"]
    w, h = 4096, 16
    ramp = np.arange(65536, dtype=np.uint16)
    px = np.zeros((h * w, 4), np.uint16)
    px[:, 0] = ramp
    px[:, 1] = ramp[::-1]
    px[:, 2] = (ramp * 7 + 13)
    px[:, 3] = ramp ^ 0x5555
    for fmt, dt in (("RGBA64_LE", "<u2"), ("RGBA64_BE", ">u2")):
        src = px.astype(dt).view(np.uint8).reshape(h, w * 8)
        exp = np.empty_like(src)
        assert o.apply(src, w * 8, exp, w * 8, w, h, fmt) == 0
        got = np.empty_like(src)
        dev.apply_host(src.reshape(-1), w * 8, got.reshape(-1), w * 8, w, h, fmt)
        assert np.array_equal(got, exp)


def test_colorlut_errors(gpu, luts):
    dev, _ = luts["identity17"]
    a = np.zeros(256, np.uint8)
    fi = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBA")
    fo = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBA")
    assert gpu.lib().mvfx_colorlut_transform_frame_host(None, ctypes.byref(fi), ctypes.byref(fo)) == gpu.ERR_NO_LUT
    fx = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBx")
    assert gpu.lib().mvfx_colorlut_transform_frame_host(dev.h, ctypes.byref(fx), ctypes.byref(fo)) == gpu.ERR_UNSUPPORTED_FORMAT
    f2 = gpu.make_frame(a.ctypes.data, 4, 2, 16, "RGBA")
    assert gpu.lib().mvfx_colorlut_transform_frame_host(dev.h, ctypes.byref(fi), ctypes.byref(f2)) == gpu.ERR_NOT_NEGOTIATED
    big, _o = luts["analytic33"]
    gpu.lib().mvfx_thread_set_options(gpu.options(placement=2).word)
    try:
        assert gpu.lib().mvfx_colorlut_transform_frame_host(big.h, ctypes.byref(fi), ctypes.byref(fo)) == gpu.ERR_INVALID_ARGUMENT
    finally:
        gpu.lib().mvfx_thread_set_options(gpu.options(placement=0).word)


# ---- RGB10A2_LE: the third format of d3d12colorlut's caps (d3d12colorlut/imp.rs:236-244), device memory only
@pytest.mark.parametrize("which", ["3d33", "3d9", "1d"])
def test_colorlut_rgb10a2_le_matches_oracle(gpu, which):
    text = {"3d33": cubes.analytic_3d(33), "3d9": cubes.analytic_3d(9),
            "1d": "LUT_1D_SIZE 3\nDOMAIN_MIN 0 0.1 0\nDOMAIN_MAX 1 0.9 2\n0 0 0\n0.25 0.9 0.5\n1 1 0.75\n"}[which]
    w, h = 1021, 67
    stride = w * 4 + 12
    rng = np.random.default_rng(0x10A2)
    src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    # one row holding every 10-bit value in every channel
    v = np.arange(1024, dtype=np.uint32)[:w]
    src[0, :w * 4] = (v | (v[::-1] << 10) | (((v * 7) % 1024) << 20) | ((v % 4) << 30)).astype("<u4").view(np.uint8)
    want = np.zeros_like(src)
    o = orc.CubeLut(text)
    assert o.apply(src, stride, want, stride, w, h, "RGB10A2_LE") == 0
    lut = gpu.CubeLut(text)
    di, do = gpu.DeviceBuffer(src.nbytes).upload(src), gpu.DeviceBuffer(src.nbytes).upload(np.zeros_like(src))
    lut.apply_device(di.ptr, stride, do.ptr, stride, w, h, "RGB10A2_LE")
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = do.download().reshape(h, stride)
    assert np.array_equal(got[:, :w * 4], want[:, :w * 4])
    assert not got[:, w * 4:].any()  # row padding untouched


def test_colorlut_content_probe_picks_the_kernel_and_never_the_bytes(gpu):
    """Round 5: placement 0 asks a content probe (one workgroup looking at 256 blocks of an earlier frame of the LUT's stream) whether the
    pictures are calm -- per-wave windows -- or busy -- the workgroup window.  The verdict moves time, never bytes: frames of both kinds
    through the automatic choice equal the oracle before a verdict exists, after a calm one and after a busy one; and the verdict is the
    right one (diagnostic accessor mvfx_cube_lut_content_verdict)."""
    import ctypes
    text = cubes.analytic_3d(33)
    o = orc.CubeLut(text)
    dev = gpu.CubeLut(text)
    L = gpu.lib()
    w, h = 3840, 2160  # (round 6: a block is busy when a channel's sixteen samples span more than 13 codes -- the same gradients at 1920 x 1080 are twice
    # as steep per 64 x 20 block and count as busy, rightly: the bench's frames are 4K)
    calm = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED0D00)).reshape(h, w, 4)
    busy = calm.copy()
    busy[..., :3] = np.clip(busy[..., :3].astype(np.int32) + np.random.default_rng(0x5EED0D01).integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)
    busy_n = ctypes.c_uint32()
    assert L.mvfx_cube_lut_content_verdict(dev.h, ctypes.byref(busy_n)) == 0
    seen = []
    for f in (calm, busy, calm):
        f2 = f.reshape(h, w * 4)
        exp = np.empty_like(f2)
        assert o.apply(f2, w * 4, exp, w * 4, w, h, "RGBA") == 0
        src = gpu.DeviceBuffer(f2.nbytes).upload(f2)
        dst = gpu.DeviceBuffer(f2.nbytes)
        for i in range(70): # the probe looks at every 32nd call's frame; its verdict lands a little later
            dev.apply_device(src.ptr, w * 4, dst.ptr, w * 4, w, h, "RGBA")
            if i in (0, 40, 69):
                gpu.check(L.mvfx_stream_synchronize(None))
                assert np.array_equal(dst.download().reshape(h, w * 4), exp)
        seen.append((L.mvfx_cube_lut_content_verdict(dev.h, ctypes.byref(busy_n)), busy_n.value))
    assert [v for v, _ in seen] == [1, 2, 1], seen
    assert seen[0][1] <= 24 and seen[1][1] > 200, seen


def test_lut_users_on_one_device_share_one_device_copy(gpu):
    """Round 6: an mvfx_cube_lut keeps one device copy PER device (round 5 held a single copy and re-uploaded it on a device switch), made by
    the first transform on that device: no copy before the first use, one after it, and still one after four other threads -- each with its
    own streams -- have graded frames through the same handle on the same device."""
    import threading
    text = cubes.analytic_3d(17)
    dev = gpu.CubeLut(text)
    o = orc.CubeLut(text)
    L = gpu.lib()
    assert L.mvfx_cube_lut_device_copies(dev.h) == 0
    w, h = 256, 64
    f = frames.random_frame(0x5EED0E00, w, h)
    exp = np.empty_like(f)
    assert o.apply(f, w * 4, exp, w * 4, w, h, "RGBA") == 0
    got = np.empty_like(f)
    dev.apply_host(f.reshape(-1), w * 4, got.reshape(-1), w * 4, w, h, "RGBA")
    assert np.array_equal(got, exp) and L.mvfx_cube_lut_device_copies(dev.h) == 1
    bad = []

    def user(k):
        gpu.check(L.mvfx_set_device(0))
        out = np.empty_like(f)
        for _ in range(5):
            dev.apply_host(f.reshape(-1), w * 4, out.reshape(-1), w * 4, w, h, "RGBA")
            if not np.array_equal(out, exp):
                bad.append(k)
    ths = [threading.Thread(target=user, args=(k,)) for k in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad and L.mvfx_cube_lut_device_copies(dev.h) == 1
    assert L.mvfx_cube_lut_device_copies(None) == 0
