"""I420 <-> RGBA converters (csrc/convert_kernels.hip) against the oracle and against outputs of the REAL GStreamer
1.14.0 videoconvert (tests/golden/videoconvert_kat.npz): bit-exact, every size incl. 4K, odd sizes, unaligned planes."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu

KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "videoconvert_kat.npz"))
META = [m.split("|") for m in KAT["meta"].tolist()]


def _to_rgba(gpu, raw, w, h, standard=0, shift=0, out_pad=0):
    """raw: I420 frame in the GstVideoInfo layout; shift: bytes of misalignment of the device copy"""
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    dbuf = gpu.DeviceBuffer(size + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(dbuf.ptr + shift), raw.ctypes.data_as(ctypes.c_void_p), size, None))
    fin = gpu.make_i420(dbuf.ptr + shift, w, h, ys, cs, uo, vo)
    ostride = w * 4 + out_pad
    dout = gpu.DeviceBuffer(ostride * h + 64)
    fout = gpu.make_frame(dout.ptr, w, h, ostride, "RGBA")
    gpu.check(gpu.lib().mvfx_convert_i420_to_rgba(ctypes.byref(fin), ctypes.byref(fout), standard, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    return dout.download(ostride * h).reshape(h, ostride)[:, : w * 4]


def _to_i420(gpu, px, w, h, stride, standard=0, shift=0):
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    din = gpu.DeviceBuffer(px.nbytes + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), px.ctypes.data_as(ctypes.c_void_p), px.nbytes, None))
    fin = gpu.make_frame(din.ptr + shift, w, h, stride, "RGBA")
    dout = gpu.DeviceBuffer(size + 64)
    fout = gpu.make_i420(dout.ptr, w, h, ys, cs, uo, vo)
    gpu.check(gpu.lib().mvfx_convert_rgba_to_i420(ctypes.byref(fin), ctypes.byref(fout), standard, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    d = dout.download(size)
    Y = d[: ys * yr].reshape(yr, ys)[:h, :w]
    U = d[uo: uo + cs * cr].reshape(cr, cs)[: (h + 1) // 2, : (w + 1) // 2]
    V = d[vo: vo + cs * cr].reshape(cr, cs)[: (h + 1) // 2, : (w + 1) // 2]
    return Y, U, V


def _to_nv12(gpu, px, w, h, stride, standard=0, shift=0):
    """GstVideoInfo layout of NV12: Y stride RU4(w) x RU2(h) rows, UV stride RU4(RU2(w)) x RU2(h)/2 rows"""
    ru = lambda v, a: (v + a - 1) // a * a
    ys, uvs, yr, cr = ru(w, 4), ru(ru(w, 2), 4), ru(h, 2), ru(h, 2) // 2
    din = gpu.DeviceBuffer(px.nbytes + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), px.ctypes.data_as(ctypes.c_void_p), px.nbytes, None))
    fin = gpu.make_frame(din.ptr + shift, w, h, stride, "RGBA")
    dout = gpu.DeviceBuffer(ys * yr + uvs * cr + 64)
    fout = gpu.PlanarFrame()
    fout.data[0], fout.data[1] = dout.ptr, dout.ptr + ys * yr
    fout.stride[0], fout.stride[1] = ys, uvs
    fout.width, fout.height, fout.format = w, h, gpu.FORMATS["NV12"]
    gpu.check(gpu.lib().mvfx_convert_rgba_to_nv12(ctypes.byref(fin), ctypes.byref(fout), standard, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    d = dout.download(ys * yr + uvs * cr)
    return d[: ys * yr].reshape(yr, ys)[:h, :w], d[ys * yr:].reshape(cr, uvs)[: (h + 1) // 2, : 2 * ((w + 1) // 2)]


def _nv12_to_rgba(gpu, raw, w, h, standard=0, shift=0, out_pad=0):
    ys, uvs, yr, cr, uvo, size = orc.nv12_layout(w, h)
    din = gpu.DeviceBuffer(size + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), raw.ctypes.data_as(ctypes.c_void_p), size, None))
    fin = gpu.PlanarFrame()
    fin.data[0], fin.data[1] = din.ptr + shift, din.ptr + shift + uvo
    fin.stride[0], fin.stride[1] = ys, uvs
    fin.width, fin.height, fin.format = w, h, gpu.FORMATS["NV12"]
    ostride = w * 4 + out_pad
    dout = gpu.DeviceBuffer(ostride * h + 64)
    fout = gpu.make_frame(dout.ptr + shift, w, h, ostride, "RGBA")
    gpu.check(gpu.lib().mvfx_convert_nv12_to_rgba(ctypes.byref(fin), ctypes.byref(fout), standard, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    return dout.download(ostride * h + shift)[shift:].reshape(h, ostride)[:, : w * 4]


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("nv12_to_rgba")], ids=lambda m: m[0])
def test_nv12_to_rgba_matches_gstreamer_goldens(gpu, meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    raw = frames.splitmix64_bytes(seed, orc.nv12_layout(w, h)[5])
    got = _nv12_to_rgba(gpu, raw, w, h)
    assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == digest


@pytest.mark.parametrize("geom", [(64, 32), (65, 33), (1, 1), (2, 2), (3, 5), (9, 3), (24, 578), (8, 2160), (1920, 1080), (1919, 1079)])
@pytest.mark.parametrize("standard", [0, 1, 2, 3])
def test_nv12_to_rgba_matches_oracle(gpu, geom, standard):
    """chroma interpolated horizontally, then vertically (the element's generic path), every colorimetry default, odd sizes, an
    unaligned frame with a padded output stride"""
    w, h = geom
    raw = frames.splitmix64_bytes(0x5EED1000 + w * 13 + h, orc.nv12_layout(w, h)[5])
    rc, want = orc.convert_nv12_to_rgba(raw, w, h, standard)
    assert rc == 0
    assert np.array_equal(_nv12_to_rgba(gpu, raw, w, h, standard), want)
    if standard in (0, 2):
        assert np.array_equal(_nv12_to_rgba(gpu, raw, w, h, standard, shift=1, out_pad=4), want)


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("rgba_to_nv12")], ids=lambda m: m[0])
def test_rgba_to_nv12_matches_gstreamer_goldens(gpu, meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    px = frames.random_frame(seed, w, h)
    Y, UV = _to_nv12(gpu, px, w, h, w * 4)
    packed = np.concatenate([Y.reshape(-1), UV.reshape(-1)])
    assert hashlib.sha256(packed.tobytes()).hexdigest() == digest


@pytest.mark.parametrize("geom", [(65, 33), (1, 1), (3, 3), (5, 7), (7, 601), (66, 33), (65, 34), (641, 481), (1919, 1079), (9, 2161)])
@pytest.mark.parametrize("standard", [0, 1, 2, 3])
def test_rgba_to_i420_and_nv12_odd_sizes_match_oracle(gpu, geom, standard):
    """odd-sized frames: the last column / row replicated to the next even size (what the element does); I420 and NV12 carry the
    same samples; 4-byte aligned and unaligned input"""
    w, h = geom
    stride = w * 4 + 4
    px = frames.random_frame(0x5EED0F00 + w * 13 + h, w, h, 4, stride)
    rc, Yw, Uw, Vw = orc.convert_rgba_to_i420(px, w, h, stride, standard)
    assert rc == 0
    for shift in (0, 1):
        Y, U, V = _to_i420(gpu, px, w, h, stride, standard, shift=shift)
        assert np.array_equal(Y, Yw) and np.array_equal(U, Uw) and np.array_equal(V, Vw)
        Y, UV = _to_nv12(gpu, px, w, h, stride, standard, shift=shift)
        assert np.array_equal(Y, Yw) and np.array_equal(UV[:, 0::2], Uw) and np.array_equal(UV[:, 1::2], Vw)


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("i420_to_rgba")], ids=lambda m: m[0])
def test_i420_to_rgba_matches_gstreamer_goldens(gpu, meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    raw = frames.splitmix64_bytes(seed, orc.i420_layout(w, h)[6])
    got = _to_rgba(gpu, raw, w, h)
    assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == digest


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("rgba_to_i420")], ids=lambda m: m[0])
def test_rgba_to_i420_matches_gstreamer_goldens(gpu, meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    px = frames.random_frame(seed, w, h)
    Y, U, V = _to_i420(gpu, px, w, h, w * 4)
    packed = np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])
    assert hashlib.sha256(packed.tobytes()).hexdigest() == digest


@pytest.mark.parametrize("geom", [(64, 32), (66, 34), (65, 33), (1, 1), (7, 5), (9, 3), (24, 578), (8, 2160), (1920, 1080), (2, 2)])
@pytest.mark.parametrize("standard", [0, 1, 2, 3])
def test_i420_to_rgba_matches_oracle(gpu, geom, standard):
    w, h = geom
    raw = frames.splitmix64_bytes(0x5EED0D00 + w * 13 + h, orc.i420_layout(w, h)[6])
    rc, want = orc.convert_i420_to_rgba(raw, w, h, standard)
    assert rc == 0
    assert np.array_equal(_to_rgba(gpu, raw, w, h, standard), want)
    if standard == 0:  # misaligned planes and a padded output stride: the per-sample path
        assert np.array_equal(_to_rgba(gpu, raw, w, h, standard, shift=1, out_pad=4), want)


@pytest.mark.parametrize("geom", [(64, 32), (66, 34), (2, 2), (4, 2), (10, 6), (24, 578), (8, 2160), (1920, 1080), (18, 600)])
@pytest.mark.parametrize("standard", [0, 1, 2, 3])
def test_rgba_to_i420_matches_oracle(gpu, geom, standard):
    w, h = geom
    stride = w * 4 + (16 if w % 8 else 0)
    px = frames.random_frame(0x5EED0E00 + w * 13 + h, w, h, 4, stride)
    rc, Yw, Uw, Vw = orc.convert_rgba_to_i420(px, w, h, stride, standard)
    assert rc == 0
    Y, U, V = _to_i420(gpu, px, w, h, stride, standard)
    assert np.array_equal(Y, Yw) and np.array_equal(U, Uw) and np.array_equal(V, Vw)
    if standard in (0, 2):
        Y, U, V = _to_i420(gpu, px, w, h, stride, standard, shift=4)  # 4-byte aligned only: per-sample path
        assert np.array_equal(Y, Yw) and np.array_equal(U, Uw) and np.array_equal(V, Vw)


def test_round_trip_is_close_and_grey_is_exact(gpu):
    """RGBA -> I420 -> RGBA of a flat grey frame at full 4K size: chroma exactly 128, luma flat, and the frame comes back
    within 4 levels (videoconvert's orc fast path multiplies splatbw(Y-128) = 257 (Y-128) by 298 >> 16, a gain of 1.1686
    instead of 1.1644, and adds 128 instead of 16 * 1.164: 180 -> 177 in the real element too)"""
    w, h = 3840, 2160
    g = np.full((h, w * 4), 180, np.uint8)
    g[:, 3::4] = 255
    Y, U, V = _to_i420(gpu, g, w, h, w * 4)
    assert (U == 128).all() and (V == 128).all() and (Y == Y[0, 0]).all()
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    raw = np.zeros(size, np.uint8)
    raw[: ys * yr].reshape(yr, ys)[:h, :w] = Y
    raw[uo: uo + cs * cr] = 128
    raw[vo: vo + cs * cr] = 128
    back = _to_rgba(gpu, raw, w, h)
    assert np.abs(back.astype(int) - g.astype(int)).max() <= 4 and (back == back[0, 0:4].tolist() * w).all()


def test_errors(gpu):
    w, h = 16, 8
    buf = gpu.DeviceBuffer(4096)
    i420 = gpu.make_i420(buf.ptr, w, h, 16, 8, 128, 192)
    rgba = gpu.make_frame(buf.ptr + 1024, w, h, w * 4, "RGBA")
    L = gpu.lib()
    assert L.mvfx_convert_i420_to_rgba(ctypes.byref(i420), ctypes.byref(rgba), 7, None) == gpu.ERR_INVALID_ARGUMENT
    bgra = gpu.make_frame(buf.ptr + 1024, w, h, w * 4, "BGRA")
    assert L.mvfx_convert_i420_to_rgba(ctypes.byref(i420), ctypes.byref(bgra), 0, None) == gpu.ERR_UNSUPPORTED_FORMAT
    small = gpu.make_frame(buf.ptr + 1024, w, h // 2, w * 4, "RGBA")
    assert L.mvfx_convert_i420_to_rgba(ctypes.byref(i420), ctypes.byref(small), 0, None) == gpu.ERR_NOT_NEGOTIATED
    odd_i = gpu.make_i420(buf.ptr, 15, 7, 16, 8, 128, 192)
    odd = gpu.make_frame(buf.ptr + 1024, 15, 7, 64, "RGBA")
    assert L.mvfx_convert_rgba_to_i420(ctypes.byref(odd), ctypes.byref(odd_i), 0, None) == gpu.OK   # odd sizes: edge replication (round 3)


# ---------------------------------------------------------------- colorlut on I420 frames (fused videoconvert ! colorlut ! videoconvert)

def _oracle_lut_i420(o, raw, w, h, standard):
    rc, rgba = orc.convert_i420_to_rgba(raw, w, h, standard)
    assert rc == 0
    lut_out = np.zeros_like(rgba)
    assert o.apply(rgba, w * 4, lut_out, w * 4, w, h, "RGBA") == 0
    rc, Y, U, V = orc.convert_rgba_to_i420(lut_out, w, h, w * 4, standard)
    assert rc == 0
    return Y, U, V


@pytest.mark.parametrize("lut_name", ["analytic33", "analytic9", "curve1d_256", "big_nodes70", "nan_domain"])
@pytest.mark.parametrize("geom", [(64, 32, 0), (2056, 6, 0), (24, 578, 0), (8, 2160, 0), (1920, 1080, 0), (36, 10, 0), (64, 32, 2),
                                  (4104, 578, 0)])  # last: co-sited filter across workgroup boundaries (> 2048 px wide, HD)
def test_colorlut_i420_matches_three_oracles(gpu, lut_name, geom):
    """fused kernel (width % 8 == 0, aligned) and the three-step path (w = 36; misaligned planes; NaN domain) against
    oracle(i420->rgba) -> oracle(colorlut) -> oracle(rgba->i420); SD / HD co-sited / UHD defaults by height"""
    from tests import cubes
    text = {"analytic33": cubes.analytic_3d(33), "analytic9": cubes.analytic_3d(9), "curve1d_256": cubes.curve_1d(256),
            "big_nodes70": cubes.identity_3d(70, 4),  # > 65: no cell-packed copy, node layout
            "nan_domain": "LUT_1D_SIZE 2\nDOMAIN_MIN nan 0 0\n0 0.1 0.2\n1 0.9 0.8\n"}[lut_name]
    if lut_name == "big_nodes70" and geom[0] * geom[1] > 100000:
        pytest.skip("one large case per LUT family is enough")
    o = orc.CubeLut(text)
    assert o.ok
    dev = gpu.CubeLut(text)
    w, h, shift = geom
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    raw = frames.splitmix64_bytes(0x5EED0F00 + w + h, size)
    Yw, Uw, Vw = _oracle_lut_i420(o, raw, w, h, 0)
    din = gpu.DeviceBuffer(size + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), raw.ctypes.data_as(ctypes.c_void_p), size, None))
    dout = gpu.DeviceBuffer(size + 64)
    fin = gpu.make_i420(din.ptr + shift, w, h, ys, cs, uo, vo)
    fout = gpu.make_i420(dout.ptr, w, h, ys, cs, uo, vo)
    gpu.check(gpu.lib().mvfx_colorlut_transform_i420(dev.h, ctypes.byref(fin), ctypes.byref(fout), 0, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    d = dout.download(size)
    Y = d[: ys * yr].reshape(yr, ys)[:h, :w]
    U = d[uo: uo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2]
    V = d[vo: vo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2]
    assert np.array_equal(Y, Yw), np.argwhere(Y != Yw)[:5]
    assert np.array_equal(U, Uw), np.argwhere(U != Uw)[:5]
    assert np.array_equal(V, Vw), np.argwhere(V != Vw)[:5]


@pytest.mark.parametrize("settings", [(90.0, 1.25, -0.05, 0.9, 0.02), (-45.0, 0.8, 0.1, 1.1, -0.03), (0.0, 1.0, 0.0, 1.0, 0.0),
                                      (float("nan"), 1.0, 0.0, 1.0, 0.0), (725.0, 2.0, 0.0, 0.5, 0.25)],
                         ids=["bench", "negshift", "identity", "nan-literal", "bigshift-literal"])
@pytest.mark.parametrize("geom", [(64, 32, 0), (24, 578, 0), (8, 2160, 0), (1920, 1080, 0), (4104, 578, 0), (36, 10, 0), (64, 32, 4)])
def test_hsvfilter_i420_matches_three_oracles(gpu, settings, geom):
    """videoconvert ! hsvfilter ! videoconvert fused (and the three-step path for w = 36 / misaligned planes) against
    oracle(i420->rgba) -> oracle(hsvfilter in place) -> oracle(rgba->i420)"""
    w, h, shift = geom
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    raw = frames.splitmix64_bytes(0x5EED1000 + w + h, size)
    rc, rgba = orc.convert_i420_to_rgba(raw, w, h, 0)
    assert rc == 0
    orc.hsvfilter(rgba, w, w * 4, "RGBA", settings)
    rc, Yw, Uw, Vw = orc.convert_rgba_to_i420(rgba, w, h, w * 4, 0)
    assert rc == 0
    din = gpu.DeviceBuffer(size + 64)
    gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), raw.ctypes.data_as(ctypes.c_void_p), size, None))
    dout = gpu.DeviceBuffer(size + 64)
    fin = gpu.make_i420(din.ptr + shift, w, h, ys, cs, uo, vo)
    fout = gpu.make_i420(dout.ptr, w, h, ys, cs, uo, vo)
    st = gpu.HsvFilterSettings(*settings)
    gpu.check(gpu.lib().mvfx_hsvfilter_transform_i420(ctypes.byref(fin), ctypes.byref(fout), ctypes.byref(st), 0, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    d = dout.download(size)
    Y = d[: ys * yr].reshape(yr, ys)[:h, :w]
    U = d[uo: uo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2]
    V = d[vo: vo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2]
    assert np.array_equal(Y, Yw), np.argwhere(Y != Yw)[:5]
    assert np.array_equal(U, Uw), np.argwhere(U != Uw)[:5]
    assert np.array_equal(V, Vw), np.argwhere(V != Vw)[:5]
    # aliasing input and output is refused
    assert gpu.lib().mvfx_hsvfilter_transform_i420(ctypes.byref(fin), ctypes.byref(fin), ctypes.byref(st), 0, None) == gpu.ERR_INVALID_ARGUMENT


def test_batched_converters_match_single_frame_calls(gpu):
    """one frame from each of n streams per launch (blockIdx.z): same bytes as n single-frame calls; batches must share geometry"""
    w, h, n = 64, 34, 5
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    raws = [frames.splitmix64_bytes(0x5EED1100 + i, size) for i in range(n)]
    dins = [gpu.DeviceBuffer(size).upload(r) for r in raws]
    douts = [gpu.DeviceBuffer(w * 4 * h) for _ in range(n)]
    fin = (gpu.PlanarFrame * n)(*[gpu.make_i420(d.ptr, w, h, ys, cs, uo, vo) for d in dins])
    fout = (gpu.Frame * n)(*[gpu.make_frame(d.ptr, w, h, w * 4, "RGBA") for d in douts])
    gpu.check(gpu.lib().mvfx_convert_i420_to_rgba_frames(fin, fout, n, 0, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    rgbas = []
    for i in range(n):
        rc, want = orc.convert_i420_to_rgba(raws[i], w, h, 0)
        got = douts[i].download(w * 4 * h).reshape(h, w * 4)
        assert np.array_equal(got, want)
        rgbas.append(want)
    back = [gpu.DeviceBuffer(size) for _ in range(n)]
    fback = (gpu.PlanarFrame * n)(*[gpu.make_i420(d.ptr, w, h, ys, cs, uo, vo) for d in back])
    gpu.check(gpu.lib().mvfx_convert_rgba_to_i420_frames(fout, fback, n, 0, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for i in range(n):
        rc, Yw, Uw, Vw = orc.convert_rgba_to_i420(rgbas[i], w, h, w * 4, 0)
        d = back[i].download(size)
        assert np.array_equal(d[: ys * yr].reshape(yr, ys)[:h, :w], Yw)
        assert np.array_equal(d[uo: uo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2], Uw)
        assert np.array_equal(d[vo: vo + cs * cr].reshape(cr, cs)[: h // 2, : w // 2], Vw)
    fout[2].stride = w * 4 + 16
    assert gpu.lib().mvfx_convert_i420_to_rgba_frames(fin, fout, n, 0, None) == gpu.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("out_fmt", ["RGBA", "ARGB", "BGRA", "ABGR"])
@pytest.mark.parametrize("settings", [(120.0, 40.0, 0.6, 0.4, 0.6, 0.4), (0.0, 10.0, 0.0, 0.15, 0.0, 0.3), (400.0, 180.0, 0.5, 1.0, 0.5, 1.0)],
                         ids=["bench", "defaults", "literal-huge-ref"])
def test_hsvdetector_i420_matches_two_oracles(gpu, out_fmt, settings):
    """videoconvert ! hsvdetector fused: oracle(i420->rgba) -> oracle(hsvdetector RGBx -> out_fmt); odd sizes, misaligned planes"""
    for (w, h, shift) in [(64, 32, 0), (65, 33, 0), (24, 578, 0), (1920, 1080, 0), (40, 10, 1)]:
        ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
        raw = frames.splitmix64_bytes(0x5EED1200 + w + h, size)
        rc, rgba = orc.convert_i420_to_rgba(raw, w, h, 0)
        assert rc == 0
        want = np.zeros((h, w * 4), np.uint8)
        assert orc.hsvdetector(rgba, w * 4, "RGBx", want, w * 4, out_fmt, w, settings) == 0
        din = gpu.DeviceBuffer(size + 64)
        gpu.check(gpu.lib().mvfx_copy_to_device(ctypes.c_void_p(din.ptr + shift), raw.ctypes.data_as(ctypes.c_void_p), size, None))
        dout = gpu.DeviceBuffer(w * 4 * h)
        fin = gpu.make_i420(din.ptr + shift, w, h, ys, cs, uo, vo)
        fout = gpu.make_frame(dout.ptr, w, h, w * 4, out_fmt)
        st = gpu.HsvDetectorSettings(*settings)
        gpu.check(gpu.lib().mvfx_hsvdetector_transform_i420(ctypes.byref(fin), ctypes.byref(fout), ctypes.byref(st), 0, None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        got = dout.download(w * 4 * h).reshape(h, w * 4)
        assert np.array_equal(got, want), (w, h, np.argwhere(got != want)[:5])


def test_fused_entry_points_reject_bad_arguments(gpu):
    from tests import cubes
    w, h = 64, 32
    ys, cs, yr, cr, uo, vo, size = orc.i420_layout(w, h)
    a = gpu.DeviceBuffer(size)
    b = gpu.DeviceBuffer(size)
    fa = gpu.make_i420(a.ptr, w, h, ys, cs, uo, vo)
    fb = gpu.make_i420(b.ptr, w, h, ys, cs, uo, vo)
    lut = gpu.CubeLut(cubes.analytic_3d(9))
    L = gpu.lib()
    assert L.mvfx_colorlut_transform_i420(lut.h, ctypes.byref(fa), ctypes.byref(fa), 0, None) == gpu.ERR_INVALID_ARGUMENT  # aliasing
    assert L.mvfx_colorlut_transform_i420(None, ctypes.byref(fa), ctypes.byref(fb), 0, None) == gpu.ERR_NO_LUT
    assert L.mvfx_colorlut_transform_i420(lut.h, ctypes.byref(fa), ctypes.byref(fb), 9, None) == gpu.ERR_INVALID_ARGUMENT
    small = gpu.make_i420(b.ptr, w, h // 2, ys, cs, uo, vo)
    assert L.mvfx_colorlut_transform_i420(lut.h, ctypes.byref(fa), ctypes.byref(small), 0, None) == gpu.ERR_NOT_NEGOTIATED
    st = gpu.HsvFilterSettings.default()
    odd = gpu.make_i420(a.ptr, w - 1, h, ys, cs, uo, vo)
    odd_out = gpu.make_i420(b.ptr, w - 1, h, ys, cs, uo, vo)
    assert L.mvfx_hsvfilter_transform_i420(ctypes.byref(odd), ctypes.byref(odd_out), ctypes.byref(st), 0, None) == gpu.ERR_INVALID_ARGUMENT
    ds = gpu.HsvDetectorSettings.default()
    out = gpu.DeviceBuffer(w * h * 4)
    rgbx = gpu.make_frame(out.ptr, w, h, w * 4, "RGBx")  # not an output format of the detector
    assert L.mvfx_hsvdetector_transform_i420(ctypes.byref(fa), ctypes.byref(rgbx), ctypes.byref(ds), 0, None) == gpu.ERR_UNSUPPORTED_FORMAT
    rgba = gpu.make_frame(out.ptr, w, h, w * 4, "RGBA")
    assert L.mvfx_convert_i420_to_rgba_frames(ctypes.byref(fa), ctypes.byref(rgba), 0, 0, None) == gpu.ERR_INVALID_ARGUMENT  # empty batch
