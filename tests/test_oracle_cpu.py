"""CPU tests that pin the oracle (oracle/*.c):
  1. against the reference's own known-answer vectors (video/hsv/src/hsvutils.rs:219-279),
  2. against a second independent restatement (tests/np_twin.py, numpy float32),
  3. against the reference's pipeline-level pins for the third-party-backed elements
     (tests/colordetect.rs:21-68, tests/videocompare.rs:57-139).
"""
import numpy as np
import pytest

from tests import cubes, frames, np_twin
from tests import oracle_binding as orc

EPSILON = 0.00001  # hsvutils.rs:40

# hsvutils.rs:219-235
RGB = {"white": (255, 255, 255), "black": (0, 0, 0), "red": (255, 0, 0), "green": (0, 255, 0), "blue": (0, 0, 255)}
BGR = {"white": (255, 255, 255), "black": (0, 0, 0), "red": (0, 0, 255), "green": (0, 255, 0), "blue": (255, 0, 0)}
HSV = {"white": (0.0, 0.0, 1.0), "black": (0.0, 0.0, 0.0), "red": (0.0, 1.0, 1.0), "green": (120.0, 1.0, 1.0),
       "blue": (240.0, 1.0, 1.0)}


def is_equivalent(hsv, expected, eps):
    """hsvutils.rs:203-217 (circular hue compare)"""
    f = np.float32
    shifted = f(hsv[0]) + (f(180.0) - f(expected[0]))
    if shifted < 0:
        shifted += f(360.0)
    shifted = np.fmod(shifted, f(360.0))
    return abs(shifted - f(180.0)) < eps and abs(hsv[1] - expected[1]) < eps and abs(hsv[2] - expected[2]) < eps


@pytest.mark.parametrize("name", sorted(RGB))
def test_kat_from_rgb_from_bgr(name):
    """hsvutils.rs:237-257"""
    assert is_equivalent(orc.from_rgb(RGB[name]), HSV[name], EPSILON)
    assert is_equivalent(orc.from_rgb(BGR[name], bgr=True), HSV[name], EPSILON)


@pytest.mark.parametrize("name", sorted(RGB))
def test_kat_to_rgb_to_bgr(name):
    """hsvutils.rs:259-279 (exact)"""
    assert orc.to_rgb(HSV[name]) == RGB[name]
    assert orc.to_rgb(HSV[name], bgr=True) == BGR[name]


def test_survey_probe_identity_is_not_identity():
    """default settings change [12,200,77] to [11,200,76]."""
    f = np.array([[12, 200, 77, 9]], dtype=np.uint8)
    assert orc.hsvfilter(f, 1, 4, "RGBA", (0, 1, 0, 1, 0)) == 0
    assert f.tolist() == [[11, 200, 76, 9]]


SETTINGS = [
    (0.0, 1.0, 0.0, 1.0, 0.0), (90.0, 1.25, -0.05, 0.9, 0.02), (-123.4, 0.5, 0.3, 1.7, -0.2),
    (360.0, 1.0, 0.0, 1.0, 0.0), (-360.0, 2.0, -0.5, 0.25, 0.5), (720.5, 0.7, 0.1, 1.2, -0.1),
    (-1e6, 1.1, 0.0, 0.9, 0.0), (float("nan"), 1.0, 0.0, 1.0, 0.0),
    (10.0, float("inf"), 0.0, float("nan"), 0.0), (float("inf"), 1.0, float("-inf"), 1.0, 0.0),
    (1e-35, 1.0, 0.0, 1.0, 0.0),
]


@pytest.mark.parametrize("settings", SETTINGS)
def test_hsvfilter_oracle_vs_numpy_twin(settings):
    n = 1 << 17
    px = frames.splitmix64_bytes(0x5EED0001, n * 4).reshape(1, n * 4)
    got = px.copy()
    assert orc.hsvfilter(got, n, n * 4, "RGBA", settings) == 0
    p = px.reshape(n, 4)
    R, G, B = np_twin.hsvfilter_pixels(p[:, 0], p[:, 1], p[:, 2], settings)
    g = got.reshape(n, 4)
    assert np.array_equal(g[:, 0], R) and np.array_equal(g[:, 1], G) and np.array_equal(g[:, 2], B)
    assert np.array_equal(g[:, 3], p[:, 3])


def test_hsvfilter_oracle_vs_twin_on_structured_grid():
    """every (R,G,B) with channels on a 17-step grid incl. 0/255 and near-equal pairs"""
    vals = np.array(sorted(set(list(range(0, 256, 16)) + [1, 2, 127, 128, 254, 255])), dtype=np.uint8)
    R, G, B = [a.reshape(-1) for a in np.meshgrid(vals, vals, vals, indexing="ij")]
    n = R.size
    px = np.stack([R, G, B, np.full(n, 7, np.uint8)], axis=1).reshape(1, n * 4).copy()
    for settings in SETTINGS[:6]:
        got = px.copy()
        orc.hsvfilter(got, n, n * 4, "RGBx", settings)
        r2, g2, b2 = np_twin.hsvfilter_pixels(R, G, B, settings)
        g = got.reshape(n, 4)
        assert np.array_equal(g[:, 0], r2) and np.array_equal(g[:, 1], g2) and np.array_equal(g[:, 2], b2)


FILTER_LAYOUT = {  # fmt -> (bpp, index of R, G, B)
    "RGBx": (4, 0, 1, 2), "RGBA": (4, 0, 1, 2), "xRGB": (4, 1, 2, 3), "ARGB": (4, 1, 2, 3),
    "BGRx": (4, 2, 1, 0), "BGRA": (4, 2, 1, 0), "xBGR": (4, 3, 2, 1), "ABGR": (4, 3, 2, 1),
    "RGB": (3, 0, 1, 2), "BGR": (3, 2, 1, 0),
}


@pytest.mark.parametrize("fmt", sorted(FILTER_LAYOUT))
def test_hsvfilter_oracle_formats_strides(fmt):
    """channel placement per format (hsvfilter/imp.rs:327-373); x byte and padding untouched"""
    bpp, ir, ig, ib = FILTER_LAYOUT[fmt]
    w, h = 37, 6
    stride = (w * bpp + 3) // 4 * 4 + 8
    while (stride * h) % bpp:
        stride += 4
    f = frames.random_frame(0x5EED0300, w, h, bpp, stride)
    got = f.copy()
    settings = (90.0, 1.25, -0.05, 0.9, 0.02)
    assert orc.hsvfilter(got, w, stride, fmt, settings) == 0
    exp = f.copy()
    px = exp[:, :w * bpp].reshape(h, w, bpp)
    R, G, B = np_twin.hsvfilter_pixels(px[..., ir].reshape(-1), px[..., ig].reshape(-1), px[..., ib].reshape(-1), settings)
    px[..., ir] = R.reshape(h, w)
    px[..., ig] = G.reshape(h, w)
    px[..., ib] = B.reshape(h, w)
    exp[:, :w * bpp] = px.reshape(h, w * bpp)
    assert np.array_equal(got, exp)


def test_hsvfilter_oracle_reference_panics():
    """642x481 RGB -> plane size % 3 != 0 -> assert_eq! (hsvfilter/imp.rs:92)"""
    f = frames.random_frame(1, 642, 481, 3, 1928)
    assert orc.hsvfilter(f, 642, 1928, "RGB", (0, 1, 0, 1, 0)) == -1
    assert orc.hsvfilter(np.zeros(64, np.uint8), 4, 16, "RGBA64_LE", (0, 1, 0, 1, 0)) == -2


DET_SETTINGS = [(0.0, 10.0, 0.0, 0.15, 0.0, 0.3), (120.0, 40.0, 0.6, 0.4, 0.6, 0.4), (350.0, 25.0, 0.5, 0.5, 0.5, 0.5),
                (-200.0, 180.0, 1.0, 1.0, 1.0, 1.0), (1e6, 90.0, 0.5, 0.3, 0.5, 0.3),
                (float("nan"), 10.0, 0.0, 0.15, 0.0, 0.3)]
DET_IN = {"RGBx": (4, 0, 1, 2), "xRGB": (4, 1, 2, 3), "BGRx": (4, 2, 1, 0), "xBGR": (4, 3, 2, 1), "RGB": (3, 0, 1, 2),
          "BGR": (3, 2, 1, 0)}
DET_OUT = {"RGBA": (0, 1, 2, 3), "ARGB": (1, 2, 3, 0), "BGRA": (2, 1, 0, 3), "ABGR": (3, 2, 1, 0)}


@pytest.mark.parametrize("in_fmt", sorted(DET_IN))
@pytest.mark.parametrize("out_fmt", sorted(DET_OUT))
def test_hsvdetector_oracle_vs_twin_all_pairs(in_fmt, out_fmt):
    """SURVEY 8a channel rule for all 24 pairs (hsvdetector/imp.rs:428-704)"""
    bpp, ir, ig, ib = DET_IN[in_fmt]
    orr, og, ob, oa = DET_OUT[out_fmt]
    w, h = 53, 5
    in_stride = (w * bpp + 3) // 4 * 4 + 4
    while (in_stride * h) % bpp:
        in_stride += 4
    out_stride = w * 4 + 12
    src = frames.random_frame(0x5EED0400, w, h, bpp, in_stride)
    for settings in DET_SETTINGS:
        out = np.full((h, out_stride), 0x5A, np.uint8)
        assert orc.hsvdetector(src, in_stride, in_fmt, out, out_stride, out_fmt, w, settings) == 0
        px = src[:, :w * bpp].reshape(h, w, bpp)
        R, G, B = px[..., ir], px[..., ig], px[..., ib]
        alpha = np_twin.hsvdetector_alpha(R.reshape(-1), G.reshape(-1), B.reshape(-1), settings).reshape(h, w)
        exp = np.full((h, out_stride), 0x5A, np.uint8)
        o = exp[:, :w * 4].reshape(h, w, 4)
        o[..., orr], o[..., og], o[..., ob], o[..., oa] = R, G, B, alpha
        exp[:, :w * 4] = o.reshape(h, w * 4)
        assert np.array_equal(out, exp), f"{in_fmt}->{out_fmt} {settings}"


def test_hsvdetector_oracle_asserts():
    a = np.zeros(64, np.uint8)
    assert orc.hsvdetector(a, 16, "RGBx", np.zeros(32, np.uint8), 16, "RGBA", 4, DET_SETTINGS[0]) == -1  # row counts differ
    assert orc.hsvdetector(a, 16, "RGBA", a.copy(), 16, "RGBA", 4, DET_SETTINGS[0]) == -2               # RGBA is not an input


# ---------------------------------------------------------------- colorlut

def _lut_cases():
    return {
        "identity17": cubes.identity_3d(17),
        "analytic9": cubes.analytic_3d(9),
        "analytic33": cubes.analytic_3d(33),
        "curve1d_256": cubes.curve_1d(256),
        "curve1d_2": cubes.curve_1d(2),
        "curve1d_domain": cubes.curve_1d(64, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))),
        "nan_nodes": "LUT_3D_SIZE 2\n" + "nan 0.5 inf\n" * 4 + "0.25 -inf 2\n" * 4,
        "nan_domain": "LUT_1D_SIZE 2\nDOMAIN_MIN nan 0 0\n0 0.1 0.2\n1 0.9 0.8\n",
    }


@pytest.mark.parametrize("name", sorted(_lut_cases()))
@pytest.mark.parametrize("fmt", ["RGBA", "RGBA64_LE", "RGBA64_BE"])
def test_colorlut_oracle_vs_numpy_twin(name, fmt):
    lut = orc.CubeLut(_lut_cases()[name])
    assert lut.ok, lut.error
    w, h = 257, 9
    wide = fmt != "RGBA"
    bpp = 8 if wide else 4
    stride = w * bpp + 16
    src = frames.random_frame(0x5EED0500, w, h, bpp, stride)
    dst = np.full((h, stride), 0xC3, np.uint8)
    assert lut.apply(src, stride, dst, stride, w, h, fmt) == 0
    if wide:
        dt = "<u2" if fmt == "RGBA64_LE" else ">u2"
        px = src[:, :w * 8].copy().view(dt).reshape(h * w, 4)
        maxv = 65535
    else:
        px = src[:, :w * 4].reshape(h * w, 4)
        maxv = 255
    vals = px[:, :3].astype(np.int64)
    if lut.is_3d:
        out = np_twin.colorlut_3d(vals, maxv, lut.rgba(), lut.size, lut.domain_scale, lut.domain_offset)
    else:
        out = np_twin.colorlut_1d(vals, maxv, [lut.table(c) for c in range(3)], lut.size, lut.domain_scale, lut.domain_offset)
    exp = np.full((h, stride), 0xC3, np.uint8)
    if wide:
        o = np.empty((h * w, 4), dtype=dt)
        o[:, :3] = out
        o[:, 3] = px[:, 3]
        exp[:, :w * 8] = o.view(np.uint8).reshape(h, w * 8)
    else:
        o = np.empty((h * w, 4), np.uint8)
        o[:, :3] = out
        o[:, 3] = px[:, 3]
        exp[:, :w * 4] = o.reshape(h, w * 4)
    assert np.array_equal(dst, exp)


def test_colorlut_identity_is_identity_u8():
    """an identity cube maps every 8-bit grey-ish sample to itself (sanity of lattice maths)"""
    lut = orc.CubeLut(cubes.identity_3d(2, 1))
    src = frames.random_frame(3, 64, 4)
    dst = np.empty_like(src)
    assert lut.apply(src, 256, dst, 256, 64, 4, "RGBA") == 0
    assert np.array_equal(src, dst)


# ---------------------------------------------------------------- videofx (third-party pins)

def test_colordetect_solid_red_is_red():
    """tests/colordetect.rs:21-68: videotestsrc pattern=red -> dominant-color 'red'"""
    for fmt, px in (("RGBA", (255, 0, 0, 255)), ("RGB", (255, 0, 0)), ("BGR", (0, 0, 255)), ("ARGB", (255, 255, 0, 0)),
                    ("BGRA", (0, 0, 255, 255))):
        frame = np.tile(np.array(px, np.uint8), 320 * 240).reshape(240, -1)
        rc, palette = orc.colordetect_palette(frame, fmt, 10, 2)
        assert rc == 2
        r, g, b = (palette[0] >> 16) & 255, (palette[0] >> 8) & 255, palette[0] & 255
        assert (r, g, b) == (252, 4, 4)
        assert orc.css_similar(r, g, b) == "red"


def test_blockhash_identical_is_zero_and_snow_differs():
    """tests/videocompare.rs:57-139: red vs red -> distance 0; snow vs red -> > 0"""
    w, h = 320, 240
    red = np.tile(np.array((255, 0, 0, 255), np.uint8), w * h).reshape(h, w * 4)
    snow = frames.random_frame(0x5EED0600, w, h)
    snow[:, 3::4] = 255
    rc, hr = orc.blockhash(red, w, h, w * 4, "RGBA")
    rc2, hs = orc.blockhash(snow, w, h, w * 4, "RGBA")
    assert rc == 0 and rc2 == 0
    assert orc.hamming(hr, hr) == 0
    assert orc.hamming(hr, hs) > 0


def test_blockhash_sums_match_numpy():
    w, h = 64, 48
    f = frames.random_frame(9, w, h, 4, w * 4 + 8)
    f[::7, 3:w * 4:4] = 0  # some fully transparent pixels count as 765
    rc, sums = orc.blockhash_sums(f, w, h, w * 4 + 8, "RGBA")
    assert rc == 0
    px = f[:, :w * 4].reshape(h, w, 4).astype(np.uint32)
    v = px[..., 0] + px[..., 1] + px[..., 2]
    v[px[..., 3] == 0] = 765
    exp = v.reshape(8, h // 8, 8, w // 8).sum(axis=(1, 3)).reshape(-1)
    assert np.array_equal(sums, exp.astype(np.uint32))
