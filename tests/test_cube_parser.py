"""CPU tests of the .cube parser: the product's host parser (host/cube_parser.cpp, through the
C ABI) and the oracle's parser, both against the reference's own known-answer tests
(video/colorlut/src/parser.rs:377-474) and against each other on edge syntax."""
import math
import os

import numpy as np
import pytest

from tests import cubes
from tests import oracle_binding as orc

# ---- the reference's unit-test inputs (parser.rs:381-473), restated as data ----
KAT_3D_SIZE2 = """
    LUT_3D_SIZE 2

    0.0 0.0 0.0
    1.0 0.0 0.0
    0.0 1.0 0.0
    1.0 1.0 0.0
    0.0 0.0 1.0
    1.0 0.0 1.0
    0.0 1.0 1.0
    1.0 1.0 1.0
"""
KAT_KEYWORD_AFTER_SIZE = """
    LUT_1D_SIZE 2

    TITLE "test"
    DOMAIN_MIN 0.0 0.0 0.0
    DOMAIN_MAX 1.0 1.0 1.0

    0.0 0.0 0.0
    1.0 0.5 0.7
"""
KAT_KEYWORD_AFTER_DATA = """
    LUT_1D_SIZE 2

    0.0 0.0 0.0
    1.0 0.0 0.0
    TITLE "invalid"
"""
KAT_KEYWORD_BETWEEN_DATA = """
    LUT_1D_SIZE 2

    0.0 0.0 0.0
    TITLE "invalid"
    1.0 0.0 0.0
"""
KAT_MULTIPLE_SIZES = """
    LUT_1D_SIZE 2
    LUT_3D_SIZE 2

    0.0 0.0 0.0
    1.0 1.0 1.0
"""


class _Product:
    name = "product"

    def __init__(self, vfx):
        self.vfx = vfx

    def parse(self, text):
        try:
            return self.vfx.CubeLut(text), None
        except self.vfx.MvfxError as e:
            assert e.status == self.vfx.ERR_PARSE
            return None, e.message


class _Oracle:
    name = "oracle"

    def parse(self, text):
        lut = orc.CubeLut(text)
        if not lut.ok:
            return None, lut.error
        lut.domain = lambda: (lut.domain_scale, lut.domain_offset)
        return lut, None


@pytest.fixture(params=["product", "oracle"])
def parser(request, vfx):
    return _Product(vfx) if request.param == "product" else _Oracle()


def test_kat_parse_3d_lut(parser):
    """parser.rs:381-408"""
    lut, err = parser.parse(KAT_3D_SIZE2)
    assert err is None
    assert lut.is_3d and lut.size == 2
    flat = lut.rgba()
    assert flat.shape == (8, 4)
    assert flat[0].tolist() == [0.0, 0.0, 0.0, 1.0]              # at(0,0,0)
    assert flat[1 + 1 * 2 + 1 * 4].tolist() == [1.0, 1.0, 1.0, 1.0]  # at(1,1,1)
    assert flat[1].tolist() == [1.0, 0.0, 0.0, 1.0]              # R is the fastest index (parser.rs:43-53)


def test_kat_keyword_after_lut_size(parser):
    """parser.rs:410-434"""
    lut, err = parser.parse(KAT_KEYWORD_AFTER_SIZE)
    assert err is None
    assert not lut.is_3d and lut.size == 2
    assert lut.table(0).tolist() == [0.0, 1.0]
    assert lut.table(1).tolist() == [0.0, 0.5]
    assert lut.table(2).tolist() == [0.0, np.float32(0.7)]


@pytest.mark.parametrize("text", [KAT_KEYWORD_AFTER_DATA, KAT_KEYWORD_BETWEEN_DATA, KAT_MULTIPLE_SIZES],
                         ids=["keyword_after_data", "keyword_between_data", "multiple_lut_sizes"])
def test_kat_rejected(parser, text):
    """parser.rs:436-473"""
    lut, err = parser.parse(text)
    assert lut is None and err


ACCEPT = {
    "comments_and_blank": "# c\n\nLUT_1D_SIZE 2\n# mid\n0 0 0\n   \n1 1 1\n",
    "crlf": "LUT_1D_SIZE 2\r\n0 0 0\r\n1 1 1\r\n",
    "rust_float_forms": "LUT_1D_SIZE 2\n1. .5 1e-3\n+1 -0.5 1E+2\n",
    "inf_nan": "LUT_1D_SIZE 2\ninf -Infinity NaN\n0 0 0\n",
    "tabs_and_unicode_space": "LUT_1D_SIZE\t2\n0 0 0\n1 1 1\n",
    "plus_size": "LUT_1D_SIZE +2\n0 0 0\n1 1 1\n",
    "title_without_arg": "TITLE\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n",
    "domain": "LUT_1D_SIZE 2\nDOMAIN_MIN -1 0 0.5\nDOMAIN_MAX 1 2 1.5\n0 0 0\n1 1 1\n",
    "nan_domain_passes": "LUT_1D_SIZE 2\nDOMAIN_MIN nan 0 0\n0 0 0\n1 1 1\n",
    "no_trailing_newline": "LUT_1D_SIZE 2\n0 0 0\n1 1 1",
}
REJECT = {
    "empty": "",
    "missing_size": "TITLE \"x\"\n",
    "data_before_size": "0 0 0\nLUT_1D_SIZE 2\n1 1 1\n",
    "unknown_keyword": "LUT_3D_SIZE 2\nLUT_3D_INPUT_RANGE 0 1\n" + "0 0 0\n" * 8,
    "four_floats": "LUT_1D_SIZE 2\n0 0 0 0\n1 1 1\n",
    "two_floats": "LUT_1D_SIZE 2\n0 0\n1 1 1\n",
    "hex_float": "LUT_1D_SIZE 2\n0x1p0 0 0\n1 1 1\n",
    "trailing_f": "LUT_1D_SIZE 2\n1.0f 0 0\n1 1 1\n",
    "bare_dot": "LUT_1D_SIZE 2\n. 0 0\n1 1 1\n",
    "bare_exp": "LUT_1D_SIZE 2\n1e 0 0\n1 1 1\n",
    "size_too_small": "LUT_1D_SIZE 1\n0 0 0\n",
    "size_1d_too_big": "LUT_1D_SIZE 65537\n",
    "size_3d_too_big": "LUT_3D_SIZE 257\n",
    "negative_size": "LUT_1D_SIZE -2\n0 0 0\n1 1 1\n",
    "size_not_int": "LUT_1D_SIZE 2.0\n0 0 0\n1 1 1\n",
    "size_extra_token": "LUT_1D_SIZE 2 2\n0 0 0\n1 1 1\n",
    "size_missing": "LUT_1D_SIZE\n0 0 0\n1 1 1\n",
    "count_short": "LUT_1D_SIZE 3\n0 0 0\n1 1 1\n",
    "count_long": "LUT_1D_SIZE 2\n0 0 0\n1 1 1\n1 1 1\n",
    "count_3d": "LUT_3D_SIZE 2\n" + "0 0 0\n" * 7,
    "domain_equal": "LUT_1D_SIZE 2\nDOMAIN_MIN 0 0 1\nDOMAIN_MAX 1 1 1\n0 0 0\n1 1 1\n",
    "domain_inverted": "LUT_1D_SIZE 2\nDOMAIN_MIN 0 2 0\n0 0 0\n1 1 1\n",
    "domain_two_values": "LUT_1D_SIZE 2\nDOMAIN_MIN 0 0\n0 0 0\n1 1 1\n",
    "domain_after_data": "LUT_1D_SIZE 2\n0 0 0\nDOMAIN_MAX 1 1 1\n1 1 1\n",
    "lowercase_keyword": "lut_1d_size 2\n0 0 0\n1 1 1\n",
    "thousands_sep": "LUT_1D_SIZE 2\n1,000 0 0\n1 1 1\n",
}


@pytest.mark.parametrize("name", sorted(ACCEPT))
def test_accepts(parser, name):
    lut, err = parser.parse(ACCEPT[name])
    assert err is None, f"{parser.name} rejected {name}: {err}"


@pytest.mark.parametrize("name", sorted(REJECT))
def test_rejects(parser, name):
    lut, err = parser.parse(REJECT[name])
    assert lut is None, f"{parser.name} accepted {name}"


def test_domain_scale_offset(parser):
    """parser.rs:264-274: scale = 1/(max-min), offset = -min*scale, in f32."""
    lut, err = parser.parse(ACCEPT["domain"])
    assert err is None
    scale, offset = lut.domain()
    f = np.float32
    exp_scale = [f(1) / (f(1) - f(-1)), f(1) / (f(2) - f(0)), f(1) / (f(1.5) - f(0.5))]
    exp_off = [-f(-1) * exp_scale[0], -f(0) * exp_scale[1], -f(0.5) * exp_scale[2]]
    assert scale.tolist() == [float(x) for x in exp_scale]
    assert offset.tolist() == [float(x) for x in exp_off]


def test_product_and_oracle_agree_on_generated_cubes(vfx):
    for text in (cubes.identity_3d(5), cubes.analytic_3d(9), cubes.curve_1d(64, ((-0.25, 0, 0.1), (1.5, 1, 0.9)))):
        a = vfx.CubeLut(text)
        b = orc.CubeLut(text)
        assert b.ok and a.is_3d == b.is_3d and a.size == b.size
        sa, oa = a.domain()
        assert sa.tolist() == b.domain_scale.tolist() and oa.tolist() == b.domain_offset.tolist()
        if a.is_3d:
            assert np.array_equal(a.rgba().view(np.uint32), b.rgba().view(np.uint32))
        else:
            for c in range(3):
                assert np.array_equal(a.table(c).view(np.uint32), b.table(c).view(np.uint32))


def test_parse_file_errors(vfx, tmp_path):
    with pytest.raises(vfx.MvfxError) as e:
        vfx.CubeLut(path=str(tmp_path / "missing.cube"))
    assert e.value.status == vfx.ERR_IO
    bad = tmp_path / "bad.cube"
    bad.write_bytes(b"LUT_1D_SIZE 2\n\xff\xfe 0 0\n1 1 1\n")
    with pytest.raises(vfx.MvfxError) as e:
        vfx.CubeLut(path=str(bad))
    assert e.value.status == vfx.ERR_IO  # fs::read_to_string rejects invalid UTF-8
    good = tmp_path / "good.cube"
    good.write_text(KAT_3D_SIZE2)
    lut = vfx.CubeLut(path=str(good))
    assert lut.is_3d and lut.size == 2
    worse = tmp_path / "worse.cube"
    worse.write_text(KAT_MULTIPLE_SIZES)
    with pytest.raises(vfx.MvfxError) as e:
        vfx.CubeLut(path=str(worse))
    assert e.value.status == vfx.ERR_PARSE and "worse.cube" in e.value.message


def test_float_parsing_is_locale_independent(vfx):
    """Rust's str::parse::<f32> knows no locale; gst-launch / GTK apps call setlocale(LC_ALL, "").  Neither parser may
    bind plain strtof (which follows LC_NUMERIC and stops at '.' under a comma-decimal locale): both use strtof_l with
    a "C" locale object.  Where a comma-decimal locale is installed the parse is also exercised under it."""
    import locale
    import subprocess
    for so in (vfx.LIB_PATH, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")):
        syms = subprocess.run(["nm", "-D", "--undefined-only", so], stdout=subprocess.PIPE, text=True, check=True).stdout
        names = {line.split()[-1].split("@")[0] for line in syms.splitlines() if line.strip()}
        assert "strtof_l" in names and "newlocale" in names, so
    text = "LUT_1D_SIZE 2\nDOMAIN_MAX 1.5 1.5 1.5\n0.0 0.25 0.5\n1.0 0.75 1e-1\n"
    saved = locale.setlocale(locale.LC_NUMERIC)
    switched = False
    for cand in ("de_DE.UTF-8", "fr_FR.UTF-8", "ru_RU.UTF-8", "de_DE", "fr_FR"):
        try:
            locale.setlocale(locale.LC_NUMERIC, cand)
            switched = True
            break
        except locale.Error:
            continue
    try:
        lut = vfx.CubeLut(text)
        assert lut.size == 2 and not lut.is_3d
        assert [lut.table(c)[1] for c in range(3)] == [1.0, 0.75, np.float32(0.1)]
    finally:
        locale.setlocale(locale.LC_NUMERIC, saved)
    print("comma-decimal locale exercised:", switched)


# ---- .cube writer: parse(write(lut)) == lut bit for bit, for the product parser AND the oracle parser
WRITER_CASES = {
    "analytic9": lambda: cubes.analytic_3d(9),
    "1d_domain_nan_inf": lambda: "LUT_1D_SIZE 4\nDOMAIN_MIN -0.25 0 0.125\nDOMAIN_MAX 1.5 2 1\n0 0 0\n0.1 0.33333334 1e-7\n0.7 nan -inf\n1 1.00000012 3.4028235e38\n",
    "3d_denormal": lambda: "LUT_3D_SIZE 2\n0 0 0\n1 0 0\n0 1 0\n1 1 0\n0 0 1\n1 0 1\n0 1 1\n0.99999994 1 1e-45\n",
}


@pytest.mark.parametrize("which", list(WRITER_CASES))
def test_cube_writer_round_trips_bit_for_bit(vfx, which):
    text = WRITER_CASES[which]()
    a = vfx.CubeLut(text)
    written = a.write()
    assert written.startswith("LUT_3D_SIZE" if a.is_3d else "LUT_1D_SIZE")
    b = vfx.CubeLut(written)
    assert (a.is_3d, a.size) == (b.is_3d, b.size)
    for u, v in zip(a.domain(), b.domain()):
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
    if a.is_3d:
        assert np.array_equal(a.rgba().view(np.uint32), b.rgba().view(np.uint32))
    else:
        for c in range(3):
            ta, tb = a.table(c), b.table(c)
            assert np.array_equal(np.isnan(ta), np.isnan(tb))
            assert np.array_equal(ta[~np.isnan(ta)].view(np.uint32), tb[~np.isnan(tb)].view(np.uint32))
    o = orc.CubeLut(written)  # the independent parser accepts the writer's text too
    assert o.ok, o.error
    assert o.apply is not None and written.count("\n") == (a.size ** 3 if a.is_3d else a.size) + (3 if "DOMAIN_MIN" in written else 1)
