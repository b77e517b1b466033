"""videocompare hash-algo = mean / gradient / vertgradient / doublegradient (image_hasher 3.1.1 on image 0.25.10,
hashed_image.rs:89-107): the C oracle against an independent numpy-f32 restatement, against properties of the
algorithms, and against the committed self-golden vectors.  PARITY UNPINNED against the crates (not under
/root/reference); the only reference pins are the pipeline tests (identical frames -> distance 0)."""
import json
import os

import numpy as np
import pytest

from tests import frames
from tests import np_twin
from tests import oracle_binding as orc

ALGOS = ["mean", "gradient", "vertgradient", "doublegradient"]
N_BITS = {"mean": 64, "gradient": 64, "vertgradient": 64, "doublegradient": 40}


def _bits_to_int(bits):
    return sum(1 << k for k, b in enumerate(bits) if b)


@pytest.mark.parametrize("fmt,bpp", [("RGBA", 4), ("RGB", 3)])
@pytest.mark.parametrize("geom", [(64, 48, 0), (100, 37, 8), (8, 8, 0), (9, 8, 0), (5, 5, 3), (3, 2, 0), (1, 1, 0), (200, 11, 0)])
def test_resize_matches_numpy_twin(fmt, bpp, geom):
    w, h, pad = geom
    stride = w * bpp + pad
    f = frames.random_frame(0x5EED0900 + w * 31 + h, w, h, bpp, stride)
    for (nw, nh) in [(8, 8), (9, 8), (8, 9), (5, 5)]:
        rc, got = orc.gray_resize_lanczos3(f, w, h, stride, fmt, nw, nh)
        assert rc == 0
        want = np_twin.gray_resize_lanczos3(f, w, h, bpp, nw, nh)
        assert np.array_equal(got, want), (geom, nw, nh)


@pytest.mark.parametrize("algo", ALGOS)
def test_hash_bits_match_numpy_twin(algo):
    w, h = 160, 90
    f = frames.random_frame(0x5EED0910, w, h)
    rc, hv, n = orc.image_hash(f, w, h, w * 4, "RGBA", algo)
    assert rc == 0 and n == N_BITS[algo]
    nw, nh = np_twin.HASH_RESIZE[algo]
    px = np_twin.gray_resize_lanczos3(f, w, h, 4, nw, nh)
    assert hv == _bits_to_int(np_twin.image_hash_bits(px, algo))


@pytest.mark.parametrize("algo", ALGOS)
def test_solid_and_identical_frames(algo):
    """tests/videocompare.rs:57-103: identical frames -> distance 0; a solid frame resizes to a solid image:
    mean -> every pixel >= mean (all ones), gradients -> no strict increase (all zeros)"""
    w, h = 320, 240
    red = np.tile(np.array((255, 0, 0, 255), np.uint8), w * h).reshape(h, w * 4)
    rc, hr, n = orc.image_hash(red, w, h, w * 4, "RGBA", algo)
    assert rc == 0
    assert hr == ((1 << 64) - 1 if algo == "mean" else 0)
    rc, hr2, _ = orc.image_hash(red.copy(), w, h, w * 4, "RGBA", algo)
    assert orc.hamming(hr, hr2) == 0
    snow = frames.random_frame(0x5EED0600, w, h)
    rc, hs, _ = orc.image_hash(snow, w, h, w * 4, "RGBA", algo)
    assert rc == 0
    if algo != "mean":
        assert orc.hamming(hr, hs) > 0  # tests/videocompare.rs:105-139 (snow vs red differ)


def test_gray_is_integer_rec709():
    px = np.array([[255, 255, 255, 7, 255, 0, 0, 9, 0, 255, 0, 1, 0, 0, 255, 200, 12, 200, 77, 0]], np.uint8)
    rc, got = orc.gray_resize_lanczos3(px, 5, 1, 20, "RGBA", 5, 1)  # same size: the copy path, no resampling
    assert rc == 0
    assert got.reshape(-1).tolist() == [255, 54, 182, 18, (2126 * 12 + 7152 * 200 + 722 * 77) // 10000]


def test_ramp_gradient_bits():
    """a left-to-right luminance ramp: every horizontal neighbour increases, no vertical one does"""
    w, h = 256, 64
    f = np.zeros((h, w, 4), np.uint8)
    f[..., 0] = f[..., 1] = f[..., 2] = np.arange(w, dtype=np.uint8)[None, :]
    f[..., 3] = 255
    f = f.reshape(h, w * 4)
    rc, hg, _ = orc.image_hash(f, w, h, w * 4, "RGBA", "gradient")
    rc, hv, _ = orc.image_hash(f, w, h, w * 4, "RGBA", "vertgradient")
    assert hg == (1 << 64) - 1 and hv == 0
    rc, hd, n = orc.image_hash(f, w, h, w * 4, "RGBA", "doublegradient")
    assert n == 40 and hd == (1 << 20) - 1


def test_self_golden_vectors():
    """committed vectors made by tests/golden/make_imghash_golden.py (self-golden, upstream-unpinned)"""
    path = os.path.join(os.path.dirname(__file__), "golden", "imghash_kat.json")
    kat = json.load(open(path))
    for case in kat["cases"]:
        f = frames.random_frame(case["seed"], case["width"], case["height"], case["bpp"], case["stride"])
        for algo in ALGOS:
            rc, hv, n = orc.image_hash(f, case["width"], case["height"], case["stride"], case["format"], algo)
            assert rc == 0 and f"{hv:016x}" == case["hash"][algo], (case, algo)


# ---- blockhash on sizes that are not multiples of 8 (image_hasher's f32 `blockhash_slow`) ------------------------
@pytest.mark.parametrize("w,h,bpp", [(1366, 768, 4), (854, 480, 3), (641, 481, 4), (7, 5, 4), (9, 9, 3), (8, 9, 4), (5, 300, 3),
                                     (300, 3, 4), (1, 1, 4), (2001, 1501, 4)])
def test_blockhash_slow_path_oracle_equals_numpy_twin(w, h, bpp):
    """Two separately written restatements agree bit for bit on the f32 sums and on the hash.  2001x1501 pushes every
    block sum past 2^24, where f32 additions round and the ORDER of the chain matters."""
    from tests import np_twin
    fmt = "RGBA" if bpp == 4 else "RGB"
    stride = w * bpp + (5 if bpp == 3 else 8)
    a = frames.random_frame(0xB10C + w, w, h, bpp, stride)
    if bpp == 4:
        a[::3, 3::16] = 0  # some fully transparent pixels count 765
    rc, s = orc.blockhash_sums(a, w, h, stride, fmt)
    assert rc == 0
    got = np.array(s, dtype=np.uint32).view(np.float32)
    want = np_twin.blockhash_slow_sums(a, w, h, bpp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if w == 2001:
        assert got.min() > 2 ** 24  # the order-dependent regime is really exercised
    rc, hh = orc.blockhash(a, w, h, stride, fmt)
    assert rc == 0 and hh == np_twin.blockhash_slow_bits(want, w, h)


def test_blockhash_slow_path_reference_pins():
    """tests/videocompare.rs:57-139 pattern at a non-multiple-of-8 size: identical -> 0, red vs snow -> > 0."""
    w, h = 854, 480
    red = np.zeros((h, w * 4), np.uint8)
    red[:, 0::4] = 255
    red[:, 3::4] = 255
    snow = frames.random_frame(0x5EED, w, h, 4)
    hr = orc.blockhash(red, w, h, w * 4, "RGBA")[1]
    assert orc.hamming(hr, orc.blockhash(red.copy(), w, h, w * 4, "RGBA")[1]) == 0
    assert orc.hamming(hr, orc.blockhash(snow, w, h, w * 4, "RGBA")[1]) > 0
