"""videocompare hash-algo = mean / gradient / vertgradient / doublegradient on the GPU (csrc/imghash_kernels.hip)
against the C oracle (oracle/videofx_oracle.c) and the committed self-golden vectors: bit-exact resized bytes and
hash bits, every size incl. the BASELINE 8K frame.  (image_hasher / image are not under /root/reference: the oracle
itself is parity-unpinned against the crates; see tests/test_imghash_oracle_cpu.py.)"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu

ALGOS = ["mean", "gradient", "vertgradient", "doublegradient"]


def _resize_gpu(gpu, f, w, h, stride, fmt, nw, nh):
    buf = gpu.DeviceBuffer(max(f.nbytes, 16)).upload(f)
    fr = gpu.make_frame(buf.ptr, w, h, stride, fmt)
    out = np.zeros((nh, nw), np.uint8)
    gpu.check(gpu.lib().mvfx_image_gray_resize_lanczos3(ctypes.byref(fr), nw, nh, out.ctypes.data, None))
    return out


@pytest.mark.parametrize("fmt,bpp", [("RGBA", 4), ("RGB", 3)])
@pytest.mark.parametrize("geom", [(64, 48, 0), (100, 37, 8), (8, 8, 0), (9, 8, 0), (5, 5, 3), (3, 2, 0), (1, 1, 0), (200, 11, 0),
                                  (640, 480, 0), (1920, 1080, 0), (333, 2500, 1), (5000, 9, 0)])
def test_resize_matches_oracle(gpu, fmt, bpp, geom):
    w, h, pad = geom
    stride = w * bpp + pad
    f = frames.random_frame(0x5EED0900 + w * 31 + h, w, h, bpp, stride)
    for (nw, nh) in [(8, 8), (9, 8), (8, 9), (5, 5), (3, 20), (64, 17), (16, 16)]:  # > 16 rows: the per-row kernel
        rc, want = orc.gray_resize_lanczos3(f, w, h, stride, fmt, nw, nh)
        assert rc == 0
        got = _resize_gpu(gpu, f, w, h, stride, fmt, nw, nh)
        assert np.array_equal(got, want), (geom, nw, nh, got, want)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("fmt,bpp", [("RGBA", 4), ("RGB", 3)])
def test_hash_matches_oracle(gpu, algo, fmt, bpp):
    for (w, h, pad) in [(320, 240, 0), (162, 92, 6), (1280, 720, 0)]:
        stride = w * bpp + pad
        f = frames.random_frame(0x5EED0920 + w, w, h, bpp, stride)
        rc, want, n = orc.image_hash(f, w, h, stride, fmt, algo)
        assert rc == 0
        assert gpu.image_hash_host(f.reshape(-1), w, h, stride, fmt, algo) == (want, n)


def test_smooth_content_and_structure(gpu):
    """smooth gradients (every resized byte depends on thousands of roundings in order), a ramp, a solid frame"""
    w, h = 1920, 1080
    y, x = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w, 4), np.uint8)
    f[..., 0] = (127 + 120 * np.sin(x / 97.0 + y / 211.0)).astype(np.uint8)
    f[..., 1] = (x * 255 // (w - 1)).astype(np.uint8)
    f[..., 2] = (y * 255 // (h - 1)).astype(np.uint8)
    f[..., 3] = 255
    f = f.reshape(h, w * 4)
    for algo in ALGOS:
        rc, want, n = orc.image_hash(f, w, h, w * 4, "RGBA", algo)
        assert gpu.image_hash_host(f.reshape(-1), w, h, w * 4, "RGBA", algo) == (want, n)
    red = np.tile(np.array((255, 0, 0, 255), np.uint8), w * h).reshape(h, w * 4)
    assert gpu.image_hash_host(red.reshape(-1), w, h, w * 4, "RGBA", "mean") == ((1 << 64) - 1, 64)
    assert gpu.image_hash_host(red.reshape(-1), w, h, w * 4, "RGBA", "doublegradient") == (0, 40)


def test_8k_pair_distance(gpu):
    """BASELINE config 5 shape with the resize hashes: 7680x4320 RGBA, A vs A = 0, A vs perturbed = the oracle's"""
    w, h = 7680, 4320
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[h // 3: h // 2, : w * 2] //= 2
    da = gpu.DeviceBuffer(a.nbytes).upload(a)
    db = gpu.DeviceBuffer(b.nbytes).upload(b)
    fa = gpu.make_frame(da.ptr, w, h, w * 4, "RGBA")
    fb = gpu.make_frame(db.ptr, w, h, w * 4, "RGBA")
    d = ctypes.c_double()
    for algo in ("gradient", "mean"):
        rc, ha, _ = orc.image_hash(a, w, h, w * 4, "RGBA", algo)
        rc, hb, _ = orc.image_hash(b, w, h, w * 4, "RGBA", algo)
        gpu.check(gpu.lib().mvfx_videocompare_distance_algo(ctypes.byref(fa), ctypes.byref(fa), gpu.HASH_ALGOS[algo], ctypes.byref(d), None))
        assert d.value == 0.0
        gpu.check(gpu.lib().mvfx_videocompare_distance_algo(ctypes.byref(fa), ctypes.byref(fb), gpu.HASH_ALGOS[algo], ctypes.byref(d), None))
        assert d.value == float(orc.hamming(ha, hb))
        h1 = ctypes.c_uint64()
        gpu.check(gpu.lib().mvfx_image_hash(ctypes.byref(fb), gpu.HASH_ALGOS[algo], ctypes.byref(h1), None, None))
        assert h1.value == hb


def test_self_golden_vectors(gpu):
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "imghash_kat.json")))
    for case in kat["cases"]:
        f = frames.random_frame(case["seed"], case["width"], case["height"], case["bpp"], case["stride"])
        for algo in ALGOS:
            hv, n = gpu.image_hash_host(f.reshape(-1), case["width"], case["height"], case["stride"], case["format"], algo)
            assert f"{hv:016x}" == case["hash"][algo], (case, algo)


def test_errors_and_dispatch(gpu):
    w, h = 64, 48
    f = frames.random_frame(1, w, h)
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    fr = gpu.make_frame(buf.ptr, w, h, w * 4, "RGBA")
    hv = ctypes.c_uint64()
    assert gpu.lib().mvfx_image_hash(ctypes.byref(fr), 4, ctypes.byref(hv), None, None) == gpu.ERR_INVALID_ARGUMENT  # blockhash has its own entry
    bad = gpu.make_frame(buf.ptr, w, h, w * 4, "BGRA")
    assert gpu.lib().mvfx_image_hash(ctypes.byref(bad), 0, ctypes.byref(hv), None, None) == gpu.ERR_UNSUPPORTED_FORMAT
    d = ctypes.c_double()
    half = gpu.make_frame(buf.ptr, w, h // 2, w * 4, "RGBA")
    assert gpu.lib().mvfx_videocompare_distance_algo(ctypes.byref(fr), ctypes.byref(half), 1, ctypes.byref(d), None) == gpu.ERR_NOT_NEGOTIATED
    # algo 4 / 5 dispatch to the blockhash / dssim paths
    gpu.check(gpu.lib().mvfx_videocompare_distance_algo(ctypes.byref(fr), ctypes.byref(fr), 4, ctypes.byref(d), None))
    assert d.value == 0.0
    gpu.check(gpu.lib().mvfx_videocompare_distance_algo(ctypes.byref(fr), ctypes.byref(fr), 5, ctypes.byref(d), None))
    assert d.value == 0.0
