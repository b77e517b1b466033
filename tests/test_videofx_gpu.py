"""GPU parity tests for the videofx kernels (colordetect histogram + palette, videocompare
blockhash, roundedcorners mask / compose) against the oracle and the libcairo goldens."""
import ctypes
import os

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CD_FORMATS = {"RGB": 3, "RGBA": 4, "ARGB": 4, "BGR": 3, "BGRA": 4}


def _gpu_hist(gpu, frame, w, h, stride, fmt, quality, first=0, n=None):
    buf = gpu.DeviceBuffer(frame.nbytes).upload(frame)
    hist = gpu.DeviceBuffer(32768 * 4)
    mm = gpu.DeviceBuffer(6 * 4)
    f = gpu.make_frame(buf.ptr, w, h, stride, fmt)
    gpu.check(gpu.lib().mvfx_colordetect_histogram(ctypes.byref(f), quality, first, gpu.ALL_SAMPLES if n is None else n,
                                                   ctypes.c_void_p(hist.ptr), ctypes.c_void_p(mm.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    return hist.download(dtype=np.uint32), mm.download(dtype=np.uint32)


@pytest.mark.parametrize("fmt", sorted(CD_FORMATS))
@pytest.mark.parametrize("quality", [1, 3, 10])
def test_colordetect_histogram_matches_oracle(gpu, fmt, quality):
    """whole plane incl. padding, every quality-th pixel, skip rules (a<125, near-white)"""
    bpp = CD_FORMATS[fmt]
    for (w, h, pad) in ((320, 240, 0), (101, 37, 8)):
        stride = (w * bpp + 3) // 4 * 4 + pad
        f = frames.random_frame(0x5EED0700 + w, w, h, bpp, stride)
        f[: h // 4] = 252  # a band of near-white pixels that must be skipped
        rc, hist, mm, n = orc.colordetect_histogram(f, fmt, quality)
        assert rc == 0
        ghist, gmm = _gpu_hist(gpu, f, w, h, stride, fmt, quality)
        assert np.array_equal(ghist, hist.astype(np.uint32))
        assert gmm.tolist() == mm
        assert int(ghist.sum()) == n


def test_colordetect_histogram_sample_ranges_add_up(gpu):
    """two sample ranges (as two ranks would take) sum to the full histogram"""
    w, h = 640, 480
    f = frames.random_frame(0x5EED0701, w, h)
    full, mm = _gpu_hist(gpu, f, w, h, w * 4, "RGBA", 10)
    total = (w * h + 9) // 10
    a, mma = _gpu_hist(gpu, f, w, h, w * 4, "RGBA", 10, 0, total // 2)
    b, mmb = _gpu_hist(gpu, f, w, h, w * 4, "RGBA", 10, total // 2, total - total // 2)
    assert np.array_equal(a + b, full)
    assert [min(mma[0], mmb[0]), max(mma[1], mmb[1])] == mm[:2].tolist()


@pytest.mark.parametrize("settings", [(10, 2), (1, 8), (5, 255), (10, 16)])
def test_colordetect_palette_matches_oracle(gpu, settings):
    quality, max_colors = settings
    w, h = 3840, 2160
    for f in (frames.smpte_like(w, h), frames.random_frame(0x5EED0001, w, h)):
        rc, pal = orc.colordetect_palette(f, "RGBA", quality, max_colors)
        assert rc > 0
        gpal, name = gpu.colordetect_palette_host(f.reshape(-1), w, h, w * 4, "RGBA", quality, max_colors)
        assert gpal == pal
        assert name == orc.css_similar((pal[0] >> 16) & 255, (pal[0] >> 8) & 255, pal[0] & 255)


def test_colordetect_8k_quality1_i32_sums_wrap(gpu):
    """33 M samples in the first box: the crate's i32 colour sums wrap in a release build (8.3e9 > 2^31); the
    median cut on the host and the oracle both carry them as wrapping 32-bit values and must agree."""
    w, h = 7680, 4320
    f = frames.random_frame(0x5EED0700, w, h)
    f[:, 3::4] |= 0x80  # opaque enough for every sample to count
    rc, pal = orc.colordetect_palette(f, "RGBA", 1, 5)
    assert rc > 0
    gpal, _ = gpu.colordetect_palette_host(f.reshape(-1), w, h, w * 4, "RGBA", 1, 5)
    assert gpal == pal


def test_colordetect_histogram_more_samples_than_one_launch_takes(gpu):
    """> 1024 groups x 65535 samples (16-bit bins per group): the second launch must ADD to the histogram and combine the
    bounds.  The top 40 % of the frame is mid-grey-ish (channels 64..191), so the extreme bins and bounds come only from rows
    that fall into the second launch."""
    w, h = 16384, 4200                      # 68.8 M samples at quality 1 > 67.1 M
    f = frames.random_frame(0x5EED0777, w, h)
    top = f[: h * 2 // 5]
    top[:] = 64 + (top >> 1)
    f[:, 3::4] = 255
    rc, hist, mm, n = orc.colordetect_histogram(f, "RGBA", 1)
    assert rc == 0 and n > 1024 * 65535
    ghist, gmm = _gpu_hist(gpu, f, w, h, w * 4, "RGBA", 1)
    assert np.array_equal(ghist, hist.astype(np.uint32))
    assert gmm.tolist() == mm
    assert mm[0] == 0 and mm[1] == 31       # bounds only the bottom rows can produce


@pytest.mark.parametrize("quality", list(range(1, 11)))
def test_colordetect_histogram_every_quality_streaming_and_per_sample_paths(gpu, quality):
    """4-byte formats on a 16-byte aligned plane take the streaming kernel (the whole plane as 16-byte loads, samples picked
    out of the registers: pixel P is a sample iff P % quality == 0); the same plane at an address that is only 4-byte aligned
    takes the per-sample kernel.  Both must equal the oracle for every quality, on a plane whose pixel count is not a multiple
    of 4 (the last partial unit goes to the per-sample kernel in accumulate mode)."""
    w, h, stride = 333, 71, 333 * 4 + 4          # 23927 pixels incl. the padding: % 4 == 3
    f = frames.random_frame(0x5EED0710 + quality, w, h, 4, stride)
    f[5:9] = 253
    rc, hist, mm, n = orc.colordetect_histogram(f, "BGRA", quality)
    assert rc == 0
    ghist, gmm = _gpu_hist(gpu, f, w, h, stride, "BGRA", quality)   # one frame: streaming kernel for quality < 4, per-sample above
    assert np.array_equal(ghist, hist.astype(np.uint32)) and gmm.tolist() == mm
    # two frames per launch: always the streaming kernel (the second frame: the first with another alpha pattern)
    f2 = f.copy()
    f2[:, 3::4] = np.where(f2[:, 3::4] < 40, 200, f2[:, 3::4])
    rc, hist2, mm2, _ = orc.colordetect_histogram(f2, "BGRA", quality)
    b1, b2 = gpu.DeviceBuffer(f.nbytes).upload(f), gpu.DeviceBuffer(f2.nbytes).upload(f2)
    arr = (gpu.Frame * 2)(gpu.make_frame(b1.ptr, w, h, stride, "BGRA"), gpu.make_frame(b2.ptr, w, h, stride, "BGRA"))
    rec2 = gpu.DeviceBuffer(2 * gpu.COLORDETECT_RECORD_WORDS * 4)
    gpu.check(gpu.lib().mvfx_colordetect_histogram_frames(arr, 2, quality, ctypes.c_void_p(rec2.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    r2 = rec2.download(dtype=np.uint32).reshape(2, gpu.COLORDETECT_RECORD_WORDS)
    assert np.array_equal(r2[0, :32768], hist.astype(np.uint32)) and r2[0, 32768:32774].tolist() == mm
    assert np.array_equal(r2[1, :32768], hist2.astype(np.uint32)) and r2[1, 32768:32774].tolist() == mm2
    # the same bytes 4 bytes into a device buffer
    buf = gpu.DeviceBuffer(f.nbytes + 16)
    shifted = np.zeros(f.nbytes + 16, np.uint8)
    shifted[4:4 + f.nbytes] = f.reshape(-1)
    buf.upload(shifted)
    out = gpu.DeviceBuffer(gpu.COLORDETECT_RECORD_WORDS * 4)
    fr = gpu.make_frame(buf.ptr + 4, w, h, stride, "BGRA")
    gpu.check(gpu.lib().mvfx_colordetect_histogram(ctypes.byref(fr), quality, 0, gpu.ALL_SAMPLES, ctypes.c_void_p(out.ptr),
                                                   ctypes.c_void_p(out.ptr + 32768 * 4), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    rec = out.download(dtype=np.uint32)
    assert np.array_equal(rec[:32768], hist.astype(np.uint32)) and rec[32768:32774].tolist() == mm


@pytest.mark.parametrize("n_frames,quality,size", [(5, 10, (640, 480)), (34, 3, (320, 180)), (16, 10, (1920, 1080)), (3, 1, (641, 481))])
def test_colordetect_histogram_frames_equals_single_frame_calls(gpu, n_frames, quality, size):
    """mvfx_colordetect_histogram_frames: one frame of each of n streams in one pair of launches (blockIdx.y = stream; > 32
    frames are split) = the oracle's histogram and bounds of every frame, record by record."""
    w, h = size
    fr_host = [frames.random_frame(0x5EED0720 + k, w, h) for k in range(n_frames)]
    for k, f in enumerate(fr_host):
        f[:, 3::4] |= 0x40 * (k & 3)                      # different alpha statistics per frame
        f[: (k % 5) * h // 10] >>= (k % 3)                  # different colour ranges per frame (bounds differ)
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in fr_host]
    arr = (gpu.Frame * n_frames)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
    out = gpu.DeviceBuffer(n_frames * gpu.COLORDETECT_RECORD_WORDS * 4)
    gpu.check(gpu.lib().mvfx_colordetect_histogram_frames(arr, n_frames, quality, ctypes.c_void_p(out.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    rec = out.download(dtype=np.uint32).reshape(n_frames, gpu.COLORDETECT_RECORD_WORDS)
    for k, f in enumerate(fr_host):
        rc, hist, mm, n = orc.colordetect_histogram(f, "RGBA", quality)
        assert rc == 0
        assert np.array_equal(rec[k, :32768], hist.astype(np.uint32)), f"frame {k}"
        assert rec[k, 32768:32774].tolist() == mm, f"frame {k}"


def test_colordetect_histogram_frames_rejects_mixed_geometry(gpu):
    b = gpu.DeviceBuffer(64 * 64 * 4)
    arr = (gpu.Frame * 2)(gpu.make_frame(b.ptr, 64, 64, 256, "RGBA"), gpu.make_frame(b.ptr, 32, 64, 256, "RGBA"))
    out = gpu.DeviceBuffer(2 * gpu.COLORDETECT_RECORD_WORDS * 4)
    assert gpu.lib().mvfx_colordetect_histogram_frames(arr, 2, 10, ctypes.c_void_p(out.ptr), None) == gpu.ERR_INVALID_ARGUMENT


def test_colordetect_two_streams_of_one_thread_do_not_share_partials(gpu):
    """The partial histograms live in scratch keyed by the stream: two asynchronous calls of ONE thread on two streams, no
    synchronisation in between, each produce their own frame's histogram."""
    w, h = 1920, 1080
    fa, fb = frames.random_frame(0x5EED0730, w, h), frames.smpte_like(w, h)
    ba, bb = gpu.DeviceBuffer(fa.nbytes).upload(fa), gpu.DeviceBuffer(fb.nbytes).upload(fb)
    oa, ob = gpu.DeviceBuffer(gpu.COLORDETECT_RECORD_WORDS * 4), gpu.DeviceBuffer(gpu.COLORDETECT_RECORD_WORDS * 4)
    s1, s2 = gpu.lib().mvfx_thread_stream(), None    # the thread's non-blocking stream and the null stream
    fra, frb = gpu.make_frame(ba.ptr, w, h, w * 4, "RGBA"), gpu.make_frame(bb.ptr, w, h, w * 4, "RGBA")
    for _ in range(8):
        gpu.check(gpu.lib().mvfx_colordetect_histogram(ctypes.byref(fra), 10, 0, gpu.ALL_SAMPLES, ctypes.c_void_p(oa.ptr),
                                                       ctypes.c_void_p(oa.ptr + 32768 * 4), ctypes.c_void_p(s1)))
        gpu.check(gpu.lib().mvfx_colordetect_histogram(ctypes.byref(frb), 10, 0, gpu.ALL_SAMPLES, ctypes.c_void_p(ob.ptr),
                                                       ctypes.c_void_p(ob.ptr + 32768 * 4), s2))
    gpu.check(gpu.lib().mvfx_stream_synchronize(ctypes.c_void_p(s1)))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for f, o in ((fa, oa), (fb, ob)):
        rc, hist, mm, n = orc.colordetect_histogram(f, "RGBA", 10)
        rec = o.download(dtype=np.uint32)
        assert np.array_equal(rec[:32768], hist.astype(np.uint32)) and rec[32768:32774].tolist() == mm


def test_colordetect_reference_pin_red(gpu):
    """tests/colordetect.rs:21-68: solid red => dominant-color 'red' (palette[0] = 252,4,4)"""
    w, h = 320, 240
    red = np.tile(np.array((255, 0, 0, 255), np.uint8), w * h).reshape(h, w * 4)
    pal, name = gpu.colordetect_palette_host(red.reshape(-1), w, h, w * 4, "RGBA", 10, 2)
    assert name == "red" and pal[0] == 0xFC0404


def test_colordetect_errors(gpu):
    a = np.zeros(64, np.uint8)
    pal = (ctypes.c_uint32 * 4)()
    n = ctypes.c_uint32()
    f = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBA")
    L = gpu.lib()
    assert L.mvfx_colordetect_palette_host(ctypes.byref(f), 0, 2, pal, ctypes.byref(n)) == gpu.ERR_REFERENCE_PANIC  # quality=0
    assert L.mvfx_colordetect_palette_host(ctypes.byref(f), 10, 1, pal, ctypes.byref(n)) == gpu.ERR_REFERENCE_PANIC
    fx = gpu.make_frame(a.ctypes.data, 4, 4, 16, "RGBx")
    assert L.mvfx_colordetect_palette_host(ctypes.byref(fx), 10, 2, pal, ctypes.byref(n)) == gpu.ERR_UNSUPPORTED_FORMAT


# ---------------------------------------------------------------- videocompare

@pytest.mark.parametrize("fmt", ["RGBA", "RGB"])
@pytest.mark.parametrize("geom", [(640, 480, 0), (64, 48, 12), (1920, 1080, 0), (8, 8, 0), (72, 40, 4), (3840, 2160, 0),
                                  (32, 1000, 0), (8224, 16, 16), (96, 8, 2)])
def test_blockhash_sums_and_hash_match_oracle(gpu, fmt, geom):
    w, h, pad = geom
    bpp = 3 if fmt == "RGB" else 4
    stride = (w * bpp + 3) // 4 * 4 + pad
    f = frames.random_frame(0x5EED0800 + w, w, h, bpp, stride)
    if bpp == 4:
        f[::5, 3:w * 4:8] = 0  # transparent pixels count as 765
    rc, sums = orc.blockhash_sums(f, w, h, stride, fmt)
    assert rc == 0
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    dsums = gpu.DeviceBuffer(256)
    fr = gpu.make_frame(buf.ptr, w, h, stride, fmt)
    gpu.check(gpu.lib().mvfx_blockhash_sums(ctypes.byref(fr), 0, h, ctypes.c_void_p(dsums.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    assert np.array_equal(dsums.download(dtype=np.uint32), sums)
    rc, hh = orc.blockhash(f, w, h, stride, fmt)
    assert gpu.blockhash_host(f.reshape(-1), w, h, stride, fmt) == hh


def test_blockhash_row_bands_add_up(gpu):
    """8 row bands (one block-row per rank) sum to the full-frame block sums"""
    w, h = 7680, 4320 // 4  # keeps the test light: 8 block rows of 135 rows
    f = frames.random_frame(0x5EED0801, w, h)
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    fr = gpu.make_frame(buf.ptr, w, h, w * 4, "RGBA")
    total = np.zeros(64, np.uint64)
    d = gpu.DeviceBuffer(256)
    for r in range(8):
        gpu.check(gpu.lib().mvfx_blockhash_sums(ctypes.byref(fr), r * h // 8, (r + 1) * h // 8, ctypes.c_void_p(d.ptr), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        part = d.download(dtype=np.uint32)
        assert not part[:r * 8].any() and not part[(r + 1) * 8:].any()
        total += part
    rc, sums = orc.blockhash_sums(f, w, h, w * 4, "RGBA")
    assert np.array_equal(total.astype(np.uint32), sums)


def test_blockhash_band_only_buffers(gpu):
    """pre-sharded input: each 'rank' holds only its band (mvfx_blockhash_sums_band)"""
    w, h = 640, 480
    f = frames.random_frame(0x5EED0802, w, h)
    total = np.zeros(64, np.uint64)
    d = gpu.DeviceBuffer(256)
    for r in range(4):
        r0, r1 = h * r // 4, h * (r + 1) // 4
        band = np.ascontiguousarray(f[r0:r1])
        buf = gpu.DeviceBuffer(band.nbytes).upload(band)
        fr = gpu.make_frame(buf.ptr, w, r1 - r0, w * 4, "RGBA")
        gpu.check(gpu.lib().mvfx_blockhash_sums_band(ctypes.byref(fr), h, r0, ctypes.c_void_p(d.ptr), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        total += d.download(dtype=np.uint32)
    rc, sums = orc.blockhash_sums(f, w, h, w * 4, "RGBA")
    assert np.array_equal(total.astype(np.uint32), sums)


@pytest.mark.parametrize("fmt", ["RGBA", "RGB"])
def test_blockhash_multi_pad_launch(gpu, fmt):
    """aggregate_frames hashes every pad per output buffer (videocompare/imp.rs:316,349-353): n pads in one launch,
    per-pad strides, also as bands of a taller frame"""
    w, h, n = 256, 64, 5
    bpp = 3 if fmt == "RGB" else 4
    strides = [w * bpp + 16 * (p % 3) for p in range(n)]
    fs = [frames.random_frame(0x5EED0810 + p, w, h, bpp, strides[p]) for p in range(n)]
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in fs]
    arr = (gpu.Frame * n)(*[gpu.make_frame(bufs[p].ptr, w, h, strides[p], fmt) for p in range(n)])
    d = gpu.DeviceBuffer(n * 256)
    gpu.check(gpu.lib().mvfx_blockhash_sums_pads(arr, n, h, 0, ctypes.c_void_p(d.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = d.download(dtype=np.uint32).reshape(n, 64)
    for p in range(n):
        rc, sums = orc.blockhash_sums(fs[p], w, h, strides[p], fmt)
        assert rc == 0 and np.array_equal(got[p], sums)
    # bands: rows 16..40 of every pad, each in its own buffer
    r0, r1 = 16, 40
    bands = [np.ascontiguousarray(f[r0:r1]) for f in fs]
    bbufs = [gpu.DeviceBuffer(b.nbytes).upload(b) for b in bands]
    barr = (gpu.Frame * n)(*[gpu.make_frame(bbufs[p].ptr, w, r1 - r0, strides[p], fmt) for p in range(n)])
    gpu.check(gpu.lib().mvfx_blockhash_sums_pads(barr, n, h, r0, ctypes.c_void_p(d.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = d.download(dtype=np.uint32).reshape(n, 64)
    for p in range(n):
        masked = fs[p].copy()
        rc, full = orc.blockhash_sums(masked, w, h, strides[p], fmt)
        masked[r0:r1] = 0
        if bpp == 4:
            masked[r0:r1, 3::4] = 255  # alpha 0 would count as white
        rc, rest = orc.blockhash_sums(masked, w, h, strides[p], fmt)
        assert np.array_equal(got[p], full - rest)
    # sizes must match (videocompare/imp.rs:337-346)
    arr[2].height = h // 2
    assert gpu.lib().mvfx_blockhash_sums_pads(arr, n, h, 0, ctypes.c_void_p(d.ptr), None) == gpu.ERR_NOT_NEGOTIATED


def test_videocompare_reference_pins(gpu):
    """tests/videocompare.rs:57-139: red vs red -> 0; snow vs red -> > 0; perturbation ladder"""
    w, h = 320, 240
    red = np.tile(np.array((255, 0, 0, 255), np.uint8), w * h).reshape(h, w * 4)
    snow = frames.random_frame(0x5EED0600, w, h)
    snow[:, 3::4] = 255
    a = gpu.DeviceBuffer(red.nbytes).upload(red)
    b = gpu.DeviceBuffer(red.nbytes).upload(red)
    c = gpu.DeviceBuffer(snow.nbytes).upload(snow)
    fa, fb, fc = (gpu.make_frame(x.ptr, w, h, w * 4, "RGBA") for x in (a, b, c))
    d = ctypes.c_double()
    gpu.check(gpu.lib().mvfx_videocompare_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), None))
    assert d.value == 0.0
    gpu.check(gpu.lib().mvfx_videocompare_distance(ctypes.byref(fa), ctypes.byref(fc), ctypes.byref(d), None))
    rc, hr = orc.blockhash(red, w, h, w * 4, "RGBA")
    rc, hs = orc.blockhash(snow, w, h, w * 4, "RGBA")
    assert d.value == float(orc.hamming(hr, hs)) and d.value > 0
    f2 = gpu.make_frame(c.ptr, w, h // 2, w * 4, "RGBA")
    assert gpu.lib().mvfx_videocompare_distance(ctypes.byref(fa), ctypes.byref(f2), ctypes.byref(d), None) == gpu.ERR_NOT_NEGOTIATED


def test_blockhash_8k_pair(gpu):
    """BASELINE config 5 shape: 7680x4320 RGBA pair, A vs A = 0, A vs perturbed/inverted"""
    w, h = 7680, 4320
    a = frames.random_frame(0x5EED0001, w, h)
    a[:, 3::4] |= 1
    rc, ha = orc.blockhash(a, w, h, w * 4, "RGBA")
    assert gpu.blockhash_host(a.reshape(-1), w, h, w * 4, "RGBA") == ha
    inv = a.copy()
    inv[:, 0::4] = 255 - inv[:, 0::4]
    inv[:, 1::4] = 255 - inv[:, 1::4]
    inv[:, 2::4] = 255 - inv[:, 2::4]
    rc, hi = orc.blockhash(inv, w, h, w * 4, "RGBA")
    assert gpu.blockhash_host(inv.reshape(-1), w, h, w * 4, "RGBA") == hi


# ---------------------------------------------------------------- roundedcorners

MASK_CASES = ["w32_h24_r8", "w64_h48_r10", "w641_h481_r33", "w640_h480_r0", "w33_h17_r40", "w1920_h1080_r1",
              "w1920_h1080_r50", "w1920_h1080_r540", "w3840_h2160_r100", "w64_h48_r30", "w1920_h1080_r700", "w100_h101_r4000"]


@pytest.fixture(scope="module")
def cairo_masks():
    return np.load(os.path.join(GOLDEN, "roundedcorners_masks.npz"))


@pytest.mark.parametrize("case", MASK_CASES)
def test_roundedcorners_mask_vs_cairo_golden(gpu, cairo_masks, case):
    """The A8 plane in HBM is byte-identical to libcairo's rendering of border/imp.rs:57-106 for every radius,
    2*radius > min(width, height) included (no tolerance: the mask is produced by the reference's own rasteriser)."""
    gold = cairo_masks[case]
    w, h, r = (int(t[1:]) for t in case.split("_"))
    rows, stride = gold.shape
    buf = gpu.DeviceBuffer(gold.nbytes)
    buf.upload(np.full(gold.nbytes, 0x77, np.uint8))
    gpu.check(gpu.lib().mvfx_roundedcorners_mask(ctypes.c_void_p(buf.ptr), w, h, stride, r, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = buf.download().reshape(rows, stride)
    assert np.array_equal(got, gold)


@pytest.mark.parametrize("w,h,r", [(64, 48, 10), (3840, 2160, 100), (641, 481, 33)])
def test_roundedcorners_compose_a420_sizes(gpu, cairo_masks, w, h, r):
    """BASELINE config 4 shape (3840x2160) and an odd size: Y/U/V copied bit-exactly into the A420 buffer with
    GStreamer's plane layout, plane 3 == the cairo mask (prepare_output_buffer, border/imp.rs:482-559)."""
    gold = cairo_masks[f"w{w}_h{h}_r{r}"]
    ys, cs = (w + 3) & ~3, (((w + 1) // 2) + 3) & ~3      # GstVideoInfo strides of I420 / A420
    hh, ch = (h + 1) & ~1, ((h + 1) & ~1) // 2            # round_up_2(height) rows of Y, half of that of chroma
    offs = [0, ys * hh, ys * hh + cs * ch, ys * hh + 2 * cs * ch]
    i420 = frames.splitmix64_bytes(0x5EED0100 + w, offs[3])
    src = gpu.DeviceBuffer(i420.nbytes).upload(i420)
    mask = gpu.DeviceBuffer(gold.nbytes)
    gpu.check(gpu.lib().mvfx_roundedcorners_mask(ctypes.c_void_p(mask.ptr), w, h, ys, r, None))
    out_size = offs[3] + ys * hh
    dst = gpu.DeviceBuffer(out_size).upload(np.full(out_size, 0x5A, np.uint8))
    fi, fo = gpu.PlanarFrame(), gpu.PlanarFrame()
    for p in range(3):
        fi.data[p] = src.ptr + offs[p]
        fo.data[p] = dst.ptr + offs[p]
        fi.stride[p] = fo.stride[p] = ys if p == 0 else cs
    fo.data[3] = dst.ptr + offs[3]
    fo.stride[3] = ys
    fi.width = fo.width = w
    fi.height = fo.height = h
    fi.format, fo.format = gpu.FORMATS["I420"], gpu.FORMATS["A420"]
    gpu.check(gpu.lib().mvfx_roundedcorners_compose_a420(ctypes.byref(fi), ctypes.c_void_p(mask.ptr), ys, ctypes.byref(fo), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    out = dst.download()
    cw, chh = (w + 1) // 2, (h + 1) // 2
    y_in, y_out = i420[:ys * hh].reshape(hh, ys), out[:ys * hh].reshape(hh, ys)
    assert np.array_equal(y_out[:h, :w], y_in[:h, :w])
    for p in (1, 2):
        c_in = i420[offs[p]:offs[p] + cs * ch].reshape(ch, cs)
        c_out = out[offs[p]:offs[p] + cs * ch].reshape(ch, cs)
        assert np.array_equal(c_out[:chh, :cw], c_in[:chh, :cw])
    a_out = out[offs[3]:].reshape(hh, ys)
    assert np.array_equal(a_out[:h, :w], gold[:h, :w])
    # row padding of the destination is not written
    if ys > w:
        assert (y_out[:h, w:] == 0x5A).all() and (a_out[:h, w:] == 0x5A).all()


def test_roundedcorners_compose_a420(gpu, cairo_masks):
    """I420 -> A420: planes copied bit-exactly, plane 3 = mask"""
    w, h, r = 64, 48, 10
    ys, cs = 64, 32
    i420 = frames.splitmix64_bytes(5, ys * h + 2 * cs * (h // 2))
    src = gpu.DeviceBuffer(i420.nbytes).upload(i420)
    mask = gpu.DeviceBuffer(64 * 48)
    gpu.check(gpu.lib().mvfx_roundedcorners_mask(ctypes.c_void_p(mask.ptr), w, h, 64, r, None))
    out_size = ys * h + 2 * cs * (h // 2) + 64 * h
    dst = gpu.DeviceBuffer(out_size)
    fi = gpu.PlanarFrame()
    fo = gpu.PlanarFrame()
    offs = [0, ys * h, ys * h + cs * (h // 2), ys * h + 2 * cs * (h // 2)]
    for p in range(3):
        fi.data[p] = src.ptr + offs[p]
        fo.data[p] = dst.ptr + offs[p]
        fi.stride[p] = fo.stride[p] = ys if p == 0 else cs
    fo.data[3] = dst.ptr + offs[3]
    fo.stride[3] = 64
    fi.width = fo.width = w
    fi.height = fo.height = h
    fi.format, fo.format = gpu.FORMATS["I420"], gpu.FORMATS["A420"]
    gpu.check(gpu.lib().mvfx_roundedcorners_compose_a420(ctypes.byref(fi), ctypes.c_void_p(mask.ptr), 64, ctypes.byref(fo), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    out = dst.download()
    assert np.array_equal(out[:offs[3]], i420)
    assert np.array_equal(out[offs[3]:], mask.download())


# ---------------------------------------------------------------- blockhash, sizes that are not multiples of 8
# image_hasher's f32 `blockhash_slow` (hashed_image.rs:24-45 accepts any size): 64 ordered f32 chains on the device.

@pytest.mark.parametrize("w,h,fmt,bpp", [(1366, 768, "RGBA", 4), (854, 480, "RGB", 3), (641, 481, "RGBA", 4), (1921, 1081, "RGBA", 4),
                                         (2001, 1501, "RGB", 3), (3841, 2161, "RGBA", 4), (7, 5, "RGBA", 4), (9, 9, "RGB", 3), (8, 9, "RGBA", 4), (5, 300, "RGB", 3),
                                         (300, 3, "RGBA", 4), (1, 1, "RGBA", 4), (1366, 5, "RGBA", 4)])
def test_blockhash_slow_path_sums_and_hash_match_oracle(gpu, w, h, fmt, bpp):
    stride = w * bpp + (5 if bpp == 3 else 12)   # padded, unaligned rows
    a = frames.random_frame(0xB10C + w, w, h, bpp, stride)
    if bpp == 4:
        a[::3, 3::16] = 0
    rc, want = orc.blockhash_sums(a, w, h, stride, fmt)
    assert rc == 0
    d = gpu.DeviceBuffer(a.nbytes).upload(a)
    sums = gpu.DeviceBuffer(64 * 4)
    f = gpu.make_frame(d.ptr, w, h, stride, fmt)
    gpu.check(gpu.lib().mvfx_blockhash_sums(ctypes.byref(f), 0, h, ctypes.c_void_p(sums.ptr), None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    got = sums.download(dtype=np.uint32)
    assert np.array_equal(got, np.array(want, dtype=np.uint32)), "f32 block sums differ in their bit patterns"
    if (w, h) in ((2001, 1501), (3841, 2161)):
        assert got.view(np.float32).min() > 2 ** 24  # past the exact regime: the order of the additions matters
    rc, hh = orc.blockhash(a, w, h, stride, fmt)
    assert gpu.blockhash_host(a.reshape(-1), w, h, stride, fmt) == hh
    out = ctypes.c_uint64()
    gpu.check(gpu.lib().mvfx_blockhash_bits((ctypes.c_uint32 * 64)(*[int(x) for x in got]), w, h, ctypes.byref(out)))
    assert out.value == hh


def test_blockhash_slow_path_pair_distance_and_errors(gpu):
    """mvfx_videocompare_distance on 854x480 (854 % 8 == 6): what the element calls per aggregate; a partial row range is
    refused (an f32 block sum is one ordered chain over all of its rows)."""
    w, h = 854, 480
    a = frames.random_frame(0xB10D, w, h, 4)
    b = a.copy()
    b[100:300, 400:2000] ^= 0x55
    da, db = gpu.DeviceBuffer(a.nbytes).upload(a), gpu.DeviceBuffer(b.nbytes).upload(b)
    fa, fb = gpu.make_frame(da.ptr, w, h, w * 4, "RGBA"), gpu.make_frame(db.ptr, w, h, w * 4, "RGBA")
    d = ctypes.c_double(-1)
    gpu.check(gpu.lib().mvfx_videocompare_distance(ctypes.byref(fa), ctypes.byref(fa), ctypes.byref(d), None))
    assert d.value == 0.0
    gpu.check(gpu.lib().mvfx_videocompare_distance(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(d), None))
    ha, hb = orc.blockhash(a, w, h, w * 4, "RGBA")[1], orc.blockhash(b, w, h, w * 4, "RGBA")[1]
    assert d.value == float(orc.hamming(ha, hb)) and d.value > 0
    sums = gpu.DeviceBuffer(64 * 4)
    assert gpu.lib().mvfx_blockhash_sums(ctypes.byref(fa), 0, 240, ctypes.c_void_p(sums.ptr), None) == gpu.ERR_INVALID_ARGUMENT
    assert "row bands" in gpu.last_error()


@pytest.mark.parametrize("size,n_pads", [((320, 240), 2), ((1920, 1080), 5), ((64, 8), 3), ((7680, 4320), 2), ((1024, 1024), 16)])
def test_videocompare_device_bits_and_distances_match_oracle(gpu, size, n_pads):
    """mvfx_videocompare_sharded_distances without a communicator (one GPU, whole frames and bands): block sums -> hash bits (4 bands
    of 16, upper median, equal-and-bright rule) and Hamming distances derived ON THE DEVICE equal the oracle's hashes / distances;
    ties (flat frames: every block sum equal) included."""
    w, h = size
    base = frames.random_frame(0x5EED0800 + w, w, h)
    pads = [base]
    for k in range(1, n_pads):
        f = base.copy()
        if k % 4 == 1:
            f[:: k + 1, :: 16 * k] ^= 0x3C
        elif k % 4 == 2:
            f[:] = 255 - f
            f[:, 3::4] = base[:, 3::4]
        elif k % 4 == 3:
            f[:] = 200                     # flat: all 64 sums equal -> the `v == median && median > half` branch decides every bit
        pads.append(f)
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in pads]
    want = [float(orc.hamming(hs[0], x)) for x in hs[1:]]
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in pads]
    arr = (gpu.Frame * n_pads)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
    got, hashes = gpu.videocompare_sharded_distances(None, arr, h, 0, None, want_hashes=True)
    assert hashes == hs and got == want
    if h % 16 == 0:   # a band alone: the totals are the band's partial sums -> the oracle's bits of those partial sums
        rows = h // 2
        band = (gpu.Frame * n_pads)(*[gpu.make_frame(b.ptr + rows * w * 4, w, rows, w * 4, "RGBA") for b in bufs])
        got_b, hashes_b = gpu.videocompare_sharded_distances(None, band, h, rows, None, want_hashes=True)
        sums = [orc.blockhash_sums(np.ascontiguousarray(f[rows:]), w, rows, w * 4, "RGBA") for f in pads]
        # the lower half of an 8 x 8 grid of (h / 8)-row blocks = block rows 4..7; orc sums of the half frame use (rows / 8)-row blocks,
        # so compare through the device's own band sums instead: they must reproduce the hashes
        dsum = gpu.DeviceBuffer(n_pads * 64 * 4)
        gpu.check(gpu.lib().mvfx_blockhash_sums_pads(band, n_pads, h, rows, ctypes.c_void_p(dsum.ptr), None))
        gpu.check(gpu.lib().mvfx_stream_synchronize(None))
        ds = dsum.download(dtype=np.uint32).reshape(n_pads, 64)
        assert hashes_b == [orc.blockhash_bits(ds[p], w, h) for p in range(n_pads)]
        assert got_b == [float(orc.hamming(hashes_b[0], x)) for x in hashes_b[1:]]


def test_videocompare_sharded_entry_rejects_what_cannot_shard(gpu):
    b = gpu.DeviceBuffer(100 * 100 * 4)
    arr = (gpu.Frame * 2)(gpu.make_frame(b.ptr, 100, 100, 400, "RGBA"), gpu.make_frame(b.ptr, 100, 100, 400, "RGBA"))
    out = (ctypes.c_double * 1)()
    L = gpu.lib()
    assert L.mvfx_videocompare_sharded_distances(None, arr, 2, 100, 0, out, None, None) == gpu.ERR_INVALID_ARGUMENT   # 100 % 8 != 0
    one = (gpu.Frame * 1)(gpu.make_frame(b.ptr, 96, 96, 400, "RGBA"))
    assert L.mvfx_videocompare_sharded_distances(None, one, 1, 96, 0, out, None, None) == gpu.ERR_INVALID_ARGUMENT    # no other pad
