"""tests/frames.videotestsrc_smpte (the frames bench.py filters: BASELINE.json's "synthetic videotestsrc buffers") against the
real element: `videotestsrc ! video/x-raw,format=RGBA` of the GStreamer in this image, byte for byte, several sizes, several
consecutive frames (the snow generator's state runs on from frame to frame)."""
import os
import tempfile

import numpy as np
import pytest

from tests import frames, gst_env

pytestmark = pytest.mark.skipif(not gst_env.available(), reason="no GStreamer in this environment")


@pytest.mark.parametrize("width,height,n", [(336, 240, 3), (641, 481, 2), (640, 480, 2), (1920, 1080, 2), (3840, 2160, 2)])
def test_generator_is_byte_identical_to_videotestsrc(width, height, n):
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "f.raw")
    r = gst_env.run([gst_env.tool("gst-launch-1.0"), "-q", "videotestsrc", f"num-buffers={n}", "!",
                     f"video/x-raw,format=RGBA,width={width},height={height}", "!", "filesink", f"location={out}"], tmp, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    real = np.fromfile(out, dtype=np.uint8)
    stride = (width * 4 + 3) // 4 * 4
    assert real.size == n * stride * height
    mine, _ = frames.videotestsrc_smpte(width, height, n)
    assert np.array_equal(real.reshape(n, height, stride), mine)


def test_lcg_affine_maps_match_the_scalar_recurrence():
    a, c = frames.vts_lcg_affine(1000)
    s, seq = 12345678, []
    for _ in range(1000):
        s = (s * frames.VTS_LCG_A + frames.VTS_LCG_C) & 0xFFFFFFFF
        seq.append(s)
    got = (a * np.uint64(12345678) + c) & np.uint64(0xFFFFFFFF)
    assert got.tolist() == seq


def test_state_continues_across_calls():
    two, _ = frames.videotestsrc_smpte(64, 48, 2)
    one, st = frames.videotestsrc_smpte(64, 48, 1)
    nxt, _ = frames.videotestsrc_smpte(64, 48, 1, state=st)
    assert np.array_equal(two[0], one[0]) and np.array_equal(two[1], nxt[0])


def test_bench_pool_filler_equals_the_generator():
    """bench.py fills its resident pool on the device with torch (affine jumps of the LCG); same bytes as the numpy generator."""
    import torch

    import bench
    W, H, n, first = 336, 240, 5, 3
    flat = torch.empty((n, W * H * 4), dtype=torch.uint8)
    bench.fill_frames(torch, torch.device("cpu"), None, flat, "videotestsrc", W, H, first_frame=first)
    want, _ = frames.videotestsrc_smpte(W, H, first + n)
    assert np.array_equal(flat.numpy().reshape(n, H, W * 4), want[first:])


@pytest.mark.parametrize("fmt,width,height", [("RGBA", 320, 240), ("RGBx", 641, 481), ("BGRx", 64, 48), ("xRGB", 96, 64), ("RGB", 100, 50),
                                              ("BGR", 66, 34), ("ARGB", 1920, 1080)])
def test_hiptestsrc_paints_the_first_videotestsrc_frame(fmt, width, height):
    """hiptestsrc (the generator-free source of tools/bench_gst_pipeline.py) on system memory: byte-identical to the first frame
    of `videotestsrc pattern=smpte` in every packed RGB format."""
    tmp = tempfile.mkdtemp()
    outs = []
    for src in ("videotestsrc", "hiptestsrc"):
        out = os.path.join(tmp, src + ".raw")
        r = gst_env.run([gst_env.tool("gst-launch-1.0"), "-q", src, "num-buffers=1", "!",
                         f"video/x-raw,format={fmt},width={width},height={height}", "!", "filesink", f"location={out}"], tmp, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:]
        outs.append(np.fromfile(out, dtype=np.uint8))
    assert outs[0].size == outs[1].size
    bpp = 3 if fmt in ("RGB", "BGR") else 4
    stride = (width * bpp + 3) // 4 * 4            # row padding (3-byte formats) is not picture content
    rows = [o.reshape(height, stride)[:, :width * bpp] for o in outs]
    assert np.array_equal(rows[0], rows[1])
