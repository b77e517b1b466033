"""roundedcorners as a gst-launch element on the CPU box: its system-memory path does what the
reference does -- libcairo renders the mask once, prepare_output_buffer appends the shared alpha
GstMemory (border/imp.rs:482-559), zero per-frame pixel work -- so it needs no device."""
import os

import numpy as np
import pytest

from tests import gst_env

pytestmark = pytest.mark.skipif(not gst_env.available(), reason="GStreamer tools or our gst plugins not present")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roundedcorners_masks.npz")


def _capture(tmp, pipeline, name):
    out = os.path.join(str(tmp), name)
    r = gst_env.run([gst_env.tool("gst-launch-1.0"), "-q"] + pipeline.split() + ["!", "filesink", f"location={out}"], tmp)
    assert r.returncode == 0, r.stdout
    assert "CRITICAL" not in r.stdout, r.stdout
    return np.fromfile(out, dtype=np.uint8)


@pytest.mark.parametrize("rad", [10, 30])
def test_i420_to_a420_pool_buffers_locked_meta(tmp_path, rad):
    """videotestsrc hands out pool buffers whose GstVideoMeta is LOCKED: the element must take the
    copy_region branch of add_video_meta (border/imp.rs:202-224), three frames in a row."""
    w, h, n = 64, 48, 3
    src = f"videotestsrc num-buffers={n} ! video/x-raw,format=I420,width={w},height={h}"
    raw = _capture(tmp_path, src, "in.raw").reshape(n, -1)
    got = _capture(tmp_path, src + f" ! roundedcorners border-radius-px={rad} ! video/x-raw,format=A420", "out.raw").reshape(n, -1)
    i420 = w * h * 3 // 2
    gold = np.load(GOLDEN)[f"w{w}_h{h}_r{rad}"]
    for k in range(n):
        assert np.array_equal(got[k, :i420], raw[k])
        assert np.array_equal(got[k, i420:].reshape(h, w), gold[:h, :w])


def test_radius_zero_is_i420_passthrough(tmp_path):
    src = "videotestsrc num-buffers=2 ! video/x-raw,format=I420,width=64,height=48"
    raw = _capture(tmp_path, src, "in.raw")
    same = _capture(tmp_path, src + " ! roundedcorners ! video/x-raw,format=I420", "pt.raw")
    assert np.array_equal(same, raw)
