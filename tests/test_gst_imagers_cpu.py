"""imagersoverlay on the CPU box: the parts of the element that involve no pixel arithmetic -- passthrough
without a location (overlay/imp.rs:527-553), the load errors (imp.rs:193-240), and the branch of transform_frame_ip that
only ATTACHES the GstVideoOverlayCompositionMeta when downstream accepts it (imp.rs:709-716) -- need no device."""
import os

import numpy as np
import pytest

from tests import gst_env

pytestmark = pytest.mark.skipif(not gst_env.available(), reason="GStreamer tools or our gst plugins not present")
LAUNCH = gst_env.tool("gst-launch-1.0")
SRC = "videotestsrc num-buffers=3 ! video/x-raw,format=RGBA,width=64,height=48"


def _logo(path):
    from PIL import Image
    rgba = np.zeros((16, 24, 4), np.uint8)
    rgba[..., 0] = 200
    rgba[..., 3] = 128
    Image.fromarray(rgba, "RGBA").save(path)


def test_no_location_is_passthrough(tmp_path):
    out = os.path.join(str(tmp_path), "o.raw")
    r = gst_env.run([LAUNCH, "-q"] + (SRC + f" ! tee name=t t. ! queue ! filesink location={tmp_path}/a.raw t. ! queue ! imagersoverlay ! filesink location={out}").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert np.array_equal(np.fromfile(out, np.uint8), np.fromfile(f"{tmp_path}/a.raw", np.uint8))


def test_load_errors(tmp_path):
    r = gst_env.run([LAUNCH] + (SRC + f" ! imagersoverlay location={tmp_path}/missing.png ! fakesink").split(), tmp_path)
    assert r.returncode != 0 and "Could not load overlay image" in r.stdout
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"this is not a png")
    r = gst_env.run([LAUNCH] + (SRC + f" ! imagersoverlay location={bad} ! fakesink").split(), tmp_path)
    assert r.returncode != 0 and "Could not decode overlay image container" in r.stdout
    logo = tmp_path / "logo.png"
    _logo(str(logo))
    r = gst_env.run([LAUNCH] + (SRC + f" ! imagersoverlay location={logo} max-alloc-bytes=100 ! fakesink").split(), tmp_path)
    assert r.returncode != 0 and "Could not decode overlay image container" in r.stdout


def test_attaches_the_composition_meta_when_downstream_takes_it(tmp_path):
    """caps with meta:GstVideoOverlayComposition downstream -> allow_attaching (imp.rs:690-699): frames pass untouched"""
    logo = tmp_path / "logo.png"
    _logo(str(logo))
    out = os.path.join(str(tmp_path), "o.raw")
    r = gst_env.run([LAUNCH, "-q"] + (SRC + f" ! tee name=t t. ! queue ! filesink location={tmp_path}/a.raw t. ! queue ! imagersoverlay location={logo} "
                                      f"offset-x=5 ! video/x-raw(memory:SystemMemory,meta:GstVideoOverlayComposition) ! filesink location={out}").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert np.array_equal(np.fromfile(out, np.uint8), np.fromfile(f"{tmp_path}/a.raw", np.uint8))
