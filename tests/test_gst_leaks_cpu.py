"""GStreamer's `leaks` tracer over the element paths that need no device (see tests/test_gst_leaks_gpu.py for the rest)."""
import numpy as np
import pytest

from tests import gst_env
from tests.test_gst_leaks_gpu import alive_objects

pytestmark = pytest.mark.skipif(not gst_env.available(), reason="GStreamer tools or our gst plugins not present")


def test_roundedcorners_system_memory(tmp_path):
    assert alive_objects(tmp_path, "videotestsrc num-buffers=20 ! video/x-raw,format=I420,width=320,height=240 ! roundedcorners "
                                   "border-radius-px=20 ! video/x-raw,format=A420 ! fakesink") == []
    assert alive_objects(tmp_path, "videotestsrc num-buffers=5 ! video/x-raw,format=I420,width=320,height=240 ! roundedcorners ! fakesink") == []


def test_imagersoverlay_attaching_the_composition_meta(tmp_path):
    from PIL import Image
    logo = tmp_path / "logo.png"
    rgba = np.zeros((16, 24, 4), np.uint8)
    rgba[..., 0] = 200
    rgba[..., 3] = 128
    Image.fromarray(rgba, "RGBA").save(logo)
    assert alive_objects(tmp_path, f"videotestsrc num-buffers=8 ! video/x-raw,format=RGBA,width=64,height=48 ! imagersoverlay location={logo} ! "
                                   "video/x-raw(memory:SystemMemory,meta:GstVideoOverlayComposition) ! fakesink") == []
