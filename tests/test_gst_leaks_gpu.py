"""GStreamer's own `leaks` tracer (GST_TRACERS=leaks) over the element layer: at process exit no GstBuffer / GstMemory / GstCaps /
GstBufferPool / element may still be alive.  Refcount mistakes in allocators, pools, metas and request pads show up here, not
in pixel comparisons.  (the reference relies on Rust ownership for this; the C++ element layer needs the check.)"""
import re

import pytest

from tests import cubes, gst_env

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gst_env.available(), reason="GStreamer tools or our gst plugins not present")]
LAUNCH = gst_env.tool("gst-launch-1.0")
ALIVE = re.compile(r"object-alive, type-name=\(string\)(\w+), address=\(gpointer\)0x")


def alive_objects(tmp_path, pipeline):
    r = gst_env.run([LAUNCH, "-q"] + pipeline.split(), tmp_path, timeout=300,
                    extra_env={"GST_TRACERS": "leaks", "GST_DEBUG": "GST_TRACER:7", "GST_DEBUG_NO_COLOR": "1"})
    assert r.returncode == 0, r.stdout[-3000:]
    return ALIVE.findall(r.stdout)


def test_the_tracer_reports_a_deliberate_leak(tmp_path):
    """self-check of the method: fakesink's `last-sample` keeps nothing alive at exit, but a pipeline killed by num-buffers on a
    tee'd branch without a sink errors out -- instead use the tracer's own log to see that it is active"""
    r = gst_env.run([LAUNCH, "-q", "videotestsrc", "num-buffers=2", "!", "fakesink"], tmp_path,
                    extra_env={"GST_TRACERS": "leaks", "GST_DEBUG": "GST_TRACER:7", "GST_DEBUG_NO_COLOR": "1"})
    assert r.returncode == 0 and "object-alive" in r.stdout      # the record format is declared when the tracer is active


SRC = "videotestsrc num-buffers=12 ! video/x-raw,format={fmt},width=640,height=360"


@pytest.mark.parametrize("name", ["host_chain", "hip_chain_queues", "device_source", "fused_i420", "colordetect", "videocompare", "overlay_hip",
                                  "roundedcorners_hip"])
def test_no_object_outlives_the_pipeline(tmp_path, name):
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    det = "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
    pipes = {
        "host_chain": SRC.format(fmt="RGBx") + f" ! hsvfilter hue-shift=45 ! {det} ! video/x-raw,format=RGBA ! colorlut location={cube} ! fakesink",
        "hip_chain_queues": SRC.format(fmt="RGBx") + f" ! hipupload ! queue ! hsvfilter hue-shift=45 ! queue ! {det} ! "
                            f"video/x-raw(memory:HIPMemory),format=RGBA ! queue ! colorlut location={cube} ! hipdownload ! fakesink",
        "device_source": "hiptestsrc num-buffers=12 ! video/x-raw(memory:HIPMemory),format=RGBx,width=640,height=360 ! hsvfilter hue-shift=45 ! "
                         f"{det} ! video/x-raw(memory:HIPMemory),format=RGBA ! colorlut location={cube} ! hipdownload ! fakesink",
        "fused_i420": SRC.format(fmt="I420") + f" ! hipupload ! colorlut location={cube} ! hsvfilter hue-shift=30 ! hipdownload ! fakesink",
        "colordetect": SRC.format(fmt="RGBA") + " ! colordetect quality=5 max-colors=4 ! fakesink",
        "videocompare": ("videocompare name=c ! fakesink videotestsrc pattern=red num-buffers=6 ! video/x-raw,format=RGBA,width=320,height=240 ! c.sink_0 "
                         "videotestsrc pattern=red num-buffers=6 ! video/x-raw,format=RGBA,width=320,height=240 ! c.sink_1"),
        "overlay_hip": None,
        "roundedcorners_hip": SRC.format(fmt="I420") + " ! hipupload ! roundedcorners border-radius-px=40 ! hipdownload ! fakesink",
    }
    if name == "overlay_hip":
        from PIL import Image
        import numpy as np
        logo = tmp_path / "logo.png"
        rgba = np.zeros((16, 24, 4), np.uint8)
        rgba[..., 0] = 200
        rgba[..., 3] = 128
        Image.fromarray(rgba, "RGBA").save(logo)
        pipes[name] = SRC.format(fmt="RGBA") + f" ! hipupload ! imagersoverlay location={logo} offset-x=10 offset-y=10 ! hipdownload ! fakesink"
    assert alive_objects(tmp_path, pipes[name]) == []
