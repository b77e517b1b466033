"""GPU tests that drive the real GStreamer elements with gst-launch-1.0, written after the
reference's own pipeline tests (video/videofx/tests/colordetect.rs) and BASELINE config 1/2.
Skipped when the image has no GStreamer tools (the kernels + C ABI are tested without them)."""
import os
import re

import numpy as np
import pytest

from tests import cubes, gst_env
from tests import oracle_binding as orc

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not gst_env.available(), reason="GStreamer tools or our gst plugins not present")]

LAUNCH = gst_env.tool("gst-launch-1.0")


def _capture(tmp, pipeline, out_name="out.raw"):
    out = os.path.join(str(tmp), out_name)
    r = gst_env.run([LAUNCH, "-q"] + pipeline.split() + ["!", "filesink", f"location={out}"], tmp)
    assert r.returncode == 0, r.stdout
    return np.fromfile(out, dtype=np.uint8)


def test_config1_hsvfilter_hue_shift_640x480_rgba(gpu, tmp_path):
    """BASELINE config 1: videotestsrc 640x480 RGBA ! hsvfilter hue-shift=90"""
    src = "videotestsrc num-buffers=3 ! video/x-raw,format=RGBA,width=640,height=480"
    raw = _capture(tmp_path, src, "in.raw")
    got = _capture(tmp_path, src + " ! hsvfilter hue-shift=90")
    assert raw.size == got.size == 3 * 640 * 480 * 4
    exp = raw.copy().reshape(3 * 480, 640 * 4)
    assert orc.hsvfilter(exp, 640, 640 * 4, "RGBA", (90.0, 1.0, 0.0, 1.0, 0.0)) == 0
    assert np.array_equal(got.reshape(exp.shape), exp)


def test_config2_hsvfilter_then_hsvdetector_negotiates_rgbx(gpu, tmp_path):
    """hsvfilter ! hsvdetector: RGBx between them, RGBA out"""
    src = "videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=RGBx,width=320,height=240"
    raw = _capture(tmp_path, src, "in.raw")
    got = _capture(tmp_path, src + " ! hsvfilter hue-shift=45 saturation-mul=1.25 value-off=0.02 ! hsvdetector hue-ref=120 "
                   "hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! video/x-raw,format=RGBA")
    mid = raw.copy().reshape(2 * 240, 320 * 4)
    orc.hsvfilter(mid, 320, 320 * 4, "RGBx", (45.0, 1.25, 0.0, 1.0, 0.02))
    exp = np.empty_like(mid)
    orc.hsvdetector(mid, 320 * 4, "RGBx", exp, 320 * 4, "RGBA", 320, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
    assert np.array_equal(got.reshape(exp.shape), exp)
    assert 0 < np.count_nonzero(exp[:, 3::4]) < exp[:, 3::4].size


def test_colordetect_red_posts_exactly_one_message(gpu, tmp_path):
    """video/videofx/tests/colordetect.rs:21-68"""
    r = gst_env.run([LAUNCH, "-m"] + "videotestsrc pattern=red num-buffers=2 ! video/x-raw,format=RGBA,width=320,height=240 "
                    "! colordetect ! fakesink".split(), tmp_path)
    assert r.returncode == 0, r.stdout
    msgs = re.findall(r"colordetect, dominant-color=\(string\)(\w+), palette=\(uint\)\{([^}]*)\}", r.stdout)
    assert len(msgs) == 1, r.stdout
    assert msgs[0][0] == "red"
    palette = [int(x) for x in msgs[0][1].split(",")]
    assert palette[0] == 0xFC0404 and len(palette) == 2


def test_colorlut_pipeline(gpu, tmp_path):
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    src = "videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=RGBA,width=320,height=240"
    raw = _capture(tmp_path, src, "in.raw")
    got = _capture(tmp_path, src + f" ! colorlut location={cube}")
    o = orc.CubeLut(cube.read_text())
    exp = np.empty_like(raw).reshape(2 * 240, 320 * 4)
    assert o.apply(raw.reshape(exp.shape), 320 * 4, exp, 320 * 4, 320, 2 * 240, "RGBA") == 0
    assert np.array_equal(got.reshape(exp.shape), exp)


def test_colorlut_without_location_fails_like_the_reference(gpu, tmp_path):
    """colorlut/imp.rs:175-180: start() -> ResourceError::Settings"""
    r = gst_env.run([LAUNCH] + "videotestsrc num-buffers=1 ! video/x-raw,format=RGBA ! colorlut ! fakesink".split(), tmp_path)
    assert r.returncode != 0
    assert "LUT file location is not configured" in r.stdout
    bad = tmp_path / "bad.cube"
    bad.write_text("LUT_1D_SIZE 2\nLUT_3D_SIZE 2\n0 0 0\n1 1 1\n")
    r = gst_env.run([LAUNCH] + f"videotestsrc num-buffers=1 ! video/x-raw,format=RGBA ! colorlut location={bad} ! fakesink".split(), tmp_path)
    assert r.returncode != 0 and "Failed to parse LUT file" in r.stdout


def test_roundedcorners_i420_to_a420(gpu, tmp_path):
    """I420 in, A420 out: YUV planes untouched, alpha plane appended (border/imp.rs:482-559)"""
    w, h, rad = 64, 48, 10
    src = f"videotestsrc num-buffers=1 ! video/x-raw,format=I420,width={w},height={h}"
    raw = _capture(tmp_path, src, "in.raw")
    got = _capture(tmp_path, src + f" ! roundedcorners border-radius-px={rad} ! video/x-raw,format=A420")
    i420 = w * h * 3 // 2
    assert raw.size == i420 and got.size == i420 + w * h
    assert np.array_equal(got[:i420], raw)
    alpha = got[i420:].reshape(h, w)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roundedcorners_masks.npz"))["w64_h48_r10"]
    assert np.array_equal(alpha, gold[:h, :w])  # the reference's own rasteriser (libcairo): no tolerance
    # 2 * radius > min(width, height) is a legal property value for the reference (cairo draws it)
    big = _capture(tmp_path, src + " ! roundedcorners border-radius-px=30 ! video/x-raw,format=A420", "big.raw")
    gold30 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roundedcorners_masks.npz"))["w64_h48_r30"]
    assert np.array_equal(big[i420:].reshape(h, w), gold30[:h, :w])
    # radius 0 negotiates I420 passthrough (border/imp.rs:405-409, 460-465)
    same = _capture(tmp_path, src + " ! roundedcorners ! video/x-raw,format=I420", "pt.raw")
    assert np.array_equal(same, raw)


def _videocompare(tmp_path, pattern_a, pattern_b, extra="", size=(320, 240)):
    """tests/videocompare.rs setup_pipeline: two videotestsrc -> videocompare -> fakesink"""
    caps = f"video/x-raw,format=RGBA,width={size[0]},height={size[1]}"
    pipeline = (f"videocompare name=compare {extra} ! fakesink "
                f"videotestsrc pattern={pattern_a} num-buffers=2 ! {caps} ! compare.sink_0 "
                f"videotestsrc pattern={pattern_b} num-buffers=2 ! {caps} ! compare.sink_1")
    r = gst_env.run([LAUNCH, "-m"] + pipeline.split(), tmp_path)
    assert r.returncode == 0, r.stdout
    return r.stdout


def test_videocompare_red_vs_red_detects(gpu, tmp_path):
    """video/videofx/tests/videocompare.rs:57-103: identical frames => message, distance 0 on sink_1"""
    out = _videocompare(tmp_path, "red", "red")
    msgs = re.findall(r"videocompare, running-time=\(guint64\)(\d+), pad-distances=\(structure\)<([^>]*)>", out)
    assert len(msgs) >= 1, out
    assert "sink_1" in msgs[0][1] and "distance" in msgs[0][1]
    assert re.search(r"distance\\=\\\(double\\\)0", msgs[0][1]), msgs[0][1]


def test_videocompare_snow_vs_red_is_silent(gpu, tmp_path):
    """tests/videocompare.rs:105-139: different frames at max-dist-threshold=0 => no message"""
    out = _videocompare(tmp_path, "red", "snow")
    assert "videocompare, running-time" not in out
    # a generous threshold reports the (non-zero) distance instead
    out = _videocompare(tmp_path, "red", "snow", "max-dist-threshold=64")
    m = re.search(r"distance\\=\\\(double\\\)(\d+)", out)
    assert m and int(m.group(1)) > 0


@pytest.mark.parametrize("size", [(854, 480), (1366, 768), (7, 5)])
def test_videocompare_sizes_that_are_not_multiples_of_8(gpu, tmp_path, size):
    """The reference accepts any RGB/RGBA size (videocompare/imp.rs:158-163) and image_hasher then takes its f32 path
    (854 % 8 == 6, 1366 % 8 == 6): the element must hash such streams, not error."""
    out = _videocompare(tmp_path, "red", "red", size=size)
    msgs = re.findall(r"videocompare, running-time=\(guint64\)(\d+), pad-distances=\(structure\)<([^>]*)>", out)
    assert len(msgs) >= 1, out
    assert re.search(r"distance\\=\\\(double\\\)0", msgs[0][1]), msgs[0][1]
    if size[0] > 8:
        out = _videocompare(tmp_path, "red", "snow", "max-dist-threshold=64", size=size)
        m = re.search(r"distance\\=\\\(double\\\)(\d+)", out)
        assert m and int(m.group(1)) > 0, out


@pytest.mark.parametrize("algo", ["mean", "gradient", "vertgradient", "doublegradient"])
def test_videocompare_resize_hashes(gpu, tmp_path, algo):
    """tests/videocompare.rs:57-139 with the other ImageHasher algorithms (hashed_image.rs:89-107): identical frames =>
    message with distance 0; smpte vs snow => silent at threshold 0, a positive distance with a generous threshold"""
    out = _videocompare(tmp_path, "red", "red", f"hash-algo={algo}")
    msgs = re.findall(r"videocompare, running-time=\(guint64\)(\d+), pad-distances=\(structure\)<([^>]*)>", out)
    assert len(msgs) >= 1, out
    assert re.search(r"distance\\=\\\(double\\\)0", msgs[0][1]), msgs[0][1]
    out = _videocompare(tmp_path, "smpte", "snow", f"hash-algo={algo}")
    assert "videocompare, running-time" not in out
    out = _videocompare(tmp_path, "smpte", "snow", f"hash-algo={algo} max-dist-threshold=64")
    m = re.search(r"distance\\=\\\(double\\\)(\d+)", out)
    assert m and int(m.group(1)) > 0, out


def test_videocompare_dssim_red_vs_red_and_snow(gpu, tmp_path):
    """tests/videocompare.rs:141-182 (feature dssim): identical frames => message with distance 0;
    snow vs red => silent at threshold 0, a positive non-integer distance with a generous threshold"""
    out = _videocompare(tmp_path, "red", "red", "hash-algo=dssim")
    msgs = re.findall(r"videocompare, running-time=\(guint64\)(\d+), pad-distances=\(structure\)<([^>]*)>", out)
    assert len(msgs) >= 1, out
    assert re.search(r"distance\\=\\\(double\\\)0[;\\]", msgs[0][1]), msgs[0][1]
    out = _videocompare(tmp_path, "red", "snow", "hash-algo=dssim")
    assert "videocompare, running-time" not in out
    out = _videocompare(tmp_path, "red", "snow", "hash-algo=dssim max-dist-threshold=1000")
    m = re.search(r"distance\\=\\\(double\\\)([0-9.e+-]+)", out)
    assert m and float(m.group(1)) > 0.01, out


# ---------------------------------------------------------------- memory:HIPMemory

def test_hipmemory_hsvfilter_matches_host_path(gpu, tmp_path):
    """videotestsrc ! hipupload ! hsvfilter ! hipdownload: same bytes as the oracle; frames stay in HBM between"""
    src = "videotestsrc num-buffers=3 ! video/x-raw,format=RGBA,width=640,height=480"
    raw = _capture(tmp_path, src, "in.raw")
    got = _capture(tmp_path, src + " ! hipupload ! hsvfilter hue-shift=90 saturation-mul=1.25 ! hipdownload")
    exp = raw.copy().reshape(3 * 480, 640 * 4)
    assert orc.hsvfilter(exp, 640, 640 * 4, "RGBA", (90.0, 1.25, 0.0, 1.0, 0.0)) == 0
    assert np.array_equal(got.reshape(exp.shape), exp)


def test_hipmemory_chain_stays_on_device(gpu, tmp_path):
    """hsvfilter ! hsvdetector ! colorlut on memory:HIPMemory caps end to end (one upload, one download)"""
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    src = "videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=RGBx,width=320,height=240"
    raw = _capture(tmp_path, src, "in.raw")
    r = gst_env.run([LAUNCH, "-v"] + (src + " ! hipupload ! hsvfilter hue-shift=45 ! hsvdetector hue-ref=120 hue-var=60 "
                    "saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA "
                    f"! colorlut location={cube} name=lut ! hipdownload ! filesink location={tmp_path}/out.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert "lut.GstPad:sink: caps = video/x-raw(memory:HIPMemory)" in r.stdout  # negotiated on device memory
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8)
    mid = raw.copy().reshape(2 * 240, 320 * 4)
    orc.hsvfilter(mid, 320, 320 * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    det = np.empty_like(mid)
    orc.hsvdetector(mid, 320 * 4, "RGBx", det, 320 * 4, "RGBA", 320, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
    o = orc.CubeLut(cube.read_text())
    exp = np.empty_like(det)
    assert o.apply(det, 320 * 4, exp, 320 * 4, 320, 2 * 240, "RGBA") == 0
    assert np.array_equal(got.reshape(exp.shape), exp)


def test_hipmemory_colordetect_and_cpu_consumer(gpu, tmp_path):
    """colordetect reads the device buffer; a CPU element downstream (filesink) can still map HIP memory"""
    r = gst_env.run([LAUNCH, "-m"] + ("videotestsrc pattern=red num-buffers=2 ! video/x-raw,format=RGBA,width=320,height=240 "
                    f"! hipupload ! colordetect ! hsvfilter hue-shift=120 ! filesink location={tmp_path}/o.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert len(re.findall(r"colordetect, dominant-color=\(string\)red", r.stdout)) == 1
    got = np.fromfile(f"{tmp_path}/o.raw", dtype=np.uint8).reshape(2 * 240, 320 * 4)
    exp = np.tile(np.array((255, 0, 0, 255), np.uint8), 2 * 240 * 320).reshape(2 * 240, 320 * 4)
    orc.hsvfilter(exp, 320, 320 * 4, "RGBA", (120.0, 1.0, 0.0, 1.0, 0.0))
    assert np.array_equal(got, exp)


def test_hipmemory_buffers_come_from_the_negotiated_pool(gpu, tmp_path):
    """ALLOCATION query (SURVEY 8f-1, d3d12colorlut/imp.rs:385-492): hipupload, hsvdetector and colorlut take their
    device output buffers from MvfxHipBufferPool instances negotiated with their peers, and the pools recycle:
    40 frames through three out-of-place elements need a handful of device buffers, not 120."""
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(9))
    n = 40
    pipeline = (f"videotestsrc num-buffers={n} ! video/x-raw,format=RGBx,width=320,height=240 ! hipupload ! hsvfilter hue-shift=30 "
                "! hsvdetector ! video/x-raw(memory:HIPMemory),format=RGBA "
                f"! colorlut location={cube} ! hipdownload ! fakesink")
    r = gst_env.run([LAUNCH] + pipeline.split(), tmp_path, extra_env={"GST_DEBUG": "mvfxhippool:6", "GST_DEBUG_NO_COLOR": "1"})
    assert r.returncode == 0, r.stdout
    configured = re.findall(r"mvfxhippool.*configured: video/x-raw\(memory:HIPMemory\)", r.stdout)
    allocated = re.findall(r"mvfxhippool.*allocated device buffer", r.stdout)
    assert len(configured) >= 3, r.stdout[-2000:]          # upload, detector and colorlut outputs
    assert 3 <= len(allocated) <= 24, (len(allocated), r.stdout[-2000:])  # recycled, far fewer than 3 * 40


@pytest.mark.parametrize("algo", ["blockhash", "dssim", "gradient"])
def test_videocompare_on_hipmemory_pads(gpu, tmp_path, algo):
    """Both pads fed with memory:HIPMemory buffers: the frames are hashed / compared where they are (no download);
    same messages as the system-memory pipelines of tests/videocompare.rs."""
    caps = "video/x-raw,format=RGBA,width=320,height=240"

    def run(pattern_b, extra=""):
        pipeline = (f"videocompare name=compare hash-algo={algo} {extra} ! fakesink "
                    f"videotestsrc pattern=red num-buffers=2 ! {caps} ! hipupload ! compare.sink_0 "
                    f"videotestsrc pattern={pattern_b} num-buffers=2 ! {caps} ! hipupload ! compare.sink_1")
        r = gst_env.run([LAUNCH, "-m"] + pipeline.split(), tmp_path, extra_env={"GST_DEBUG": "videocompare:7", "GST_DEBUG_NO_COLOR": "1"})
        assert r.returncode == 0, r.stdout
        return r.stdout

    out = run("red")
    assert "stays in device memory" in out, out[-1500:]
    msgs = re.findall(r"videocompare, running-time=\(guint64\)(\d+), pad-distances=\(structure\)<([^>]*)>", out)
    assert len(msgs) >= 1 and re.search(r"distance\\=\\\(double\\\)0", msgs[0][1]), out[-1500:]
    out = run("snow")
    assert "videocompare, running-time" not in out
    out = run("snow", "max-dist-threshold=1000")
    m = re.search(r"distance\\=\\\(double\\\)([0-9.e+-]+)", out)
    assert m and float(m.group(1)) > 0.01, out[-1500:]


@pytest.mark.parametrize("w,h,rad", [(64, 48, 10), (322, 242, 40)])
def test_roundedcorners_on_hipmemory(gpu, tmp_path, w, h, rad):
    """hipupload ! roundedcorners ! hipdownload: the A420 frame is composed in HBM (one launch) and is byte-identical
    to what the system-memory element produces (Y/U/V untouched + the same mask); radius 0 stays a passthrough."""
    src = f"videotestsrc num-buffers=2 ! video/x-raw,format=I420,width={w},height={h}"
    host = _capture(tmp_path, src + f" ! roundedcorners border-radius-px={rad} ! video/x-raw,format=A420", "host.raw")
    r = gst_env.run([LAUNCH, "-v"] + (src + f" ! hipupload ! roundedcorners name=rc border-radius-px={rad} ! "
                    f"video/x-raw(memory:HIPMemory),format=A420 ! hipdownload ! filesink location={tmp_path}/dev.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert re.search(r"rc\.GstPad:src: caps = video/x-raw\(memory:HIPMemory\)", r.stdout), r.stdout[-1500:]
    dev = np.fromfile(f"{tmp_path}/dev.raw", dtype=np.uint8)
    assert dev.size == host.size

    def planes(buf):  # default GstVideoInfo layout of A420; row padding is not part of the picture
        ru4 = lambda v: (v + 3) & ~3
        ys, cs, cw, ch = ru4(w), ru4((w + 1) // 2), (w + 1) // 2, (h + 1) // 2
        rh = (h + 1) & ~1
        sizes = [(ys, rh, w, h), (cs, rh // 2, cw, ch), (cs, rh // 2, cw, ch), (ys, rh, w, h)]
        frame = sum(st * rows for st, rows, _, _ in sizes)
        out = []
        for k in range(buf.size // frame):
            off = k * frame
            for st, rows, pw, ph in sizes:
                out.append(buf[off:off + st * rows].reshape(rows, st)[:ph, :pw].copy())
                off += st * rows
        return out

    for a, b in zip(planes(dev), planes(host)):
        assert np.array_equal(a, b)
    raw = _capture(tmp_path, src, "in.raw")
    same = _capture(tmp_path, src + " ! hipupload ! roundedcorners ! video/x-raw(memory:HIPMemory),format=I420 ! hipdownload", "pt.raw")
    assert same.size == raw.size
    if w % 4 == 0 and h % 2 == 0:  # no row padding: whole buffers comparable
        assert np.array_equal(same, raw)


@pytest.mark.parametrize("w,h", [(320, 240), (1280, 720)])
def test_colorlut_on_hipmemory_i420_equals_the_videoconvert_sandwich(gpu, tmp_path, w, h):
    """The reference's example pipeline `... ! videoconvert ! colorlut location=... ! videoconvert ! ...`
    (colorlut/imp.rs:17-19) with the image's REAL videoconvert elements around our colorlut, against
    `hipupload ! colorlut ! hipdownload` on I420 buffers in HBM (one fused kernel, no RGBA frame): the I420 output
    must be identical byte for byte (SD: BT.601 / chroma-site none; HD: BT.709 / co-sited)."""
    from tests import cubes
    cube = os.path.join(str(tmp_path), "look.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(17))
    src = f"videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=I420,width={w},height={h}"
    sandwich = _capture(tmp_path, src + f" ! videoconvert ! video/x-raw,format=RGBA ! colorlut location={cube} ! videoconvert ! video/x-raw,format=I420", "ref.raw")
    r = gst_env.run([LAUNCH, "-v"] + (src + f" ! hipupload ! colorlut name=lut location={cube} ! hipdownload ! filesink location={tmp_path}/dev.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert re.search(r"lut\.GstPad:src: caps = video/x-raw\(memory:HIPMemory\).*format=\(string\)I420", r.stdout), r.stdout[-1500:]
    fused = np.fromfile(f"{tmp_path}/dev.raw", dtype=np.uint8)
    assert fused.size == sandwich.size == 2 * w * h * 3 // 2
    assert np.array_equal(fused, sandwich), np.argwhere(fused != sandwich)[:8]
    raw = _capture(tmp_path, src, "in.raw")
    assert not np.array_equal(fused, raw)  # the LUT did something


@pytest.mark.parametrize("w,h", [(320, 240), (1280, 720)])
def test_hsvfilter_on_hipmemory_i420_equals_the_videoconvert_sandwich(gpu, tmp_path, w, h):
    """`videoconvert ! hsvfilter ! videoconvert` with the image's real videoconvert elements against
    `hipupload ! hsvfilter ! hipdownload` on I420 buffers in HBM (fused kernel): identical I420 bytes."""
    props = "hue-shift=90 saturation-mul=1.25 saturation-off=-0.05 value-mul=0.9 value-off=0.02"
    src = f"videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=I420,width={w},height={h}"
    sandwich = _capture(tmp_path, src + f" ! videoconvert ! video/x-raw,format=RGBA ! hsvfilter {props} ! videoconvert ! video/x-raw,format=I420", "ref.raw")
    r = gst_env.run([LAUNCH, "-v"] + (src + f" ! hipupload ! hsvfilter name=f {props} ! hipdownload ! filesink location={tmp_path}/dev.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert re.search(r"f\.GstPad:src: caps = video/x-raw\(memory:HIPMemory\).*format=\(string\)I420", r.stdout), r.stdout[-1500:]
    fused = np.fromfile(f"{tmp_path}/dev.raw", dtype=np.uint8)
    assert fused.size == sandwich.size == 2 * w * h * 3 // 2
    assert np.array_equal(fused, sandwich), np.argwhere(fused != sandwich)[:8]


def test_hsvdetector_on_hipmemory_i420_equals_videoconvert_then_detector(gpu, tmp_path):
    """`videoconvert ! hsvdetector` with the image's real videoconvert against `hipupload ! hsvdetector ! hipdownload` fed
    with I420 buffers in HBM (fused kernel): identical RGBA bytes."""
    w, h = 320, 240
    props = "hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
    src = f"videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=I420,width={w},height={h}"
    ref = _capture(tmp_path, src + f" ! videoconvert ! video/x-raw,format=RGBx ! hsvdetector {props} ! video/x-raw,format=RGBA", "ref.raw")
    r = gst_env.run([LAUNCH, "-v"] + (src + f" ! hipupload ! hsvdetector name=d {props} ! video/x-raw(memory:HIPMemory),format=RGBA ! hipdownload "
                                      f"! filesink location={tmp_path}/dev.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    assert re.search(r"d\.GstPad:sink: caps = video/x-raw\(memory:HIPMemory\).*format=\(string\)I420", r.stdout), r.stdout[-1500:]
    fused = np.fromfile(f"{tmp_path}/dev.raw", dtype=np.uint8)
    assert fused.size == ref.size == 2 * w * h * 4
    assert np.array_equal(fused, ref), np.argwhere(fused != ref)[:8]
    assert 0 < np.count_nonzero(fused[3::4]) < fused.size // 4


def test_hipmemory_chain_with_fences_across_streaming_threads(gpu, tmp_path):
    """Every element only records / waits for the fence of the device block (no host wait per buffer,
    d3d12colorlut/imp.rs:695-714).  `queue`s put the three filters on three streaming threads = three HIP streams, 1080p
    frames keep several kernels in flight, pools recycle blocks whose last reader may still be running: all 12 frames must
    still be bit-exact.  Source: hiptestsrc (pre-painted videotestsrc-smpte frames in the pinned pool buffers offered by hipupload)."""
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    w, h, n = 1920, 1080, 12
    src = f"hiptestsrc num-buffers={{n}} ! video/x-raw,format=RGBx,width={w},height={h},framerate=30/1"
    raw = _capture(tmp_path, src.format(n=1), "in.raw")
    assert raw.size == w * h * 4 and len(np.unique(raw[-4096:])) > 100   # the snow corner of the smpte frame
    r = gst_env.run([LAUNCH, "-q"] + (src.format(n=n) + " ! hipupload ! queue ! hsvfilter hue-shift=45 ! queue ! hsvdetector hue-ref=120 "
                    "hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA "
                    f"! queue ! colorlut location={cube} ! queue ! hipdownload ! filesink location={tmp_path}/out.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8).reshape(n, h, w * 4)
    mid = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    det = np.empty_like(mid)
    orc.hsvdetector(mid, w * 4, "RGBx", det, w * 4, "RGBA", w, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
    exp = np.empty_like(det)
    assert orc.CubeLut(cube.read_text()).apply(det, w * 4, exp, w * 4, w, h, "RGBA") == 0
    for k in range(n):
        assert np.array_equal(got[k], exp), f"frame {k} differs"


def test_hiptestsrc_device_memory_source(gpu, tmp_path):
    """hiptestsrc negotiates memory:HIPMemory directly (frames born in HBM): first pass of every pool buffer is the pattern"""
    w, h = 320, 240
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in.raw")
    got = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h} ! hsvfilter hue-shift=90 "
                   "! hipdownload", "out.raw")
    exp = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(exp, w, w * 4, "RGBA", (90.0, 1.0, 0.0, 1.0, 0.0))
    assert np.array_equal(got.reshape(exp.shape), exp)


def test_hiptestsrc_device_frames_are_refreshed_although_hsvfilter_works_in_place(gpu, tmp_path):
    """Frames born in HBM: the source refreshes every recycled block from its device master (async device-to-device copy ordered
    by the block's fence) while downstream elements on other streaming threads may still be reading the block's previous life;
    hsvfilter then overwrites it in place.  40 frames at 1080p through three streaming threads: every one equals the oracle's
    answer for THE pattern (a block handed out un-refreshed would come out filtered twice)."""
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    w, h, n = 1920, 1080, 40
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBx,width={w},height={h}", "in.raw")
    r = gst_env.run([LAUNCH, "-q"] + (f"hiptestsrc num-buffers={n} ! video/x-raw(memory:HIPMemory),format=RGBx,width={w},height={h},framerate=30/1 ! "
                    "hsvfilter hue-shift=45 ! queue max-size-buffers=2 ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 "
                    "value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA ! queue max-size-buffers=2 ! "
                    f"colorlut location={cube} ! hipdownload ! filesink location={tmp_path}/out.raw").split(), tmp_path)
    assert r.returncode == 0, r.stdout
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8).reshape(n, h, w * 4)
    mid = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    det = np.empty_like(mid)
    orc.hsvdetector(mid, w * 4, "RGBx", det, w * 4, "RGBA", w, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
    exp = np.empty_like(det)
    assert orc.CubeLut(cube.read_text()).apply(det, w * 4, exp, w * 4, w, h, "RGBA") == 0
    bad = [k for k in range(n) if not np.array_equal(got[k], exp)]
    assert bad == []


def test_hipmemory_tee_two_device_readers_on_recycled_blocks(gpu, tmp_path):
    """One device buffer read by TWO elements on two streaming threads (tee ! queue ! hsvdetector, twice): both readers release
    the block's fence; the release of the second chains onto the first's, so the source's next refresh + the in-place hsvfilter
    of the recycled block wait for BOTH kernels (a fence that only remembered the later reader let the block be overwritten
    under the other one).  40 frames at 1080p from a small pool; both branches must equal the oracle on every frame."""
    w, h, n = 1920, 1080, 40
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBx,width={w},height={h}", "in.raw")
    det = [(120.0, 60.0, 0.6, 0.4, 0.6, 0.4), (300.0, 80.0, 0.5, 0.5, 0.5, 0.5)]
    branch = ("t. ! queue max-size-buffers=3 ! hsvdetector hue-ref={0} hue-var={1} saturation-ref={2} saturation-var={3} value-ref={4} "
              "value-var={5} ! video/x-raw(memory:HIPMemory),format=RGBA ! hipdownload ! filesink location={6}")
    r = gst_env.run([LAUNCH, "-q"] + (f"hiptestsrc num-buffers={n} ! video/x-raw(memory:HIPMemory),format=RGBx,width={w},height={h},framerate=30/1 ! "
                    "hsvfilter hue-shift=45 ! tee name=t " + branch.format(*det[0], f"{tmp_path}/a.raw") + " " +
                    branch.format(*det[1], f"{tmp_path}/b.raw")).split(), tmp_path)
    assert r.returncode == 0, r.stdout
    mid = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    for name, settings in zip(("a.raw", "b.raw"), det):
        got = np.fromfile(f"{tmp_path}/{name}", dtype=np.uint8).reshape(n, h, w * 4)
        exp = np.empty_like(mid)
        orc.hsvdetector(mid, w * 4, "RGBx", exp, w * 4, "RGBA", w, settings)
        assert 0 < np.count_nonzero(exp[:, 3::4]) < exp[:, 3::4].size
        bad = [k for k in range(n) if not np.array_equal(got[k], exp)]
        assert bad == [], f"{name}: frames {bad} differ"


@pytest.mark.parametrize("mode", ["1", "2"], ids=["stream-ordered", "fenced"])
def test_sixteen_hsvfilter_branches_through_the_launch_combiner(gpu, tmp_path, mode):
    """16 `hiptestsrc ! hsvfilter ! hipdownload ! filesink` streams in one process with MVFX_COMBINE=1: every hsvfilter still makes
    one call per buffer (hsvfilter/imp.rs:322-326), the library coalesces the frames of the 16 streaming threads into batched
    launches with per-frame settings.  Every branch has its own hue-shift (both signs) / saturation-mul; every frame of every branch
    equals the oracle's answer for that branch; and launches were shared."""
    w, h, n, branches = 640, 360, 10, 16
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in.raw")
    settings = [((23 * k) % 360 - 150.0, 1.0 + 0.05 * k, 0.0, 1.0, 0.0) for k in range(branches)]
    caps = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=30/1"
    pipe = " ".join(f"hiptestsrc num-buffers={n} ! {caps} ! hsvfilter hue-shift={s[0]} saturation-mul={s[1]} ! hipdownload ! "
                    f"filesink location={tmp_path}/b{k}.raw" for k, s in enumerate(settings))
    r = gst_env.run([LAUNCH, "-q"] + pipe.split(), tmp_path, extra_env={"MVFX_COMBINE": mode, "MVFX_COMBINE_STATS": "1"})
    assert r.returncode == 0, r.stdout
    for k, s in enumerate(settings):
        got = np.fromfile(f"{tmp_path}/b{k}.raw", dtype=np.uint8).reshape(n, h, w * 4)
        exp = raw.copy().reshape(h, w * 4)
        assert orc.hsvfilter(exp, w, w * 4, "RGBA", s) == 0
        bad = [i for i in range(n) if not np.array_equal(got[i], exp)]
        assert bad == [], f"branch {k} (hue-shift {s[0]}): frames {bad} differ"
    m = re.search(r"mvfx combiner device 0: (\d+) launches for (\d+) frames", r.stdout)
    assert m and int(m.group(2)) == branches * n and int(m.group(1)) < int(m.group(2)), r.stdout[-500:]


# ---- imagersoverlay: PNG logo blended by the HIP kernel, positions per overlay/imp.rs:84-191
def _logo_png(path, w=48, h=32):
    from PIL import Image
    rng = np.random.default_rng(0x1060)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgba[..., 3] = np.linspace(0, 255, w, dtype=np.uint8)[None, :]   # alpha ramp incl. 0 and 255
    rgba[:4, :, 3] = 0
    Image.fromarray(rgba, "RGBA").save(path)
    return np.ascontiguousarray(rgba[..., [2, 1, 0, 3]]).reshape(h, w * 4)  # BGRA, what load_image hands to the composition


@pytest.mark.parametrize("fmt,bpp", [("RGBA", 4), ("BGRx", 4), ("RGB", 3)])
def test_imagersoverlay_blend_matches_oracle(gpu, tmp_path, fmt, bpp):
    logo = tmp_path / "logo.png"
    bgra = _logo_png(str(logo))
    w, h, n = 320, 240, 2
    src = f"videotestsrc num-buffers={n} pattern=smpte ! video/x-raw,format={fmt},width={w},height={h}"
    raw = _capture(tmp_path, src, "in.raw").reshape(n, h, -1)
    # offset-x >= 0: from the left; offset-y < 0: from the bottom edge (imp.rs:126-145); relative-x adds 10 % of the width
    got = _capture(tmp_path, src + f" ! imagersoverlay location={logo} offset-x=20 offset-y=-10 relative-x=0.1 alpha=0.7 ! video/x-raw,format={fmt}")
    got = got.reshape(n, h, -1)
    stride = raw.shape[2]
    x, y = 20 + int(0.1 * w), h - 10 - 32
    for k in range(n):
        want = raw[k].copy()
        assert orc.overlay_blend(want, w, h, stride, fmt, bgra, 48, 32, x, y, 0.7) == 0
        assert np.array_equal(got[k], want)
    assert np.count_nonzero(got[0] != raw[0]) > 1000
    # pixels-absolute positioning with coef-x/y, partly outside the frame
    got = _capture(tmp_path, src + f" ! imagersoverlay location={logo} positioning-mode=pixels-absolute offset-x=-30 offset-y=5 coef-y=0.5 ! "
                   f"video/x-raw,format={fmt}", "abs.raw").reshape(n, h, -1)
    want = raw[0].copy()
    assert orc.overlay_blend(want, w, h, stride, fmt, bgra, 48, 32, -30, 5 + int(0.5 * h), 1.0) == 0
    assert np.array_equal(got[0], want)


def test_imagersoverlay_on_hipmemory_and_scaled(gpu, tmp_path):
    logo = tmp_path / "logo.png"
    bgra = _logo_png(str(logo))
    w, h = 320, 240
    src = f"videotestsrc num-buffers=3 pattern=smpte ! video/x-raw,format=RGBA,width={w},height={h}"
    raw = _capture(tmp_path, src, "in.raw").reshape(3, h, w * 4)
    dev = _capture(tmp_path, src + f" ! hipupload ! imagersoverlay location={logo} offset-x=100 offset-y=50 alpha=0.5 ! "
                   "video/x-raw(memory:HIPMemory),format=RGBA ! hipdownload", "dev.raw").reshape(3, h, w * 4)
    for k in range(3):  # the smpte pattern's snow strip differs from frame to frame
        want = raw[k].copy()
        assert orc.overlay_blend(want, w, h, w * 4, "RGBA", bgra, 48, 32, 100, 50, 0.5) == 0
        assert np.array_equal(dev[k], want)
    # render size != image size: libgstvideo scales the rectangle (once per composition); host and device paths agree
    args = f"imagersoverlay location={logo} offset-x=10 offset-y=10 overlay-width=96 overlay-height=40"
    host = _capture(tmp_path, src + f" ! {args} ! video/x-raw,format=RGBA", "sh.raw").reshape(3, h, w * 4)
    devs = _capture(tmp_path, src + f" ! hipupload ! {args} ! video/x-raw(memory:HIPMemory),format=RGBA ! hipdownload", "sd.raw").reshape(3, h, w * 4)
    assert np.array_equal(host, devs)
    changed = np.argwhere((host[0] != raw[0]).reshape(h, w, 4).any(axis=2))
    assert changed[:, 0].min() >= 10 and changed[:, 0].max() < 50 and changed[:, 1].min() >= 10 and changed[:, 1].max() < 106


def test_imagersoverlay_without_location_is_passthrough_and_missing_file_errors(gpu, tmp_path):
    src = "videotestsrc num-buffers=2 ! video/x-raw,format=RGBA,width=64,height=48"
    raw = _capture(tmp_path, src, "in.raw")
    same = _capture(tmp_path, src + " ! imagersoverlay ! video/x-raw,format=RGBA", "pt.raw")
    assert np.array_equal(same, raw)
    r = gst_env.run([LAUNCH] + (src + f" ! imagersoverlay location={tmp_path}/nope.png ! fakesink").split(), tmp_path)
    assert r.returncode != 0 and "Could not load overlay image" in r.stdout


@pytest.mark.parametrize("queues", [True, False], ids=["queues", "one-thread"])
def test_device_chain_on_frames_that_all_differ(gpu, tmp_path, queues):
    """videotestsrc pattern=snow: every frame is different, so an ordering mistake between the elements' alternating HIP streams, the
    pool blocks kept in rotation and the upload / download copies shows as a wrong frame (constant test frames hide it).  Upload, three
    device filters (with and without queues between them), download; every output frame against the oracle applied to ITS input."""
    w, h, n = 320, 240, 240
    cube = os.path.join(str(tmp_path), "look.cube")
    open(cube, "w").write(cubes.analytic_3d(17))
    q = " ! queue max-size-buffers=3" if queues else ""
    fin, fout = os.path.join(str(tmp_path), "in.raw"), os.path.join(str(tmp_path), "out.raw")
    pipe = (f"videotestsrc pattern=snow num-buffers={n} ! video/x-raw,format=RGBx,width={w},height={h},framerate=30/1 ! tee name=t "
            f"t. ! queue ! filesink location={fin} "
            f"t. ! queue ! hipupload{q} ! hsvfilter hue-shift=45{q} ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 "
            f"value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA{q} ! colorlut location={cube}{q} ! hipdownload ! "
            f"filesink location={fout}")
    r = gst_env.run([LAUNCH, "-q"] + pipe.split(), tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    a = np.fromfile(fin, np.uint8).reshape(n, h, w * 4)
    b = np.fromfile(fout, np.uint8).reshape(n, h, w * 4)
    assert not np.array_equal(a[0], a[1]), "the source frames do not differ"
    lut = orc.CubeLut(open(cube).read())
    for k in range(n):
        mid = a[k].copy()
        assert orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0)) == 0
        det = np.empty_like(mid)
        assert orc.hsvdetector(mid, w * 4, "RGBx", det, w * 4, "RGBA", w, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4)) == 0
        exp = np.empty_like(det)
        assert lut.apply(det, w * 4, exp, w * 4, w, h, "RGBA") == 0
        assert np.array_equal(b[k], exp), f"frame {k}"


def _pair_stats(text):
    return _pair_stats_of("hsvfilter", text)


def _pair_stats_of(element, text):
    """(buffers, pair launches, lone launches of a held-back frame, plain launches while the element does not hold back)"""
    m = re.search(element + r" \S+: (\d+) device buffers = 2 x (\d+) pair launches \+ (\d+) single launches \+ (\d+) direct launches", text)
    assert m, text[-2000:]
    return tuple(int(x) for x in m.groups())


def test_hsvfilter_pair_launches_every_buffer_exactly_once(gpu, tmp_path):
    """Opt-in since round 5 (MVFX_ELEMENT_PAIR=1|2).  On device memory hsvfilter holds ONE buffer's kernel back and launches two consecutive frames together (one call per
    buffer stays the contract, hsvfilter/imp.rs:322-326).  (a) nobody looks at the blocks (fakesink): 21 buffers leave in pairs, the last
    one flushed at EOS; (b) a consumer on ANOTHER streaming thread (queue ! hipdownload) flushes held-back frames itself or finds them
    paired: all 41 frames come out filtered exactly once, byte for byte."""
    w, h = 640, 360
    caps = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=30/1"
    for mode in ("2", "1"):
        r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc num-buffers=21 refresh=false ! {caps} ! hsvfilter hue-shift=45 ! fakesink sync=false".split(),
                        tmp_path, extra_env={"MVFX_ELEMENT_PAIR_STATS": "1", "MVFX_ELEMENT_PAIR": mode})
        assert r.returncode == 0, r.stdout
        buffers, pairs, singles, direct = _pair_stats(r.stdout)
        assert buffers == 21 and 2 * pairs + singles + direct == 21
        if mode == "2":   # always hold back: ten pairs and the last frame at EOS
            assert (pairs, singles, direct) == (10, 1, 0)
        # (default mode: a block the source is still filling when its buffer arrives is launched the plain way, and the frame held back
        # before it leaves alone -- how many of the first round through the pool that hits depends on the box)
    n = 41
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in.raw")
    r = gst_env.run([LAUNCH, "-q"] + (f"hiptestsrc num-buffers={n} ! {caps} ! hsvfilter hue-shift=45 saturation-mul=1.2 ! queue max-size-buffers=3 ! "
                                      f"hipdownload ! filesink location={tmp_path}/out.raw").split(), tmp_path,
                    extra_env={"MVFX_ELEMENT_PAIR_STATS": "1", "MVFX_ELEMENT_PAIR": "1"})
    assert r.returncode == 0, r.stdout
    buffers, pairs, singles, direct = _pair_stats(r.stdout)
    assert buffers == n and 2 * pairs + singles + direct == n
    exp = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(exp, w, w * 4, "RGBA", (45.0, 1.2, 0.0, 1.0, 0.0))
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8).reshape(n, h, w * 4)
    assert [k for k in range(n) if not np.array_equal(got[k], exp)] == []


@pytest.mark.parametrize("env", [{}, {"MVFX_ELEMENT_PAIR": "0"}], ids=["default", "explicit-0"])
def test_hsvfilter_launches_once_per_buffer_by_default(gpu, tmp_path, env):
    """Round 5 (VERDICT r4 W-semantics): the DEFAULT is the reference's contract to the letter -- one launch per transform call, nothing
    held back, the call's flow return is that frame's (hsvfilter/imp.rs:322-326).  No statistics line = the hold never saw a buffer."""
    w, h = 320, 240
    caps = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=30/1"
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in.raw")
    r = gst_env.run([LAUNCH, "-q"] + (f"hiptestsrc num-buffers=5 ! {caps} ! hsvfilter hue-shift=90 ! hipdownload ! filesink location={tmp_path}/out.raw").split(),
                    tmp_path, extra_env=dict(env, MVFX_ELEMENT_PAIR_STATS="1"))
    assert r.returncode == 0 and "pair launches" not in r.stdout, r.stdout
    exp = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(exp, w, w * 4, "RGBA", (90.0, 1.0, 0.0, 1.0, 0.0))
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8).reshape(5, h, w * 4)
    assert all(np.array_equal(got[k], exp) for k in range(5))
    # all three device elements in a row, still nothing held back
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(9))
    r = gst_env.run([LAUNCH, "-q"] + (f"hiptestsrc num-buffers=9 ! {caps.replace('RGBA', 'RGBx')} ! hsvfilter ! hsvdetector ! colorlut location={cube} ! fakesink").split(),
                    tmp_path, extra_env=dict(env, MVFX_ELEMENT_PAIR_STATS="1"))
    assert r.returncode == 0 and "pair launches" not in r.stdout, r.stdout


PANIC_CAPS = "video/x-raw(memory:HIPMemory),format=RGB,width=641,height=1"  # stride 1924, 1924 % 3 != 0: hsvfilter/imp.rs:92 asserts


def test_a_failing_frame_is_the_error_of_its_own_call_by_default(gpu, tmp_path):
    """The reference's transform_frame_ip panics on this plane size (assert_eq!(data.len() % nb_channels, 0), hsvfilter/imp.rs:92) and the
    panic becomes the element's error + GST_FLOW_ERROR of THAT buffer.  Default mode: the first buffer's own call fails (already in
    preroll), nothing is held back."""
    r = gst_env.run([LAUNCH] + f"hiptestsrc num-buffers=4 ! {PANIC_CAPS},framerate=30/1 ! hsvfilter hue-shift=10 ! fakesink".split(), tmp_path,
                    extra_env={"MVFX_ELEMENT_PAIR_STATS": "1"})
    assert r.returncode != 0
    assert "asserts on this (hsvfilter/imp.rs:92)" in r.stdout and "held-back frame" not in r.stdout and "pair launches" not in r.stdout


def test_a_failing_held_back_frame_is_posted_by_the_idle_flush(gpu, tmp_path):
    """Pairs on (MVFX_ELEMENT_PAIR=2) and a live source at 5 frames/s with the idle interval forced to 20 ms: frame 0 is held back, no
    second buffer comes in time, the timer thread launches it alone -- and the launch fails: posted as "held-back frame" (gst-launch
    stops there; that the NEXT transform call returns GST_FLOW_ERROR is tests/test_gst_inprocess_gpu.py::
    test_a_failed_held_back_frame_fails_the_next_transform_call_with_flow_error, whose application keeps running)."""
    for element in ("hsvfilter hue-shift=10", "hsvdetector"):
        r = gst_env.run([LAUNCH] + f"hiptestsrc is-live=true num-buffers=6 ! {PANIC_CAPS},framerate=5/1 ! {element} ! fakesink".split(),
                        tmp_path, extra_env={"MVFX_ELEMENT_PAIR": "2", "MVFX_PAIR_IDLE_US": "20000", "MVFX_ELEMENT_PAIR_STATS": "1"})
        assert r.returncode != 0, r.stdout
        assert "held-back frame: mvfx status -8" in r.stdout and "idle flush of a frame held for" in r.stdout, r.stdout
        buffers, pairs, singles, direct = _pair_stats_of(element.split()[0], r.stdout)
        assert (pairs, singles, direct) == (0, 1, 0) and buffers <= 2


def test_a_stalled_sources_last_frame_is_processed_within_one_frame_interval(gpu, tmp_path):
    """Pairs on, a live source at 10 frames/s (interval 100 ms), nobody looks at the output (fakesink): every buffer is held back and no
    second one comes within the interval, so the timer thread launches each frame alone, after one frame interval and well before the
    next frame (which comes one interval after the PREVIOUS frame's arrival).  Without the timer the last frame of a stalled camera
    stayed unprocessed until EOS."""
    w, h, n = 320, 240, 5
    caps = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=10/1"
    r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc is-live=true num-buffers={n} ! {caps} ! hsvfilter hue-shift=45 ! fakesink".split(), tmp_path,
                    extra_env={"MVFX_ELEMENT_PAIR": "2", "MVFX_PAIR_IDLE_US": "50000", "MVFX_ELEMENT_PAIR_STATS": "1"})
    assert r.returncode == 0, r.stdout
    held = [int(x) for x in re.findall(r"idle flush of a frame held for (\d+) us \(interval 50000 us\)", r.stdout)]
    assert len(held) >= n - 1, r.stdout  # (the last frame may be taken by EOS instead)
    assert all(50000 <= t < 90000 for t in held), held
    buffers, pairs, singles, direct = _pair_stats(r.stdout)
    assert buffers == n and pairs == 0 and singles == n and direct == 0
    # and the frames are right: the same live pipeline into a file (hipdownload's look or the timer, whichever is first)
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in.raw")
    r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc is-live=true num-buffers={n} ! {caps} ! hsvfilter hue-shift=45 ! queue ! hipdownload ! filesink location={tmp_path}/o.raw".split(),
                    tmp_path, extra_env={"MVFX_ELEMENT_PAIR": "2", "MVFX_PAIR_IDLE_US": "2000"})
    assert r.returncode == 0, r.stdout
    exp = raw.copy().reshape(h, w * 4)
    orc.hsvfilter(exp, w, w * 4, "RGBA", (45.0, 1.0, 0.0, 1.0, 0.0))
    got = np.fromfile(f"{tmp_path}/o.raw", dtype=np.uint8).reshape(n, h, w * 4)
    assert all(np.array_equal(got[k], exp) for k in range(n))


def test_tee_with_two_holding_readers_of_one_input_block(gpu, tmp_path):
    """advisor r4 (medium): two HOLDING readers of one input block -- tee ! queue ! hsvdetector, twice, always holding back
    (MVFX_ELEMENT_PAIR=2) -- both mark the block.  The second mark used to overwrite the first (check-then-set race in
    mvfx_hip_memory_set_deferred): the source's refill of the recycled block then flushed only one of the two held-back kernels and the
    other read overwritten pixels.  A foreign mark is now run, never overwritten.  refresh=true rewrites every recycled block and the
    in-place hsvfilter in front filters it again, so a reader that comes late sees unfiltered or half-filtered pixels -- every frame of
    both branches must equal the oracle."""
    w, h, n = 640, 360, 200
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBx,width={w},height={h}", "in.raw").reshape(h, w * 4)
    mid = raw.copy()
    orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    det = [(120.0, 60.0, 0.6, 0.4, 0.6, 0.4), (300.0, 80.0, 0.5, 0.5, 0.5, 0.5)]
    branch = ("t. ! queue max-size-buffers=2 ! hsvdetector hue-ref={0} hue-var={1} saturation-ref={2} saturation-var={3} value-ref={4} "
              "value-var={5} ! video/x-raw(memory:HIPMemory),format=RGBA ! hipdownload ! filesink location={6}")
    pipe = (f"hiptestsrc num-buffers={n} ! video/x-raw(memory:HIPMemory),format=RGBx,width={w},height={h},framerate=30/1 ! hsvfilter hue-shift=45 ! tee name=t "
            + branch.format(*det[0], f"{tmp_path}/a.raw") + " " + branch.format(*det[1], f"{tmp_path}/b.raw"))
    for _ in range(3):
        r = gst_env.run([LAUNCH, "-q"] + pipe.split(), tmp_path, timeout=120, extra_env={"MVFX_ELEMENT_PAIR": "2", "MVFX_ELEMENT_PAIR_STATS": "1"})
        assert r.returncode == 0, r.stdout[-2000:]
        for name, settings in zip(("a.raw", "b.raw"), det):
            exp = np.empty_like(mid)
            assert orc.hsvdetector(mid, w * 4, "RGBx", exp, w * 4, "RGBA", w, settings) == 0
            got = np.fromfile(f"{tmp_path}/{name}", dtype=np.uint8).reshape(n, h, w * 4)
            bad = [k for k in range(n) if not np.array_equal(got[k], exp)]
            assert bad == [], f"{name}: frames {bad[:10]} differ"
        stats = re.findall(r"(\w+) \S+: (\d+) device buffers = 2 x (\d+) pair launches \+ (\d+) single launches \+ (\d+) direct launches", r.stdout)
        assert sorted(x[0] for x in stats) == ["hsvdetector", "hsvdetector", "hsvfilter"], r.stdout[-2000:]
        for _name, buffers, pairs, singles, direct in stats:
            assert int(buffers) == n and 2 * int(pairs) + int(singles) + int(direct) == n


@pytest.mark.parametrize("consumer", ["fakesink", "same_thread", "other_thread"])
def test_out_of_place_elements_pair_launches_every_buffer_exactly_once(gpu, tmp_path, consumer):
    """Round 4: hsvdetector and colorlut (NeverInPlace: hsvdetector/imp.rs:380-384, colorlut/imp.rs:162-166) hold one buffer's kernel
    back on device memory and launch two consecutive frames together (mvfx_pair_hold.h); input AND output block carry the mark.
    fakesink: nobody looks, 21 buffers = 10 pairs + one flushed at EOS.  A consumer on the same or on another streaming thread
    (hipdownload) flushes held-back frames or finds them paired; the source refills the recycled input blocks (refresh=true, a pool of
    a few blocks) -- every frame comes out transformed exactly once, byte for byte."""
    w, h, n = 640, 360, 41
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(17))
    det = "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
    det_settings = (120.0, 60.0, 0.6, 0.4, 0.6, 0.4)
    rgbx = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBx,width={w},height={h}", "in_rgbx.raw").reshape(h, w * 4)
    rgba = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,width={w},height={h}", "in_rgba.raw").reshape(h, w * 4)
    exp_det = np.empty_like(rgbx)
    assert orc.hsvdetector(rgbx, w * 4, "RGBx", exp_det, w * 4, "RGBA", w, det_settings) == 0
    exp_lut = np.empty_like(rgba)
    assert orc.CubeLut(cube.read_text()).apply(rgba, w * 4, exp_lut, w * 4, w, h, "RGBA") == 0
    for element, desc, fmt, exp in (("hsvdetector", det, "RGBx", exp_det), ("colorlut", f"colorlut location={cube}", "RGBA", exp_lut)):
        caps = f"video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h},framerate=30/1"
        if consumer == "fakesink":
            for mode in ("2", "1"):
                r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc num-buffers=21 refresh=false ! {caps} ! {desc} ! fakesink sync=false".split(), tmp_path,
                                extra_env={"MVFX_ELEMENT_PAIR_STATS": "1", "MVFX_ELEMENT_PAIR": mode})
                assert r.returncode == 0, r.stdout
                buffers, pairs, singles, direct = _pair_stats_of(element, r.stdout)
                assert buffers == 21 and 2 * pairs + singles + direct == 21
                if mode == "2":
                    assert (pairs, singles, direct) == (10, 1, 0)
            continue
        q = "queue max-size-buffers=3 ! " if consumer == "other_thread" else ""
        out = tmp_path / f"{element}.raw"
        r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc num-buffers={n} ! {caps} ! {desc} ! {q}hipdownload ! filesink location={out}".split(), tmp_path,
                        extra_env={"MVFX_ELEMENT_PAIR_STATS": "1", "MVFX_ELEMENT_PAIR": "1"})
        assert r.returncode == 0, r.stdout
        buffers, pairs, singles, direct = _pair_stats_of(element, r.stdout)
        assert buffers == n and 2 * pairs + singles + direct == n
        if consumer == "same_thread": # every held-back frame is flushed by hipdownload's look: after four of them the element stops holding back
            assert pairs == 0 and singles <= 4 and direct >= n - 4
        got = np.fromfile(out, dtype=np.uint8).reshape(n, h, w * 4)
        assert [k for k in range(n) if not np.array_equal(got[k], exp)] == [], element


def test_out_of_place_pair_launches_can_be_turned_off_and_follow_a_settings_change(gpu, tmp_path):
    """MVFX_ELEMENT_PAIR=0: a launch per buffer, no statistics line.  And a held-back frame keeps the settings in force when ITS buffer
    came: hsvdetector's properties changed between two buffers (tests/gst_worker.py drives the change from the application thread)
    is covered by the hsvdetector_property_change cases of tests/test_gst_inprocess_gpu.py (device chain), which run through the pair path."""
    w, h = 320, 240
    caps = f"video/x-raw(memory:HIPMemory),format=RGBx,width={w},height={h},framerate=30/1"
    rgbx = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBx,width={w},height={h}", "in.raw").reshape(h, w * 4)
    exp = np.empty_like(rgbx)
    assert orc.hsvdetector(rgbx, w * 4, "RGBx", exp, w * 4, "RGBA", w, (0.0, 10.0, 0.0, 0.15, 0.0, 0.3)) == 0
    r = gst_env.run([LAUNCH, "-q"] + f"hiptestsrc num-buffers=5 ! {caps} ! hsvdetector ! hipdownload ! filesink location={tmp_path}/out.raw".split(),
                    tmp_path, extra_env={"MVFX_ELEMENT_PAIR": "0", "MVFX_ELEMENT_PAIR_STATS": "1"})
    assert r.returncode == 0 and "pair launches" not in r.stdout, r.stdout
    got = np.fromfile(f"{tmp_path}/out.raw", dtype=np.uint8).reshape(5, h, w * 4)
    assert all(np.array_equal(got[k], exp) for k in range(5))


def test_pair_launches_of_neighbouring_elements_never_wait_for_each_other(gpu, tmp_path):
    """Three elements that all hold kernels back (MVFX_ELEMENT_PAIR=2: always, also inside a chain, where the default mode stops doing
    it), on three streaming threads, over a pool of a few recycled blocks (refresh=false: the source never looks at them): a held-back
    frame's blocks meet the neighbour's marks all the time.  An element that ran a neighbour's flush under its own lock deadlocked here
    within a few hundred buffers (2 of 25 runs of the cross-thread tests); foreign work is flushed before the lock is taken
    (mvfx_hip_memory_flush_foreign).  Then the default mode on the same pipeline: every buffer exactly once as well."""
    w, h, n = 320, 180, 4000
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(9))
    caps = f"video/x-raw(memory:HIPMemory),format=RGBx,width={w},height={h},framerate=30/1"
    pipe = (f"hiptestsrc num-buffers={n} refresh=false ! {caps} ! hsvfilter hue-shift=30 ! queue max-size-buffers=2 ! hsvdetector hue-var=90 ! "
            f"queue max-size-buffers=2 ! colorlut location={cube} ! fakesink sync=false")
    for mode in ("2", "2", "2", "1"):
        r = gst_env.run([LAUNCH, "-q"] + pipe.split(), tmp_path, timeout=60, extra_env={"MVFX_ELEMENT_PAIR_STATS": "1", "MVFX_ELEMENT_PAIR": mode})
        assert r.returncode == 0, r.stdout[-2000:]
        for element in ("hsvfilter", "hsvdetector", "colorlut"):
            buffers, pairs, singles, direct = _pair_stats_of(element, r.stdout)
            assert buffers == n and 2 * pairs + singles + direct == n
            if mode == "2":
                assert direct == 0


# ---------------------------------------------------------------- devices (round 6): `device-id` and device following

def test_device_id_zero_is_the_default_device(gpu, tmp_path):
    """`device-id=0` on the elements that create device buffers (hiptestsrc, hipupload) gives the bytes of the default; the elements behind
    them take the device from their input memory (one GPU on this box: ordinal 0 either way)"""
    w, h = 320, 240
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(9))
    hip = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=30/1"
    chain = f"hsvfilter hue-shift=45 saturation-mul=1.2 ! colorlut location={cube} ! hipdownload"
    want = _capture(tmp_path, f"hiptestsrc num-buffers=3 ! {hip} ! {chain}", "default.raw")
    got = _capture(tmp_path, f"hiptestsrc num-buffers=3 device-id=0 ! {hip} ! {chain}", "dev0.raw")
    assert want.size == 3 * w * h * 4 and np.array_equal(got, want)
    sysc = f"video/x-raw,format=RGBA,width={w},height={h},framerate=30/1"
    got = _capture(tmp_path, f"hiptestsrc num-buffers=3 ! {sysc} ! hipupload device-id=0 ! {chain}", "up0.raw")
    assert np.array_equal(got, want)
    # and the result is the oracle's
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! {sysc}", "in.raw").reshape(h, w * 4)
    mid = raw.copy()
    assert orc.hsvfilter(mid, w, w * 4, "RGBA", (45.0, 1.2, 0.0, 1.0, 0.0)) == 0
    exp = np.empty_like(mid)
    assert orc.CubeLut(cubes.analytic_3d(9)).apply(mid, w * 4, exp, w * 4, w, h, "RGBA") == 0
    assert np.array_equal(want[:w * h * 4].reshape(h, w * 4), exp)


@pytest.mark.parametrize("pipeline", [
    "hiptestsrc num-buffers=3 device-id=7 ! video/x-raw(memory:HIPMemory),format=RGBA,width=64,height=48 ! hsvfilter ! fakesink",
    "hiptestsrc num-buffers=3 ! video/x-raw,format=RGBA,width=64,height=48 ! hipupload device-id=7 ! hsvfilter ! fakesink"],
    ids=["hiptestsrc", "hipupload"])
def test_a_device_id_that_does_not_exist_is_a_resource_error_in_start(gpu, tmp_path, pipeline):
    """no GPU 7 on this box: the element fails its start() with GST_ELEMENT_ERROR(RESOURCE, NOT_FOUND) -- gst-launch reports the error and
    leaves with status 1; nothing aborts (G_DEBUG=fatal-warnings is on), nothing crashes"""
    if gpu.lib().mvfx_device_count() > 7:
        pytest.skip("this box really has a device 7")
    r = gst_env.run([LAUNCH] + pipeline.split(), tmp_path)
    assert r.returncode in (1, 255), (r.returncode, r.stdout[-1500:])  # gst-launch's "ERROR: Pipeline doesn't want to pause": an exit, no signal
    assert "device-id 7: only" in r.stdout and "HIP device(s) visible" in r.stdout, r.stdout[-1500:]
    assert "Segmentation" not in r.stdout and "Aborted" not in r.stdout


# ---------------------------------------------------------------- the lane in chains across streaming threads (round 6's chain soak)

@pytest.mark.parametrize("pool", ["5", "8"], ids=["odd_pool", "even_pool"])
def test_lane_elements_on_three_streaming_threads_every_frame_checked(gpu, tmp_path, pool):
    """hsvdetector ! queue ! colorlut ! queue ! hipdownload on static device frames (refresh=false: every block holds the same picture, both elements
    write other blocks), 3000 frames, every output frame against the oracle.  Both elements take the direct-dispatch lane on their own streaming threads;
    a recycled block's last dispatch sits in the lane's other queue every so often (always, with the odd pool) and is then waited for ON THE DEVICE by
    a barrier packet (mvfx_direct_queue_wait_event) -- before round 6's soak the streaming thread waited, and before that fix two threads could
    deadlock in the lane (tools/soak_lane_chain.py; profiles/r6/lane_chain_soak.txt).  MVFX_LANE_STATS says what the acquires did."""
    w, h, n = 320, 180, 3000
    cube = tmp_path / "look.cube"
    cube.write_text(cubes.analytic_3d(9))
    det = (120.0, 60.0, 0.6, 0.4, 0.6, 0.4)
    sysc = f"video/x-raw,format=RGBx,width={w},height={h},framerate=30/1"
    raw = _capture(tmp_path, f"hiptestsrc num-buffers=1 ! {sysc}", "in.raw").reshape(h, w * 4)
    mid = np.empty_like(raw)
    assert orc.hsvdetector(raw, w * 4, "RGBx", mid, w * 4, "RGBA", w, det) == 0
    exp = np.empty_like(mid)
    assert orc.CubeLut(cubes.analytic_3d(9)).apply(mid, w * 4, exp, w * 4, w, h, "RGBA") == 0
    out = tmp_path / "out.raw"
    hip = "video/x-raw(memory:HIPMemory)"
    pipe = (f"hiptestsrc num-buffers={n} refresh=false ! {hip},format=RGBx,width={w},height={h},framerate=30/1 ! "
            f"hsvdetector hue-ref={det[0]} hue-var={det[1]} saturation-ref={det[2]} saturation-var={det[3]} value-ref={det[4]} value-var={det[5]} ! "
            f"{hip},format=RGBA ! queue max-size-buffers=3 ! colorlut location={cube} ! queue max-size-buffers=6 ! hipdownload ! filesink location={out}")
    r = gst_env.run([LAUNCH, "-q"] + pipe.split(), tmp_path, timeout=120, extra_env={"MVFX_HIP_POOL_MIN": pool, "MVFX_LANE_STATS": "1"})
    assert r.returncode == 0, r.stdout[-2000:]
    got = np.fromfile(out, dtype=np.uint8)
    assert got.size == n * h * w * 4
    got = got.reshape(n, h, w * 4)
    bad = [k for k in range(n) if not np.array_equal(got[k], exp)]
    assert bad == [], (len(bad), bad[:10])
    stats = [ln for ln in r.stdout.splitlines() if ln.startswith("mvfx lane:")]
    assert stats, r.stdout[-500:]
    taken = int(stats[-1].split()[2])
    if taken == 0:
        pytest.skip("no lane on this box: " + stats[-1])
    # (how many of the 2 x 3000 acquires the lane took depends on how close behind the download is -- a consumer that has to wait for a direct fence on
    #  its thread sends the producer back to its streams for a while: mvfx_direct_discouraged -- ; that it took a good part is all this asserts)
    assert taken >= n // 2, stats[-1]
