#!/usr/bin/env python3
"""In-process GStreamer scenarios (PyGObject + the image's GStreamer 1.14), started by tests/gst_inprocess.py in a child
process with the plugin environment.  Each scenario drives the REAL elements the way an application does -- pad probes,
property changes while PLAYING, caps changes in mid-stream, repeated NULL <-> PLAYING cycles -- and prints one JSON line
with what it saw; the assertions live in the pytest files.  Frames are compared with the oracle here (the worker has the
frames in hand)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import gi  # noqa: E402

gi.require_version("Gst", "1.0")
from gi.repository import Gst  # noqa: E402

from tests import frames  # noqa: E402
from tests import oracle_binding as orc  # noqa: E402

Gst.init(None)


def pull_all(sink):
    """[(caps string, bytes)] of every sample until EOS."""
    out = []
    while True:
        s = sink.emit("try-pull-sample", 20 * Gst.SECOND)   # None at EOS -- and after 20 s without a sample instead of hanging
        if s is None:
            return out
        b = s.get_buffer()
        ok, mi = b.map(Gst.MapFlags.READ)
        assert ok
        out.append((s.get_caps().to_string(), bytes(mi.data)))
        b.unmap(mi)


def wait_eos(pipe, seconds=60):
    msg = pipe.get_bus().timed_pop_filtered(seconds * Gst.SECOND, Gst.MessageType.EOS | Gst.MessageType.ERROR)
    if msg is None:
        return "timeout"
    if msg.type == Gst.MessageType.ERROR:
        err, dbg = msg.parse_error()
        return f"error: {err.message} ({dbg})"
    return "eos"


def caps_size(caps):
    st = Gst.Caps.from_string(caps).get_structure(0)
    return st.get_value("width"), st.get_value("height")


# ---------------------------------------------------------------------------------------------------------- scenarios

def hsvfilter_property_change(chain):
    """hue-shift (and saturation-mul) changed from a pad probe while PLAYING: frames before the change carry the old settings,
    frames after it the new ones (hsvfilter/imp.rs:215-260 takes the settings lock per frame).  `chain` = the elements around
    hsvfilter: '' for system memory, 'hip' for hipupload ! hsvfilter ! hipdownload."""
    w, h, n, switch = 160, 120, 8, 4
    a = (45.0, 1.0, 0.0, 1.0, 0.0)
    b = (-120.0, 1.5, 0.1, 0.8, 0.05)
    pre, post = ("hipupload ! ", " ! hipdownload") if chain == "hip" else ("", "")
    pipe = Gst.parse_launch(f"videotestsrc num-buffers={n} ! video/x-raw,format=RGBA,width={w},height={h} ! {pre}"
                            f"hsvfilter name=f hue-shift={a[0]} ! {post.lstrip(' !')}{' ! ' if post else ''}appsink name=sink sync=false")
    f, sink = pipe.get_by_name("f"), pipe.get_by_name("sink")
    seen = [0]

    def probe(pad, info):
        if seen[0] == switch:
            for name, v in zip(("hue-shift", "saturation-mul", "saturation-off", "value-mul", "value-off"), b):
                f.set_property(name, v)
        seen[0] += 1
        return Gst.PadProbeReturn.OK

    f.get_static_pad("sink").add_probe(Gst.PadProbeType.BUFFER, probe)
    pipe.set_state(Gst.State.PLAYING)
    got = pull_all(sink)
    pipe.set_state(Gst.State.NULL)
    src, _ = frames.videotestsrc_smpte(w, h, n)
    mismatches = []
    for i, (_, data) in enumerate(got):
        want = src[i].copy()
        assert orc.hsvfilter(want, w, w * 4, "RGBA", a if i < switch else b) == 0
        if not np.array_equal(np.frombuffer(data, dtype=np.uint8).reshape(h, w * 4), want):
            mismatches.append(i)
    return {"frames": len(got), "mismatches": mismatches}


def renegotiate(chain):
    """The capsfilter's size changes in mid-stream (videotestsrc renegotiates): every frame after the change has the new
    size and is still right.  chain: 'hsv' (system memory hsvfilter), 'hsv-hip' (device memory), 'rounded' (roundedcorners:
    the cairo mask must be re-rendered for the new size)."""
    sizes = [(96, 64), (160, 120)]
    n, switch = 8, 4
    if chain == "rounded":
        body, fmt = "roundedcorners border-radius-px=12", "I420"
    elif chain == "hsv-hip":
        body, fmt = "hipupload ! hsvfilter hue-shift=30 ! hipdownload", "RGBA"
    else:
        body, fmt = "hsvfilter hue-shift=30", "RGBA"
    pipe = Gst.parse_launch(f"videotestsrc num-buffers={n} ! capsfilter name=cf caps=video/x-raw,format={fmt},width={sizes[0][0]},"
                            f"height={sizes[0][1]} ! {body} ! appsink name=sink sync=false")
    cf, sink = pipe.get_by_name("cf"), pipe.get_by_name("sink")
    seen = [0]

    def probe(pad, info):
        seen[0] += 1
        if seen[0] == switch:
            cf.set_property("caps", Gst.Caps.from_string(f"video/x-raw,format={fmt},width={sizes[1][0]},height={sizes[1][1]}"))
        return Gst.PadProbeReturn.OK

    cf.get_static_pad("src").add_probe(Gst.PadProbeType.BUFFER, probe)
    pipe.set_state(Gst.State.PLAYING)
    got = pull_all(sink)
    pipe.set_state(Gst.State.NULL)
    seen_sizes, bad = [], []
    settings = (30.0, 1.0, 0.0, 1.0, 0.0)
    for i, (caps, data) in enumerate(got):
        w, h = caps_size(caps)
        if not seen_sizes or seen_sizes[-1] != [w, h]:
            seen_sizes.append([w, h])
        arr = np.frombuffer(data, dtype=np.uint8)
        if chain == "rounded":
            # A420: the alpha plane is the cairo mask of THIS size
            import _pkg
            vfx = _pkg.vfx
            mask = np.zeros((h, w), dtype=np.uint8)
            assert vfx.lib().mvfx_roundedcorners_mask_host(mask.ctypes.data, w, h, w, 12) == 0
            _, _, _, off = a420_layout(w, h)
            if not np.array_equal(arr[off:off + w * h].reshape(h, w), mask):
                bad.append(i)
        else:
            # videotestsrc's snow generator runs on across the renegotiation (and the frame size changes how much of it a frame
            # consumes), so the bars -- a function of position only -- are compared with the oracle's output and the snow
            # rectangle through a property: see below
            src, _ = frames.videotestsrc_smpte(w, h, 1)
            want = src[0].copy()
            assert orc.hsvfilter(want, w, w * 4, "RGBA", settings) == 0
            x0, y0 = frames.vts_snow_geometry(w, h)
            g = arr.reshape(h, w, 4).copy()
            wv = want.reshape(h, w, 4).copy()
            g[y0:, x0:] = 0
            wv[y0:, x0:] = 0
            if not np.array_equal(g, wv):
                bad.append(i)
            # snow is grey (r == g == b): hsvfilter maps grey v to grey clamp(v * value_mul + value_off) -- with these settings
            # value is unchanged, so the snow must still be grey with alpha 255
            snow = arr.reshape(h, w, 4)[y0:, x0:]
            if not (np.array_equal(snow[..., 0], snow[..., 1]) and np.array_equal(snow[..., 1], snow[..., 2]) and (snow[..., 3] == 255).all()):
                bad.append(i)
    return {"frames": len(got), "sizes": seen_sizes, "mismatches": sorted(set(bad))}


def a420_layout(w, h):
    """(y, u, v, a) plane offsets of GStreamer's A420 for even sizes with w % 8 == 0 (strides: w, w/2, w/2, w)."""
    return 0, w * h, w * h + (w // 2) * (h // 2), w * h + 2 * (w // 2) * (h // 2)


def rounded_radius_change(_arg):
    """border-radius-px changed while PLAYING (mutable in PLAYING, border/imp.rs:299-310): the mask follows, and 0 turns the
    element into an I420 passthrough (the src caps are reconfigured)."""
    w, h, n = 96, 64, 9
    radii = {0: 10, 3: 30, 6: 0}          # buffer index at which the probe sets the radius
    pipe = Gst.parse_launch(f"videotestsrc num-buffers={n} ! video/x-raw,format=I420,width={w},height={h} ! "
                            "roundedcorners name=r border-radius-px=10 ! appsink name=sink sync=false")
    r, sink = pipe.get_by_name("r"), pipe.get_by_name("sink")
    seen = [0]

    def probe(pad, info):
        if seen[0] in radii:
            r.set_property("border-radius-px", radii[seen[0]])
        seen[0] += 1
        return Gst.PadProbeReturn.OK

    r.get_static_pad("sink").add_probe(Gst.PadProbeType.BUFFER, probe)
    pipe.set_state(Gst.State.PLAYING)
    got = pull_all(sink)
    pipe.set_state(Gst.State.NULL)
    import _pkg
    vfx = _pkg.vfx
    formats, bad = [], []
    for i, (caps, data) in enumerate(got):
        fmt = Gst.Caps.from_string(caps).get_structure(0).get_value("format")
        formats.append(fmt)
        radius = 10 if i < 3 else (30 if i < 6 else 0)
        arr = np.frombuffer(data, dtype=np.uint8)
        if radius == 0:
            if fmt != "I420" or arr.size != w * h * 3 // 2:
                bad.append(i)
            continue
        mask = np.zeros((h, w), dtype=np.uint8)
        assert vfx.lib().mvfx_roundedcorners_mask_host(mask.ctypes.data, w, h, w, radius) == 0
        off = a420_layout(w, h)[3]
        if fmt != "A420" or not np.array_equal(arr[off:off + w * h].reshape(h, w), mask):
            bad.append(i)
    return {"frames": len(got), "formats": formats, "mismatches": bad}


def hsvdetector_property_change(chain):
    """All six hsvdetector properties changed from a pad probe while PLAYING (hsvdetector/imp.rs:254-320)."""
    w, h, n, switch = 160, 120, 8, 3
    a = (0.0, 10.0, 0.0, 0.15, 0.0, 0.3)          # the defaults
    b = (120.0, 60.0, 0.6, 0.5, 0.5, 0.6)
    pre, post = ("hipupload ! ", "hipdownload ! ") if chain == "hip" else ("", "")
    pipe = Gst.parse_launch(f"videotestsrc num-buffers={n} ! video/x-raw,format=RGBx,width={w},height={h} ! {pre}"
                            f"hsvdetector name=d ! {post}video/x-raw,format=RGBA ! appsink name=sink sync=false")
    d, sink = pipe.get_by_name("d"), pipe.get_by_name("sink")
    seen = [0]

    def probe(pad, info):
        if seen[0] == switch:
            for name, v in zip(("hue-ref", "hue-var", "saturation-ref", "saturation-var", "value-ref", "value-var"), b):
                d.set_property(name, v)
        seen[0] += 1
        return Gst.PadProbeReturn.OK

    d.get_static_pad("sink").add_probe(Gst.PadProbeType.BUFFER, probe)
    pipe.set_state(Gst.State.PLAYING)
    got = pull_all(sink)
    pipe.set_state(Gst.State.NULL)
    src, _ = frames.videotestsrc_smpte(w, h, n)   # RGBx from videotestsrc carries x = 255, like RGBA's alpha
    bad, opaque = [], []
    for i, (_, data) in enumerate(got):
        want = np.zeros((h, w * 4), dtype=np.uint8)
        assert orc.hsvdetector(src[i], w * 4, "RGBx", want, w * 4, "RGBA", w, a if i < switch else b) == 0
        out = np.frombuffer(data, dtype=np.uint8).reshape(h, w * 4)
        if not np.array_equal(out, want):
            bad.append(i)
        opaque.append(int(np.count_nonzero(out[:, 3::4])))
    return {"frames": len(got), "mismatches": bad, "opaque_pixels": opaque}


def overlay_property_change(chain):
    """imagersoverlay: alpha and offsets changed while PLAYING (mutable in PLAYING, overlay/imp.rs:317-420): every frame is
    blended with the values in force when it passes."""
    from PIL import Image
    w, h, n, switch = 320, 240, 6, 3
    tmp = os.environ.get("MVFX_WORKER_TMP", "/tmp")
    logo = os.path.join(tmp, "logo.png")
    rng = np.random.default_rng(7)
    rgba = rng.integers(0, 256, (32, 48, 4), dtype=np.uint8)
    rgba[..., 3] = np.linspace(0, 255, 48, dtype=np.uint8)[None, :]
    Image.fromarray(rgba, "RGBA").save(logo)
    bgra = np.ascontiguousarray(rgba[..., [2, 1, 0, 3]]).reshape(32, 48 * 4)
    pre, post = ("hipupload ! ", "hipdownload ! ") if chain == "hip" else ("", "")
    pipe = Gst.parse_launch(f"videotestsrc num-buffers={n} ! video/x-raw,format=RGBA,width={w},height={h} ! {pre}"
                            f"imagersoverlay name=o location={logo} offset-x=10 offset-y=20 alpha=1.0 ! {post}video/x-raw,format=RGBA ! "
                            "appsink name=sink sync=false")
    o, sink = pipe.get_by_name("o"), pipe.get_by_name("sink")
    seen = [0]

    def probe(pad, info):
        if seen[0] == switch:
            o.set_property("alpha", 0.4)
            o.set_property("offset-x", 100)
            o.set_property("offset-y", -30)      # from the bottom edge
        seen[0] += 1
        return Gst.PadProbeReturn.OK

    o.get_static_pad("sink").add_probe(Gst.PadProbeType.BUFFER, probe)
    pipe.set_state(Gst.State.PLAYING)
    got = pull_all(sink)
    pipe.set_state(Gst.State.NULL)
    src, _ = frames.videotestsrc_smpte(w, h, n)
    bad = []
    for i, (_, data) in enumerate(got):
        want = src[i].copy()
        x, y, alpha = (10, 20, 1.0) if i < switch else (100, h - 30 - 32, 0.4)
        assert orc.overlay_blend(want, w, h, w * 4, "RGBA", bgra, 48, 32, x, y, alpha) == 0
        if not np.array_equal(np.frombuffer(data, dtype=np.uint8).reshape(h, w * 4), want):
            bad.append(i)
    return {"frames": len(got), "mismatches": bad}


def videocompare_three_pads(_arg):
    """The aggregator the way an application builds it (videocompare/imp.rs:188-256, 283-377): request pads sink_0..2 by
    template, sink_0 is the reference; red vs red vs smpte at max-dist-threshold 0 -> every aggregate posts ONE element message
    that lists BOTH other pads (distance 0 for the red one, > 0 for the colour bars; a solid blue frame would hash like the red one:
    blockhash of a constant image is constant); then release a pad and run again."""
    import re

    def build(patterns, n=4):
        pipe = Gst.Pipeline.new(None)
        vc = Gst.ElementFactory.make("videocompare", "compare")
        sink = Gst.ElementFactory.make("fakesink", None)
        pipe.add(vc)
        pipe.add(sink)
        assert vc.link(sink)
        pads = []
        for k, pat in enumerate(patterns):
            src = Gst.ElementFactory.make("videotestsrc", None)
            src.set_property("num-buffers", n)
            Gst.util_set_object_arg(src, "pattern", pat)
            cf = Gst.ElementFactory.make("capsfilter", None)
            cf.set_property("caps", Gst.Caps.from_string("video/x-raw,format=RGBA,width=320,height=240,framerate=30/1"))
            pipe.add(src)
            pipe.add(cf)
            assert src.link(cf)
            pad = vc.get_request_pad("sink_%u")
            assert pad is not None and pad.get_name() == f"sink_{k}", pad.get_name() if pad else None
            assert cf.get_static_pad("src").link(pad) == Gst.PadLinkReturn.OK
            pads.append(pad)
        return pipe, vc, pads

    def run(pipe):
        pipe.set_state(Gst.State.PLAYING)
        bus, msgs = pipe.get_bus(), []
        while True:
            m = bus.timed_pop_filtered(30 * Gst.SECOND, Gst.MessageType.EOS | Gst.MessageType.ERROR | Gst.MessageType.ELEMENT)
            if m is None or m.type == Gst.MessageType.EOS:
                break
            if m.type == Gst.MessageType.ERROR:
                msgs.append("ERROR " + m.parse_error()[0].message)
                break
            st = m.get_structure()
            if st is not None and st.get_name() == "videocompare":
                text = st.to_string()
                found = {name: float(d) for name, d in re.findall(r"\b(sink_\d+).*?distance\W+double\W+([0-9.e+-]+)", text)}
                msgs.append(found if found else "UNPARSED " + text)
        pipe.set_state(Gst.State.NULL)
        return msgs

    pipe, vc, pads = build(["red", "red", "smpte"])
    first = run(pipe)
    # second pipeline: the same three branches, then the third pad released before PLAYING -> two-pad aggregates
    pipe2, vc2, pads2 = build(["red", "red", "smpte"])
    peer = pads2[2].get_peer()
    peer.unlink(pads2[2])
    vc2.release_request_pad(pads2[2])
    # the unlinked third branch would stop the pipeline with not-linked: give it a sink of its own
    fs = Gst.ElementFactory.make("fakesink", None)
    pipe2.add(fs)
    assert peer.link(fs.get_static_pad("sink")) == Gst.PadLinkReturn.OK
    second = run(pipe2)
    return {"first": first, "second": second, "sink_pads_after_release": sorted(p.get_name() for p in vc2.sinkpads)}


def colorlut_relocation(chain):
    """`location` is mutable in READY (colorlut/imp.rs:118-140): a pipeline taken to READY with one file configured, then given another
    file (another size: every device table differs) and played -- it must grade with the second LUT; a control run keeps the first."""
    from tests import cubes
    w, h, n = 160, 120, 3
    tmp = os.environ.get("MVFX_WORKER_TMP", "/tmp")
    texts = [cubes.analytic_3d(17), cubes.identity_3d(33).replace("LUT_3D_SIZE 33", "LUT_3D_SIZE 33\nDOMAIN_MIN 0 0 0\nDOMAIN_MAX 1 1 1")]
    paths = []
    for k, t in enumerate(texts):
        paths.append(os.path.join(tmp, f"lut{k}.cube"))
        with open(paths[-1], "w") as f:
            f.write(t)
    pre, post = ("hipupload ! ", "hipdownload ! ") if chain == "hip" else ("", "")
    desc = (f"videotestsrc num-buffers={n} ! video/x-raw,format=RGBA,width={w},height={h} ! {pre}colorlut name=l location={paths[0]} ! "
            f"{post}video/x-raw,format=RGBA ! appsink name=sink sync=false")
    src, _ = frames.videotestsrc_smpte(w, h, 1)      # the bars; the snow differs from frame to frame: left out of the comparison
    runs = []
    for k in range(2):
        pipe = Gst.parse_launch(desc)
        lut, sink = pipe.get_by_name("l"), pipe.get_by_name("sink")
        if k == 1:   # READY with the first file configured, then the other one
            pipe.set_state(Gst.State.READY)
            pipe.get_state(5 * Gst.SECOND)
            lut.set_property("location", paths[1])
        pipe.set_state(Gst.State.PLAYING)
        got = pull_all(sink)
        pipe.set_state(Gst.State.NULL)
        o = orc.CubeLut(texts[k])
        x0, y0 = frames.vts_snow_geometry(w, h)
        bad = []
        for i, (_, data) in enumerate(got):
            out = np.frombuffer(data, dtype=np.uint8).reshape(h, w * 4)
            want = np.empty_like(src[0])
            assert o.apply(src[0], w * 4, want, w * 4, w, h, "RGBA") == 0
            a = out.reshape(h, w, 4).copy()
            b = want.reshape(h, w, 4).copy()
            a[y0:, x0:] = 0
            b[y0:, x0:] = 0
            if not np.array_equal(a, b):
                bad.append(i)
        runs.append({"frames": len(got), "mismatches": bad})
    # the two LUTs must actually differ on the bars, or the test proves nothing
    w0, w1 = np.empty_like(src[0]), np.empty_like(src[0])
    orc.CubeLut(texts[0]).apply(src[0], w * 4, w0, w * 4, w, h, "RGBA")
    orc.CubeLut(texts[1]).apply(src[0], w * 4, w1, w * 4, w, h, "RGBA")
    return {"runs": runs, "luts_differ": bool(np.count_nonzero(w0 != w1) > 1000)}


def state_cycles(n_cycles):
    """NULL -> PLAYING -> EOS -> NULL over and over on the device-memory chain: nothing may accumulate on the device (pools and
    allocator freelists are released in stop() / trimmed)."""
    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(), ctypes.c_size_t()

    def free_mb():
        assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
        return free_b.value / 1e6

    from tests import cubes
    cube = os.path.join(os.environ.get("MVFX_WORKER_TMP", "/tmp"), "cycle.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(17))
    desc = ("hiptestsrc num-buffers=12 ! video/x-raw,format=RGBx,width=1920,height=1080 ! hipupload ! hsvfilter hue-shift=20 ! "
            "hsvdetector ! video/x-raw(memory:HIPMemory),format=RGBA ! colorlut location=" + cube + " ! hipdownload ! fakesink sync=false")
    series, results = [], []
    for c in range(n_cycles):
        pipe = Gst.parse_launch(desc)
        pipe.set_state(Gst.State.PLAYING)
        results.append(wait_eos(pipe))
        pipe.set_state(Gst.State.NULL)
        del pipe
        series.append(free_mb())
    return {"results": sorted(set(results)), "free_mb": series}


def held_back_failure(which):
    """Pair launches on (MVFX_ELEMENT_PAIR=2), a live source at 5 frames/s, the idle interval forced to 20 ms, frames on which the
    reference panics (RGB 641x1: plane size 1924 is no multiple of 3, hsvfilter/imp.rs:92, hsvdetector/imp.rs:122).  Frame 0 is held
    back, the timer launches it alone, the launch fails.  Unlike gst-launch this application does NOT stop at the first error message:
    it keeps the pipeline PLAYING and records everything the bus says for 1.5 s, and how many buffers entered the element."""
    import time
    os.environ.update(MVFX_ELEMENT_PAIR="2", MVFX_PAIR_IDLE_US="20000")
    el = "hsvfilter hue-shift=10" if which == "hsvfilter" else "hsvdetector"
    pipe = Gst.parse_launch("hiptestsrc name=src is-live=true num-buffers=6 ! video/x-raw(memory:HIPMemory),format=RGB,width=641,height=1,framerate=5/1 ! "
                            + el + " name=e ! fakesink")
    seen = {"in": 0, "out": 0}

    def count(key):
        def probe(pad, info):
            seen[key] += 1
            return Gst.PadProbeReturn.OK
        return probe
    e = pipe.get_by_name("e")
    e.get_static_pad("sink").add_probe(Gst.PadProbeType.BUFFER, count("in"))
    e.get_static_pad("src").add_probe(Gst.PadProbeType.BUFFER, count("out"))
    bus = pipe.get_bus()
    pipe.set_state(Gst.State.PLAYING)
    errors, t_end = [], time.time() + 1.5
    while time.time() < t_end:
        msg = bus.timed_pop_filtered(50 * Gst.MSECOND, Gst.MessageType.ERROR | Gst.MessageType.EOS)
        if msg is not None and msg.type == Gst.MessageType.ERROR:
            err, dbg = msg.parse_error()
            errors.append({"src": msg.src.get_name(), "message": err.message, "debug": dbg or ""})
    pipe.set_state(Gst.State.NULL)
    return {"errors": errors, "buffers_in": seen["in"], "buffers_out": seen["out"]}


SCENARIOS = {"hsvfilter_property_change": hsvfilter_property_change, "renegotiate": renegotiate,
             "rounded_radius_change": rounded_radius_change, "hsvdetector_property_change": hsvdetector_property_change,
             "overlay_property_change": overlay_property_change, "videocompare_three_pads": videocompare_three_pads,
             "colorlut_relocation": colorlut_relocation, "held_back_failure": held_back_failure,
             "state_cycles": lambda n: state_cycles(int(n))}

if __name__ == "__main__":
    name, arg = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    print("RESULT " + json.dumps(SCENARIOS[name](arg)), flush=True)
