"""The direct-dispatch lane (round 6, MVFX_OPT_DIRECT_DISPATCH; gst-plugin-rs_amd/csrc/direct_dispatch.h): one-frame hsvfilter calls as AQL packets
without the barrier bit on the library's own HSA queue, the frame's completion carried by the thread's completion event (a "direct fence").
New kernels (csrc/direct/hsv_direct_kernels.hip) => their own proofs: all 2^24 colours against the oracle; and the fence semantics the element
layer relies on."""
import ctypes

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu

BENCH = (90.0, 1.25, -0.05, 0.9, 0.02)
NEG = (-123.4, 0.5, 0.3, 1.7, -0.2)


class Event:
    def __init__(self, vfx):
        self.vfx, self.h = vfx, ctypes.c_void_p()
        vfx.check(vfx.lib().mvfx_event_create(ctypes.byref(self.h)))

    def __del__(self):
        try:
            self.vfx.lib().mvfx_event_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass


def direct_filter(vfx, ptr, w, h, stride, fmt, settings, ev, nontemporal=False, stream=None):
    """one call with the lane allowed; returns (status, launches that carried the event, is_direct)"""
    L = vfx.lib()
    f = vfx.make_frame(ptr, w, h, stride, fmt)
    s = vfx.HsvFilterSettings(*settings)
    vfx.check(L.mvfx_thread_set_options(vfx.options(nontemporal=nontemporal, direct=True).word))
    try:
        vfx.check(L.mvfx_thread_set_completion_event(ev.h))
        rc = L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(s), stream)
        carried = L.mvfx_thread_clear_completion_event()
    finally:
        L.mvfx_thread_set_options(0)
    return rc, carried, L.mvfx_event_is_direct(ev.h)


def test_lane_takes_a_flat_4k_frame_and_the_event_is_its_fence(gpu):
    w, h = 3840, 2160
    vts, _ = frames.videotestsrc_smpte(w, h, 1)
    want = vts[0].copy()
    assert orc.hsvfilter(want, w, w * 4, "RGBA", BENCH) == 0
    buf = gpu.DeviceBuffer(vts[0].nbytes).upload(vts[0])
    ev = Event(gpu)
    L = gpu.lib()
    assert L.mvfx_event_is_direct(ev.h) == 0
    rc, carried, direct = direct_filter(gpu, buf.ptr, w, h, w * 4, "RGBA", BENCH, ev)
    assert rc == 0 and carried == 1
    if direct != 1:
        pytest.fail("the lane did not take an eligible frame on this box (no HSA queue?): " + gpu.last_error())
    gpu.check(L.mvfx_event_synchronize(ev.h))   # NOT a stream synchronise: nothing was enqueued on a stream
    assert L.mvfx_event_query(ev.h) == 1
    assert np.array_equal(buf.download().reshape(h, w * 4), want)
    # a HIP stream "waiting" for the fired fence: returns at once, and a kernel of that stream sees the filtered frame
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    gpu.check(L.mvfx_stream_wait_event(st, ev.h))
    out = gpu.DeviceBuffer(h * w * 4)
    fi, fo = gpu.make_frame(buf.ptr, w, h, w * 4, "RGBx"), gpu.make_frame(out.ptr, w, h, w * 4, "RGBA")
    ds = (120.0, 40.0, 0.6, 0.4, 0.6, 0.4)
    gpu.check(L.mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(gpu.HsvDetectorSettings(*ds)), st))
    gpu.check(L.mvfx_stream_synchronize(st))
    exp = np.empty_like(want)
    assert orc.hsvdetector(want, w * 4, "RGBx", exp, w * 4, "RGBA", w, ds) == 0
    assert np.array_equal(out.download().reshape(h, w * 4), exp)
    # the same event recorded the ordinary way afterwards is an ordinary event again
    gpu.check(L.mvfx_event_record(ev.h, st))
    assert L.mvfx_event_is_direct(ev.h) == 0
    gpu.check(L.mvfx_event_synchronize(ev.h))


@pytest.mark.parametrize("fmt", ["RGBA", "xBGR", "BGRx", "ARGB"])
@pytest.mark.parametrize("settings", [BENCH, NEG, (0.0, 1.0, 0.0, 1.0, 0.0), (360.0, 1.0, 0.0, 1.0, 0.0)], ids=["bench", "negative", "defaults", "360"])
@pytest.mark.parametrize("nontemporal", [False, True], ids=["cached", "nt"])
def test_lane_kernels_on_all_2_24_colours(gpu, fmt, settings, nontemporal):
    """mvfx_direct_hsvfilter4_{pos,neg}{,_nt}: every (R, G, B) triple, four byte layouts, both hue-shift signs, both cache policies, against the oracle"""
    ex = frames.exhaustive_rgbx()
    want = ex.copy()
    assert orc.hsvfilter(want, 4096, 4096 * 4, fmt, settings) == 0
    buf = gpu.DeviceBuffer(ex.nbytes).upload(ex)
    ev = Event(gpu)
    rc, carried, direct = direct_filter(gpu, buf.ptr, 4096, 4096, 4096 * 4, fmt, settings, ev, nontemporal)
    assert rc == 0 and carried == 1 and direct == 1
    gpu.check(gpu.lib().mvfx_event_synchronize(ev.h))
    got = buf.download().reshape(want.shape)
    assert np.array_equal(got, want), int(np.count_nonzero(got != want))


@pytest.mark.parametrize("case", [("RGBA", 1918, 9, 0), ("RGBA", 64, 48, 32), ("RGB", 64, 48, 0), ("RGBA", 2, 2, 0), ("RGBA", 4, 1, 0), ("RGBA", 1021, 7, 0)],
                         ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}+pad{c[3]}")
def test_frames_the_lane_cannot_take_run_on_the_stream_with_the_same_bytes(gpu, case):
    """row padding, 3-byte pixels, widths that are not multiples of four: launched on `stream` as if the bit were clear -- the event is then an ordinary
    stop event; tiny flat frames the lane does take.  Bytes equal the oracle's either way; the row padding stays untouched."""
    fmt, w, h, pad = case
    bpp = 3 if fmt == "RGB" else 4
    stride = w * bpp + pad
    f = frames.random_frame(0x5EED0C00 + w, w, h, bpp, stride)
    want = f.copy()
    assert orc.hsvfilter(want, w, stride, fmt, BENCH) == 0
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    ev = Event(gpu)
    st = ctypes.c_void_p(gpu.lib().mvfx_thread_stream())
    rc, carried, direct = direct_filter(gpu, buf.ptr, w, h, stride, fmt, BENCH, ev, stream=st)
    assert rc == 0 and carried == 1
    eligible = bpp == 4 and pad == 0 and (w * h) % 4 == 0   # an unpadded frame is one run of pixels: its length counts, not the row's
    assert direct == (1 if eligible else 0), (case, direct)
    gpu.check(gpu.lib().mvfx_event_synchronize(ev.h))
    assert np.array_equal(buf.download().reshape(h, stride), want)


def test_settings_outside_the_fast_domain_and_calls_without_an_event_stay_on_the_stream(gpu):
    w, h = 256, 64
    f = frames.random_frame(0x5EED0C77, w, h)
    L = gpu.lib()
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    for settings in ((720.5, 1.0, 0.0, 1.0, 0.0), (float("nan"), 1.0, 0.0, 1.0, 0.0)):
        want = f.copy()
        assert orc.hsvfilter(want, w, w * 4, "RGBA", settings) == 0
        buf = gpu.DeviceBuffer(f.nbytes).upload(f)
        ev = Event(gpu)
        rc, carried, direct = direct_filter(gpu, buf.ptr, w, h, w * 4, "RGBA", settings, ev, stream=st)
        assert rc == 0 and carried == 1 and direct == 0
        gpu.check(L.mvfx_event_synchronize(ev.h))
        assert np.array_equal(buf.download().reshape(h, w * 4), want)
    # the bit without a completion event: the caller orders by its stream, so the stream it is
    want = f.copy()
    assert orc.hsvfilter(want, w, w * 4, "RGBA", BENCH) == 0
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    with gpu.options(direct=True):
        gpu.hsvfilter_device(buf.ptr, w, h, w * 4, "RGBA", gpu.HsvFilterSettings(*BENCH), stream=st)
        gpu.check(L.mvfx_stream_synchronize(st))
    assert np.array_equal(buf.download().reshape(h, w * 4), want)


def test_many_frames_in_flight_each_with_its_own_fence(gpu):
    """96 frames back to back through the lane (nothing orders them: no barrier bit), 32 events in rotation -- an event is re-used only after its
    frame has been seen finished, as the element layer's fence pool does; every frame equals the oracle's"""
    w, h, n, nev = 1920, 1080, 96, 32
    src = [frames.random_frame(0x5EED0D00 + k, w, h) for k in range(4)]
    want = []
    for k in range(4):
        x = src[k].copy()
        assert orc.hsvfilter(x, w, w * 4, "RGBA", BENCH) == 0
        want.append(x)
    bufs = [gpu.DeviceBuffer(src[k % 4].nbytes).upload(src[k % 4]) for k in range(n)]
    evs = [Event(gpu) for _ in range(nev)]
    L = gpu.lib()
    for k in range(n):
        e = evs[k % nev]
        if k >= nev:
            gpu.check(L.mvfx_event_synchronize(e.h))
        rc, carried, direct = direct_filter(gpu, bufs[k].ptr, w, h, w * 4, "RGBA", BENCH, e)
        assert rc == 0 and carried == 1 and direct == 1
    for e in evs:
        gpu.check(L.mvfx_event_synchronize(e.h))
        assert L.mvfx_event_query(e.h) == 1
    for k in range(n):
        assert np.array_equal(bufs[k].download().reshape(h, w * 4), want[k % 4]), k


def test_a_pending_direct_fence_makes_the_waiting_thread_wait_not_the_stream(gpu):
    """mvfx_stream_wait_event on a direct fence that has not fired: the call returns only when it has (a HIP stream cannot wait for an HSA signal on
    the device), so work enqueued on the stream afterwards is ordered behind the frame"""
    w, h = 7680, 4320
    f = frames.random_frame(0x5EED0C99, w, h)
    want = f.copy()
    assert orc.hsvfilter(want, w, w * 4, "RGBA", BENCH) == 0
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    ev = Event(gpu)
    L = gpu.lib()
    rc, carried, direct = direct_filter(gpu, buf.ptr, w, h, w * 4, "RGBA", BENCH, ev)
    assert rc == 0 and direct == 1
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    gpu.check(L.mvfx_stream_wait_event(st, ev.h))
    assert L.mvfx_event_query(ev.h) == 1          # the wait was the thread's
    copy = gpu.DeviceBuffer(f.nbytes)
    gpu.check(L.mvfx_copy_device_to_device_async(ctypes.c_void_p(copy.ptr), ctypes.c_void_p(buf.ptr), f.nbytes, st))
    gpu.check(L.mvfx_stream_synchronize(st))
    assert np.array_equal(copy.download().reshape(h, w * 4), want)


# ---------------------------------------------------------------- hsvdetector through the lane, and dependencies by queue order

DET = (120.0, 40.0, 0.6, 0.4, 0.6, 0.4)


def direct_detect(vfx, src_ptr, dst_ptr, w, h, in_fmt, out_fmt, settings, ev, stream, only=False):
    L = vfx.lib()
    fi, fo = vfx.make_frame(src_ptr, w, h, w * 4, in_fmt), vfx.make_frame(dst_ptr, w, h, w * 4, out_fmt)
    vfx.check(L.mvfx_thread_set_options(vfx.OPT_DIRECT_DISPATCH | (vfx.OPT_DIRECT_ONLY if only else 0)))
    try:
        vfx.check(L.mvfx_thread_set_completion_event(ev.h))
        rc = L.mvfx_hsvdetector_transform_frame(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(vfx.HsvDetectorSettings(*settings)), stream)
        carried = L.mvfx_thread_clear_completion_event()
    finally:
        L.mvfx_thread_set_options(0)
    return rc, carried, L.mvfx_event_is_direct(ev.h)


@pytest.mark.parametrize("in_fmt,out_fmt", [("RGBx", "RGBA"), ("xBGR", "ARGB"), ("BGRx", "BGRA"), ("xRGB", "ABGR")])
@pytest.mark.parametrize("settings", [DET, (540.0, 0.0, 0.5, 0.5, 0.5, 0.5), (-180.0, 180.0, 0.25, 0.0, 1.0, 0.0), (350.0, 25.0, 0.5, 0.5, 0.5, 0.5)],
                         ids=["bench", "edge-540", "edge-minus180", "wrap"])
def test_lane_detector_on_all_2_24_colours(gpu, in_fmt, out_fmt, settings):
    """mvfx_direct_hsvdetector4: every (R, G, B) triple, four layout pairs, bench settings + both edges of the hue test's domain + a wrap through 0"""
    ex = frames.exhaustive_rgbx()
    want = np.empty_like(ex)
    assert orc.hsvdetector(ex, 4096 * 4, in_fmt, want, 4096 * 4, out_fmt, 4096, settings) == 0
    src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
    dst = gpu.DeviceBuffer(ex.nbytes)
    ev = Event(gpu)
    st = ctypes.c_void_p(gpu.lib().mvfx_thread_stream())
    rc, carried, direct = direct_detect(gpu, src.ptr, dst.ptr, 4096, 4096, in_fmt, out_fmt, settings, ev, st)
    assert rc == 0 and carried == 1 and direct == 1
    gpu.check(gpu.lib().mvfx_event_synchronize(ev.h))
    got = dst.download().reshape(want.shape)
    assert np.array_equal(got, want), int(np.count_nonzero(got != want))


def test_filter_then_detector_on_one_lane_queue_need_no_wait_in_between(gpu):
    """The element chain's pattern: hsvfilter (in place) and hsvdetector (reads the filtered frame) of ONE frame are dispatched with the same stream
    hint, hence on the same in-order lane queue: the detector is enqueued while the filter may still be running, and nothing waits on the host.
    Eight frames alternate between the two queues (the streams' parity), each pair checked against the oracle chain."""
    L = gpu.lib()
    w, h, n = 3840, 2160, 8
    src = [frames.random_frame(0x5EED0F00 + k, w, h) for k in range(2)]
    want = []
    for k in range(2):
        mid = src[k].copy()
        assert orc.hsvfilter(mid, w, w * 4, "RGBx", BENCH) == 0
        out = np.empty_like(mid)
        assert orc.hsvdetector(mid, w * 4, "RGBx", out, w * 4, "RGBA", w, DET) == 0
        want.append((mid, out))
    bufs = [gpu.DeviceBuffer(src[k % 2].nbytes).upload(src[k % 2]) for k in range(n)]
    outs = [gpu.DeviceBuffer(h * w * 4) for _ in range(n)]
    fev, dev_ = [Event(gpu) for _ in range(n)], [Event(gpu) for _ in range(n)]
    streams = [ctypes.c_void_p(L.mvfx_thread_stream_n(k)) for k in range(2)]
    assert {L.mvfx_direct_queue_of_stream(s) for s in streams} == {0, 1}
    for k in range(n):
        st = streams[k & 1]
        rc, _, direct = direct_filter(gpu, bufs[k].ptr, w, h, w * 4, "RGBx", BENCH, fev[k], stream=st)
        assert rc == 0 and direct == 1 and L.mvfx_event_direct_queue(fev[k].h) == L.mvfx_direct_queue_of_stream(st)
        rc, _, direct = direct_detect(gpu, bufs[k].ptr, outs[k].ptr, w, h, "RGBx", "RGBA", DET, dev_[k], st, only=True)
        assert rc == 0 and direct == 1 and L.mvfx_event_direct_queue(dev_[k].h) == L.mvfx_event_direct_queue(fev[k].h)
    for k in range(n):
        gpu.check(L.mvfx_event_synchronize(dev_[k].h))
        assert L.mvfx_event_query(fev[k].h) == 1   # in order: the detector's completion implies the filter's
        assert np.array_equal(bufs[k].download().reshape(h, w * 4), want[k % 2][0]), k
        assert np.array_equal(outs[k].download().reshape(h, w * 4), want[k % 2][1]), k


def test_direct_only_refuses_instead_of_moving_a_frame_to_the_stream(gpu):
    """MVFX_OPT_DIRECT_ONLY: a frame the lane cannot take comes back with MVFX_ERR_DIRECT_UNAVAILABLE and untouched (a caller that relies on the lane's
    queue order must not be moved to a stream behind its back); an eligible frame is taken as ever"""
    L = gpu.lib()
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    for fmt, w, h, pad in (("RGBA", 64, 48, 32), ("RGB", 64, 48, 0), ("RGBA", 1021, 7, 0)):
        bpp = 3 if fmt == "RGB" else 4
        stride = w * bpp + pad
        f = frames.random_frame(0x5EED0F50 + w, w, h, bpp, stride)
        buf = gpu.DeviceBuffer(f.nbytes).upload(f)
        ev = Event(gpu)
        fr = gpu.make_frame(buf.ptr, w, h, stride, fmt)
        gpu.check(L.mvfx_thread_set_options(gpu.OPT_DIRECT_DISPATCH | gpu.OPT_DIRECT_ONLY))
        try:
            gpu.check(L.mvfx_thread_set_completion_event(ev.h))
            rc = L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr), ctypes.byref(gpu.HsvFilterSettings(*BENCH)), st)
            carried = L.mvfx_thread_clear_completion_event()
        finally:
            L.mvfx_thread_set_options(0)
        assert rc == gpu.ERR_DIRECT_UNAVAILABLE and carried == 0, (fmt, w, h, pad, rc)
        gpu.check(L.mvfx_stream_synchronize(st))
        assert np.array_equal(buf.download().reshape(h, stride), f)
    # settings outside the strength-reduced domain: refused too
    f = frames.random_frame(0x5EED0F60, 256, 64)
    buf = gpu.DeviceBuffer(f.nbytes).upload(f)
    fr = gpu.make_frame(buf.ptr, 256, 64, 1024, "RGBA")
    ev = Event(gpu)
    gpu.check(L.mvfx_thread_set_options(gpu.OPT_DIRECT_DISPATCH | gpu.OPT_DIRECT_ONLY))
    try:
        gpu.check(L.mvfx_thread_set_completion_event(ev.h))
        assert L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr), ctypes.byref(gpu.HsvFilterSettings(720.5, 1.0, 0.0, 1.0, 0.0)), st) == gpu.ERR_DIRECT_UNAVAILABLE
        L.mvfx_thread_clear_completion_event()
        # and without a completion event
        assert L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr), ctypes.byref(gpu.HsvFilterSettings(*BENCH)), st) == gpu.ERR_DIRECT_UNAVAILABLE
    finally:
        L.mvfx_thread_set_options(0)
    assert np.array_equal(buf.download().reshape(64, 1024), f)


def test_four_threads_share_the_lane(gpu):
    """the lane's queues are multi-producer: four threads, each with its own streams and fences, push 48 frames each through it at once"""
    import threading
    w, h, per = 1920, 1080, 48
    src = frames.random_frame(0x5EED0FA0, w, h)
    want = src.copy()
    assert orc.hsvfilter(want, w, w * 4, "RGBA", BENCH) == 0
    L = gpu.lib()
    bad = []

    def worker(t):
        try:
            gpu.check(L.mvfx_set_device(0))
            bufs = [gpu.DeviceBuffer(src.nbytes).upload(src) for _ in range(8)]
            evs = [Event(gpu) for _ in range(8)]
            streams = [ctypes.c_void_p(L.mvfx_thread_stream_n(k)) for k in range(2)]
            for i in range(per):
                k = i % 8
                if i >= 8:
                    gpu.check(L.mvfx_event_synchronize(evs[k].h))
                    if not np.array_equal(bufs[k].download().reshape(h, w * 4), want):
                        bad.append((t, i - 8))
                    bufs[k].upload(src)
                rc, carried, direct = direct_filter(gpu, bufs[k].ptr, w, h, w * 4, "RGBA", BENCH, evs[k], stream=streams[i & 1])
                if rc != 0 or direct != 1:
                    bad.append((t, i, rc, direct))
            for k in range(8):
                gpu.check(L.mvfx_event_synchronize(evs[k].h))
                if not np.array_equal(bufs[k].download().reshape(h, w * 4), want):
                    bad.append((t, "tail", k))
        except Exception as e:  # noqa: BLE001
            bad.append((t, repr(e)))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad, bad[:5]


def test_the_lane_can_be_switched_off_by_the_environment(gpu, tmp_path):
    """MVFX_DIRECT_DISPATCH=0: the option bit is accepted and ignored, the frame goes out on the stream, the event is an ordinary one"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "import numpy as np, _pkg\n"
        "from tests import frames\n"
        "from tests import oracle_binding as orc\n"
        "vfx = _pkg.vfx; L = vfx.lib(); vfx.check(L.mvfx_set_device(0))\n"
        "f = frames.random_frame(7, 256, 64); want = f.copy(); orc.hsvfilter(want, 256, 1024, 'RGBA', (90.0, 1.25, -0.05, 0.9, 0.02))\n"
        "buf = vfx.DeviceBuffer(f.nbytes).upload(f); ev = ctypes.c_void_p(); vfx.check(L.mvfx_event_create(ctypes.byref(ev)))\n"
        "fr = vfx.make_frame(buf.ptr, 256, 64, 1024, 'RGBA'); st = ctypes.c_void_p(L.mvfx_thread_stream())\n"
        "vfx.check(L.mvfx_thread_set_options(vfx.OPT_DIRECT_DISPATCH)); vfx.check(L.mvfx_thread_set_completion_event(ev))\n"
        "vfx.check(L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr), ctypes.byref(vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)), st))\n"
        "carried = L.mvfx_thread_clear_completion_event(); L.mvfx_thread_set_options(0)\n"
        "vfx.check(L.mvfx_event_synchronize(ev))\n"
        "print('direct', L.mvfx_event_is_direct(ev), 'carried', carried, 'equal', bool(np.array_equal(buf.download().reshape(64, 1024), want)))\n" % root)
    for env_val, expect in (("0", "direct 0 carried 1 equal True"), ("1", "direct 1 carried 1 equal True")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVFX_DIRECT_DISPATCH=env_val), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=300)
        assert r.returncode == 0 and expect in r.stdout, (env_val, r.stdout[-500:], r.stderr[-500:])


@pytest.mark.timeout(180)
@pytest.mark.parametrize("shape", [(128, 64, 2, 60000), (1920, 1080, 3, 8000)], ids=["2_threads_small_frames", "3_threads_1080p"])
def test_threads_with_few_fences_share_the_lane(gpu, shape):
    """Threads share the lane, each re-using THREE fences over thousands of one-frame calls (libmvfxbench.so: mvfxbench_lane_threads): the argument-slot
    ring wraps, every slot's last user is a signal that has been armed again since.  Round 6's first lane armed a dispatch's signal in front of the slot wait,
    and two streaming threads (a `queue` between two lane elements) then waited for each other's unsubmitted dispatches -- found by
    tools/soak_lane_chain.py; here as a test with a timeout: with the arming order of that lane BOTH shapes deadlock within seconds, every time
    (tools/check_lane_threads.py).  The bytes of lane dispatches are the business of the tests above."""
    import os
    bench = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gst-plugin-rs_amd", "libmvfxbench.so"))
    L = gpu.lib()
    w, h, threads, launches = shape
    per = 2
    identity = (0.0, 1.0, 0.0, 1.0, 0.0)
    host = [frames.random_frame(0x5EED1400 + k, w, h) for k in range(threads * per)]
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in host]
    fr = (gpu.Frame * (threads * per))(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
    s = gpu.HsvFilterSettings(*identity)
    took = ctypes.c_uint64()
    rc = bench.mvfxbench_lane_threads(0, threads, 3, launches, fr, per, ctypes.byref(s), ctypes.byref(took))
    assert rc == 0, gpu.last_error()
    if took.value == 0:
        pytest.skip("no lane on this box")
    assert took.value == threads * launches


def test_a_dependency_across_the_lane_queues_is_waited_for_on_the_device(gpu):
    """hsvfilter in place on lane queue 0, then -- no host wait -- hsvdetector reading that frame on lane queue 1 behind mvfx_direct_queue_wait_event(1, fence
    of the filter): a barrier packet with the filter's completion signal goes into queue 1, the detector's packet (barrier bit) runs behind it.  What the
    element layer does when a block's last dispatch sits in the other queue (round 6: before, the streaming thread waited itself)."""
    L = gpu.lib()
    w, h = 3840, 2160
    s0, s1 = ctypes.c_void_p(L.mvfx_thread_stream_n(0)), ctypes.c_void_p(L.mvfx_thread_stream_n(1))
    assert L.mvfx_direct_queue_of_stream(s0) == 0 and L.mvfx_direct_queue_of_stream(s1) == 1
    armed = 0
    for rep in range(10):
        f = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED1500 + rep)).reshape(h, w * 4)
        mid = f.copy()
        assert orc.hsvfilter(mid, w, w * 4, "RGBx", BENCH) == 0
        want = np.empty_like(mid)
        assert orc.hsvdetector(mid, w * 4, "RGBx", want, w * 4, "RGBA", w, DET) == 0
        a, b = gpu.DeviceBuffer(f.nbytes).upload(f), gpu.DeviceBuffer(f.nbytes)
        e1, e2 = Event(gpu), Event(gpu)
        rc, _, direct = direct_filter(gpu, a.ptr, w, h, w * 4, "RGBx", BENCH, e1, stream=s0)
        assert rc == 0 and direct == 1 and L.mvfx_event_direct_queue(e1.h) == 0
        got = L.mvfx_direct_queue_wait_event(1, e1.h)
        assert got in (0, 1), gpu.last_error()   # 0: the 11 us kernel had finished before we asked
        armed += got
        rc, _, direct = direct_detect(gpu, a.ptr, b.ptr, w, h, "RGBx", "RGBA", DET, e2, s1, only=True)
        assert rc == 0 and direct == 1 and L.mvfx_event_direct_queue(e2.h) == 1
        gpu.check(L.mvfx_event_synchronize(e2.h))
        assert L.mvfx_event_query(e1.h) == 1
        assert np.array_equal(b.download().reshape(h, w * 4), want), rep
    assert armed >= 1, "never saw the filter still running: the device-side wait was not exercised"
    # an ordinary event, a fired fence, the same queue
    ev = Event(gpu)
    gpu.check(L.mvfx_event_record(ev.h, s0))
    assert L.mvfx_direct_queue_wait_event(1, ev.h) == 0
    gpu.check(L.mvfx_event_synchronize(ev.h))
    assert L.mvfx_direct_queue_wait_event(0, e1.h) == 0


def test_the_lane_can_be_parked_and_comes_back(gpu):
    """mvfx_direct_lane_park: the lane's two hardware queues are drained and destroyed (hardware queues are few: idle ones beside HIP's slow busy HIP streams
    down, csrc/direct_dispatch.h "PARKING"); dispatches in flight finish first, the next dispatch makes the queues again -- same bytes throughout"""
    L = gpu.lib()
    w, h = 3840, 2160
    f = frames.random_frame(0x5EED1600, w, h)
    want = f.copy()
    assert orc.hsvfilter(want, w, w * 4, "RGBA", BENCH) == 0
    assert L.mvfx_direct_lane_park() in (0, 1)
    for rep in range(3):
        bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for _ in range(6)]
        evs = [Event(gpu) for _ in range(6)]
        sts = [ctypes.c_void_p(L.mvfx_thread_stream_n(k & 1)) for k in range(6)]
        for k in range(6):
            rc, _, direct = direct_filter(gpu, bufs[k].ptr, w, h, w * 4, "RGBA", BENCH, evs[k], stream=sts[k])
            assert rc == 0 and direct == 1, (rep, k, rc, direct, gpu.last_error())
        assert L.mvfx_direct_lane_park() == 1           # six 4K frames in flight: drained first
        assert all(L.mvfx_event_query(e.h) == 1 for e in evs)
        assert L.mvfx_direct_lane_park() == 0           # parked already
        for k in range(6):
            assert np.array_equal(bufs[k].download().reshape(h, w * 4), want), (rep, k)
