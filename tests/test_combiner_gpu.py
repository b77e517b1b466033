"""The launch combiner (gst-plugin-rs_amd/csrc/combiner.hip): one call per buffer stays the element's contract
(hsvfilter/imp.rs:322-326), frames that the streaming threads of a process submit at about the same time share one batched launch
with per-frame settings.  Checked: the bytes equal the oracle's for every stream's own settings (both signs of hue-shift, settings
outside the strength-reduced kernel's domain, a stream with another frame size), calls are ordered with the caller's stream, and
launches really are shared."""
import ctypes
import threading

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu

SETTINGS = [(90.0, 1.25, -0.05, 0.9, 0.02), (-45.0, 0.8, 0.1, 1.1, -0.03), (0.0, 1.0, 0.0, 1.0, 0.0), (359.5, 2.0, -0.5, 0.5, 0.25),
            (-360.0, 1.0, 0.0, 1.0, 0.0), (500.0, 1.0, 0.0, 1.0, 0.0), (12.5, 0.0, 0.5, 1.0, 0.0), (-0.0, 1.5, 0.0, 0.75, 0.1)]


def test_frames_with_their_own_settings_in_one_call(gpu):
    """mvfx_hsvfilter_transform_frames_ip_settings: settings[i] for frames[i]; frames of one sign of hue-shift share a launch, settings
    outside the proven domain (|shift| > 360) take the literal kernel -- same bytes as the oracle frame by frame."""
    w, h, n = 640, 360, len(SETTINGS)
    host = [frames.random_frame(0xC0B0 + k, w, h) for k in range(n)]
    bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in host]
    arr = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "BGRx") for b in bufs])
    st = (gpu.HsvFilterSettings * n)(*[gpu.HsvFilterSettings(*s) for s in SETTINGS])
    gpu.check(gpu.lib().mvfx_hsvfilter_transform_frames_ip_settings(arr, n, st, None))
    gpu.check(gpu.lib().mvfx_stream_synchronize(None))
    for k in range(n):
        exp = host[k].copy()
        assert orc.hsvfilter(exp, w, w * 4, "BGRx", SETTINGS[k]) == 0
        assert np.array_equal(bufs[k].download().reshape(h, w * 4), exp), f"frame {k} settings {SETTINGS[k]}"


@pytest.mark.parametrize("fenced", [False, True], ids=["stream-ordered", "fenced"])
def test_sixteen_threads_through_the_combiner(gpu, fenced):
    """16 host threads (GStreamer: one streaming thread per stream), each with its own HIP stream, frames and settings, each making
    single-frame calls through the combiner; thread 15 works on another frame size (never shares a launch).  Every frame equals the
    oracle's answer for its stream's settings; the combiner needed fewer launches than frames."""
    L = gpu.lib()
    n_threads, per_thread = 16, 12
    w, h = 960, 540
    nb0, nf0 = ctypes.c_uint64(), ctypes.c_uint64()
    gpu.check(L.mvfx_combiner_stats(0, ctypes.byref(nb0), ctypes.byref(nf0)))
    errors, results = [], {}
    start = threading.Barrier(n_threads)

    def body(t):
        try:
            gpu.check(L.mvfx_set_device(0))
            tw, th = (w, h) if t != 15 else (320, 200)
            s = SETTINGS[t % len(SETTINGS)]
            st = gpu.HsvFilterSettings(*s)
            stream = L.mvfx_thread_stream()
            host = [frames.random_frame(0xC100 + 97 * t + k, tw, th) for k in range(per_thread)]
            bufs = [gpu.DeviceBuffer(f.nbytes) for f in host]
            ev_in = ctypes.c_void_p()
            gpu.check(L.mvfx_event_create(ctypes.byref(ev_in)))
            start.wait()
            for k in range(per_thread):
                # upload on the caller's stream, filter through the combiner, download on the caller's stream: the combined call must
                # order behind the copy before it and ahead of the copy after it
                gpu.check(L.mvfx_copy_to_device_async(ctypes.c_void_p(bufs[k].ptr), host[k].ctypes.data_as(ctypes.c_void_p), host[k].nbytes,
                                                      ctypes.c_void_p(stream)))
                f = gpu.make_frame(bufs[k].ptr, tw, th, tw * 4, "RGBA")
                if fenced:
                    # the frame's fence in (the upload's event), the batch's event out; this thread's stream then waits for it
                    gpu.check(L.mvfx_event_record(ev_in, ctypes.c_void_p(stream)))
                    done = ctypes.c_void_p()
                    gpu.check(L.mvfx_hsvfilter_transform_frame_ip_fenced(ctypes.byref(f), ctypes.byref(st), ev_in, ctypes.byref(done)))
                    assert done.value
                    gpu.check(L.mvfx_stream_wait_event(ctypes.c_void_p(stream), done))
                else:
                    gpu.check(L.mvfx_hsvfilter_transform_frame_ip_combined(ctypes.byref(f), ctypes.byref(st), ctypes.c_void_p(stream)))
            out = [np.empty_like(f) for f in host]
            for k in range(per_thread):
                gpu.check(L.mvfx_copy_to_host_async(out[k].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(bufs[k].ptr), out[k].nbytes,
                                                    ctypes.c_void_p(stream)))
            gpu.check(L.mvfx_stream_synchronize(ctypes.c_void_p(stream)))
            results[t] = (host, out, s, tw)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
    for th_ in ths:
        th_.start()
    for th_ in ths:
        th_.join()
    assert errors == []
    for t, (host, out, s, tw) in results.items():
        for k in range(per_thread):
            exp = host[k].copy()
            assert orc.hsvfilter(exp, tw, tw * 4, "RGBA", s) == 0
            assert np.array_equal(out[k], exp), f"thread {t} frame {k}"
    nb1, nf1 = ctypes.c_uint64(), ctypes.c_uint64()
    gpu.check(L.mvfx_combiner_stats(0, ctypes.byref(nb1), ctypes.byref(nf1)))
    assert nf1.value - nf0.value == n_threads * per_thread
    assert nb1.value - nb0.value < nf1.value - nf0.value, "no launch was shared"


LONE = r"""
import ctypes, sys, time
sys.path.insert(0, {root!r})
import _pkg
from tests import frames
gpu = _pkg.vfx
L = gpu.lib()
gpu.check(L.mvfx_set_device(0))
w, h = 320, 240
f0 = frames.random_frame(0xC200, w, h)
buf = gpu.DeviceBuffer(f0.nbytes).upload(f0)
f = gpu.make_frame(buf.ptr, w, h, w * 4, "RGBA")
st = gpu.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
for _ in range(5):
    gpu.check(L.mvfx_hsvfilter_transform_frame_ip_combined(ctypes.byref(f), ctypes.byref(st), None))
gpu.check(L.mvfx_stream_synchronize(None))
t = time.perf_counter()
for _ in range(50):
    gpu.check(L.mvfx_hsvfilter_transform_frame_ip_combined(ctypes.byref(f), ctypes.byref(st), None))
gpu.check(L.mvfx_stream_synchronize(None))
print("PER_CALL_US", (time.perf_counter() - t) / 50 * 1e6)
"""


def test_a_lone_stream_is_not_held_back(gpu):
    """One caller is the only recently active stream: every call is launched at once, however long the collection window is
    (here 20 ms: 50 calls would take a second if each waited for it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVFX_COMBINE_WINDOW_US="20000")
    r = subprocess.run([sys.executable, "-c", LONE.format(root=root)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    us = float([ln for ln in r.stdout.splitlines() if ln.startswith("PER_CALL_US")][-1].split()[1])
    assert us < 2000.0, f"{us:.0f} us per call: the lone stream waited for the window"


def test_threads_that_come_and_go_recycle_their_streams(gpu):
    """Sixteen threads per generation, each rotating single-frame hsvfilter launches over its four private streams
    (mvfx_thread_stream_n) and exiting with work still queued -- GStreamer streaming threads at end-of-stream.  The streams of an
    exiting thread go back to a per-device pool (destroying them concurrently crashed inside the HIP runtime: capi_common.hip,
    IdleStreams): later generations get the same handles again, and the bytes are the oracle's."""
    L = gpu.lib()
    L.mvfx_thread_stream_n.restype = ctypes.c_void_p
    L.mvfx_thread_stream_n.argtypes = [ctypes.c_uint32]
    n_threads, generations, w, h = 16, 6, 640, 360
    s = SETTINGS[0]
    host = frames.random_frame(0xC200, w, h)
    exp = host.copy()
    assert orc.hsvfilter(exp, w, w * 4, "RGBA", s) == 0
    seen, errors = [], []
    for g in range(generations):
        handles, bufs = {}, {}
        start = threading.Barrier(n_threads)

        def body(t):
            try:
                gpu.check(L.mvfx_set_device(0))
                st = gpu.HsvFilterSettings(*s)
                streams = [L.mvfx_thread_stream_n(i) for i in range(4)]
                assert all(streams) and len(set(streams)) == 4
                handles[t] = streams
                mine = [gpu.DeviceBuffer(host.nbytes).upload(host) for _ in range(4)]
                start.wait()
                for k in range(4):
                    f = gpu.make_frame(mine[k].ptr, w, h, w * 4, "RGBA")
                    gpu.check(L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(st), ctypes.c_void_p(streams[k])))
                bufs[t] = mine  # the thread exits with its launches still in flight
            except Exception as e:  # noqa: BLE001
                errors.append((g, t, repr(e)))

        ths = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
        for th_ in ths:
            th_.start()
        for th_ in ths:
            th_.join()
        assert errors == []
        for t in range(n_threads):
            for k in range(4):
                gpu.check(L.mvfx_stream_synchronize(ctypes.c_void_p(handles[t][k])))
                assert np.array_equal(bufs[t][k].download().reshape(h, w * 4), exp), f"generation {g} thread {t} stream {k}"
        seen.append({x for v in handles.values() for x in v})
    assert all(len(x) == n_threads * 4 for x in seen)
    # generation 0 may draw on streams that threads of earlier tests left in the pool; from then on nothing new is created
    # (without the pool: 6 x 64 distinct handles)
    assert len(set().union(*seen)) <= 2 * n_threads * 4, "later generations of threads did not get pooled streams back"


def test_event_query_never_blocks_and_turns_one(gpu):
    """mvfx_event_query (hipEventQuery behind the C ABI): 1 on an event nothing was recorded on and once the work before the record has
    finished, 0 (not an error, mvfx_last_error untouched by the next call) while it is still running, MVFX_ERR_INVALID_ARGUMENT on
    NULL.  The elements ask it of their input block before they hold a kernel back (host/gst/mvfx_pair_hold.h)."""
    L = gpu.lib()
    w, h, n = 3840, 2160, 24
    ev = ctypes.c_void_p()
    gpu.check(L.mvfx_event_create(ctypes.byref(ev)))
    assert L.mvfx_event_query(ev) == 1
    assert L.mvfx_event_query(None) < 0
    host = frames.random_frame(7, w, h)
    bufs = [gpu.DeviceBuffer(host.nbytes).upload(host) for _ in range(n)]
    gpu.check(L.mvfx_stream_synchronize(None))
    arr = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
    s = gpu.HsvFilterSettings(90.0, 1.0, 0.0, 1.0, 0.0)
    seen = set()
    for _ in range(8):   # ~0.3 ms of device work per round, queried right behind the record
        gpu.check(L.mvfx_hsvfilter_transform_frames_ip(arr, n, ctypes.byref(s), None))
        gpu.check(L.mvfx_event_record(ev, None))
        seen.add(L.mvfx_event_query(ev))
        gpu.check(L.mvfx_stream_synchronize(None))
        assert L.mvfx_event_query(ev) == 1
    assert seen <= {0, 1} and 0 in seen
    gpu.check(L.mvfx_event_destroy(ev))


def test_completion_event_rides_on_the_kernels_of_the_next_call(gpu):
    """mvfx_thread_set_completion_event / _clear_completion_event: while an event is set every kernel the thread launches through the
    library carries it as its stop event (hipExtLaunchKernelGGL, csrc/mvfx_internal.h MVFX_LAUNCH); _clear says how many did.  The event
    then behaves like a recorded one: not reached right behind a long launch, reached after it, and a stream that waits for it sees the
    kernel's bytes.  Nothing launched -> 0: the caller records it itself."""
    L = gpu.lib()
    w, h, n = 3840, 2160, 16
    ev = ctypes.c_void_p()
    gpu.check(L.mvfx_event_create(ctypes.byref(ev)))
    gpu.check(L.mvfx_thread_set_completion_event(ev))
    assert L.mvfx_thread_clear_completion_event() == 0
    host = frames.random_frame(11, w, h)
    bufs = [gpu.DeviceBuffer(host.nbytes).upload(host) for _ in range(n)]
    gpu.check(L.mvfx_stream_synchronize(None))
    arr = (gpu.Frame * n)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
    s = gpu.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    seen = set()
    for _ in range(6):
        gpu.check(L.mvfx_thread_set_completion_event(ev))
        gpu.check(L.mvfx_hsvfilter_transform_frames_ip(arr, n, ctypes.byref(s), None))
        assert L.mvfx_thread_clear_completion_event() >= 1
        seen.add(L.mvfx_event_query(ev))
        gpu.check(L.mvfx_event_synchronize(ev))
        assert L.mvfx_event_query(ev) == 1
    assert 0 in seen and seen <= {0, 1}
    # launches behind a cleared setting carry nothing: the event stays reached
    gpu.check(L.mvfx_hsvfilter_transform_frames_ip(arr, n, ctypes.byref(s), None))
    assert L.mvfx_event_query(ev) == 1
    gpu.check(L.mvfx_stream_synchronize(None))
    exp = host.copy()
    for _ in range(7):
        assert orc.hsvfilter(exp, w, w * 4, "RGBA", (90.0, 1.25, -0.05, 0.9, 0.02)) == 0
    assert np.array_equal(bufs[3].download().reshape(h, w * 4), exp)
    gpu.check(L.mvfx_event_destroy(ev))
