"""colorlut through the direct-dispatch lane (round 6; gst-plugin-rs_amd/csrc/direct_dispatch_colorlut.h, csrc/direct/colorlut_direct_kernels.hip): one RGBA8
frame through a 3-D LUT as a packet of the library's own, with either x-prelerped window kernel and write-through stores.  The kernels are new
instantiations of the window bodies => their own proofs: all 2^24 colours against the oracle through both, pictures at sizes that are not multiples
of the blocks, row padding kept; and what the lane does not take."""
import ctypes

import numpy as np
import pytest

from tests import cubes, frames
from tests import oracle_binding as orc
from tests.test_direct_dispatch_gpu import Event

pytestmark = pytest.mark.gpu


def direct_lut(vfx, dev, src_ptr, src_stride, dst_ptr, dst_stride, w, h, ev, stream, fmt="RGBA", wg_window=None, only=False, placement=0, unordered=False):
    """one colorlut call with the lane allowed (wg_window None: the content probe chooses; True / False: forced workgroup window / per-wave windows;
    unordered: MVFX_OPT_DIRECT_UNORDERED, the packet without the barrier bit); returns (status, launches that carried the event the HIP way, is_direct)"""
    L = vfx.lib()
    fi, fo = vfx.make_frame(src_ptr, w, h, src_stride, fmt), vfx.make_frame(dst_ptr, w, h, dst_stride, fmt)
    word = vfx.OPT_DIRECT_DISPATCH | (vfx.OPT_DIRECT_ONLY if only else 0) | (vfx.OPT_DIRECT_UNORDERED if unordered else 0)
    if wg_window is True:
        word |= vfx.OPT_LUT_WG_WINDOW
    elif wg_window is False:
        placement = 7
    word |= vfx.options(placement=placement).word
    vfx.check(L.mvfx_thread_set_options(word))
    try:
        vfx.check(L.mvfx_thread_set_completion_event(ev.h))
        rc = L.mvfx_colorlut_transform_frame(dev.h, ctypes.byref(fi), ctypes.byref(fo), stream)
        carried = L.mvfx_thread_clear_completion_event()
    finally:
        L.mvfx_thread_set_options(0)
    return rc, carried, L.mvfx_event_is_direct(ev.h)


def _domain33():
    return cubes.analytic_3d(33).replace("DOMAIN_MIN 0.0 0.0 0.0", "DOMAIN_MIN -0.25 0.0 0.1").replace("DOMAIN_MAX 1.0 1.0 1.0", "DOMAIN_MAX 1.5 1.0 0.9")


CUBES = {"analytic33": lambda: cubes.analytic_3d(33), "analytic65": lambda: cubes.analytic_3d(65), "analytic9": lambda: cubes.analytic_3d(9),
         "analytic5": lambda: cubes.analytic_3d(5), "domain33": _domain33, "identity17": lambda: cubes.identity_3d(17)}


@pytest.fixture(scope="module")
def luts(gpu):
    out = {}
    for name, make in CUBES.items():
        text = make()
        o = orc.CubeLut(text)
        assert o.ok, o.error
        out[name] = (gpu.CubeLut(text), o)
    return out


@pytest.mark.parametrize("name", sorted(CUBES))
@pytest.mark.parametrize("wg_window", [True, False], ids=["workgroup_window", "wave_windows"])
@pytest.mark.parametrize("unordered", [False, True], ids=["in_order", "unordered"])
def test_lane_colorlut_on_all_2_24_colours(gpu, luts, name, wg_window, unordered):
    """every RGB triple (byte 3 = a hash of the pixel index, which must come through untouched) through both lane kernels; the exhaustive frame is the
    worst case for the windows (its colours scatter): the miss paths and the workgroup window's far path run too"""
    dev, o = luts[name]
    L = gpu.lib()
    ex = frames.exhaustive_rgbx()
    exp = np.empty_like(ex)
    assert o.apply(ex, 4096 * 4, exp, 4096 * 4, 4096, 4096, "RGBA") == 0
    src = gpu.DeviceBuffer(ex.nbytes).upload(ex)
    dst = gpu.DeviceBuffer(ex.nbytes)
    ev = Event(gpu)
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    rc, carried, direct = direct_lut(gpu, dev, src.ptr, 4096 * 4, dst.ptr, 4096 * 4, 4096, 4096, ev, st, wg_window=wg_window, unordered=unordered)
    assert rc == 0, gpu.last_error()
    assert carried == 1 and direct == 1, (carried, direct)
    gpu.check(L.mvfx_event_synchronize(ev.h))  # the fence, not the stream: nothing was enqueued on it
    assert np.array_equal(dst.download().reshape(ex.shape), exp)


@pytest.mark.parametrize("wg_window", [True, False, None], ids=["workgroup_window", "wave_windows", "probe"])
def test_lane_colorlut_pictures_partial_blocks_and_row_padding(gpu, luts, wg_window):
    """natural-like pictures with and without noise, flat bars, uniform-random colours at sizes that are not multiples of the wave block (64 x 20) or of
    the workgroup's (128 x 40), with row padding on either side (kept); the automatic choice (content probe on the caller's stream) too"""
    dev, o = luts["analytic33"]
    L = gpu.lib()
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    for k, (w, h, ipad, opad) in enumerate(((3840, 2160, 0, 0), (1000, 250, 0, 0), (68, 20, 16, 0), (132, 44, 0, 32), (1920, 1080, 64, 64), (4, 1, 0, 0), (128, 40, 0, 0))):
        kind = k % 4
        if kind == 0:
            f = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED0D00 + k)).reshape(h, w, 4)
        elif kind == 1:
            f = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED0D00 + k)).reshape(h, w, 4).copy()
            nz = np.random.default_rng(0x5EED0D20 + k).integers(-12, 13, (h, w, 3))
            f[..., :3] = np.clip(f[..., :3].astype(np.int32) + nz, 0, 255).astype(np.uint8)
        elif kind == 2:
            f = np.ascontiguousarray(frames.smpte_like(w, h)).reshape(h, w, 4)
        else:
            f = np.ascontiguousarray(frames.random_frame(0x5EED0D10 + k, w, h)).reshape(h, w, 4)
        istride, ostride = w * 4 + ipad, w * 4 + opad
        src = np.full((h, istride), 0x3C, np.uint8)
        src[:, :w * 4] = f.reshape(h, w * 4)
        fill = np.full((h, ostride), 0xA5, np.uint8)
        exp = fill.copy()
        assert o.apply(src, istride, exp, ostride, w, h, "RGBA") == 0
        din = gpu.DeviceBuffer(src.nbytes).upload(src)
        dout = gpu.DeviceBuffer(fill.nbytes).upload(fill)
        ev = Event(gpu)
        for _ in range(3 if wg_window is None else 1):  # the probe's verdict of call n arrives for call n + 1
            rc, carried, direct = direct_lut(gpu, dev, din.ptr, istride, dout.ptr, ostride, w, h, ev, st, wg_window=wg_window)
            assert rc == 0, gpu.last_error()
            assert carried == 1 and direct == 1, (w, h, carried, direct)
            gpu.check(L.mvfx_event_synchronize(ev.h))
            got = dout.download().reshape(h, ostride)
            assert np.array_equal(got, exp), (w, h, ipad, opad, int(np.count_nonzero(got != exp)))
            gpu.check(L.mvfx_stream_synchronize(st))


def test_what_the_colorlut_lane_does_not_take(gpu, luts):
    """1-D LUTs, RGBA64, widths that are not multiples of four, unaligned rows, batches, the other placements: launched on `stream` as ever (the event an
    ordinary stop event, same bytes) -- or, with MVFX_OPT_DIRECT_ONLY, refused with nothing done"""
    L = gpu.lib()
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    d33, o33 = luts["analytic33"]
    t1 = cubes.curve_1d(256)
    o1 = orc.CubeLut(t1)
    d1 = gpu.CubeLut(t1)
    cases = [("1-D LUT", d1, o1, "RGBA", 256, 64, 0, 0), ("RGBA64", d33, o33, "RGBA64_LE", 256, 64, 0, 0), ("width % 4", d33, o33, "RGBA", 1021, 7, 0, 0),
             ("rows not 16-byte aligned", d33, o33, "RGBA", 64, 48, 8, 0), ("baked table", d33, o33, "RGBA", 256, 64, 0, 6), ("cell window", d33, o33, "RGBA", 256, 64, 0, 5),
             ("literal", d33, o33, "RGBA", 256, 64, 0, 4)]
    for what, dev, o, fmt, w, h, pad, placement in cases:
        bpp = 8 if fmt != "RGBA" else 4
        stride = w * bpp + pad
        src = frames.random_frame(0x5EED0E00 + w + placement, w, h, bpp, stride)
        exp = np.full((h, stride), 0x77, np.uint8)
        assert o.apply(src, stride, exp, stride, w, h, fmt) == 0
        din = gpu.DeviceBuffer(src.nbytes).upload(src)
        fill = np.full((h, stride), 0x77, np.uint8)
        dout = gpu.DeviceBuffer(fill.nbytes).upload(fill)
        ev = Event(gpu)
        rc, carried, direct = direct_lut(gpu, dev, din.ptr, stride, dout.ptr, stride, w, h, ev, st, fmt=fmt, only=True, placement=placement)
        assert rc == gpu.ERR_DIRECT_UNAVAILABLE and carried == 0 and direct == 0, (what, rc, carried, direct)
        gpu.check(L.mvfx_stream_synchronize(st))
        assert np.array_equal(dout.download().reshape(h, stride), fill), what
        rc, carried, direct = direct_lut(gpu, dev, din.ptr, stride, dout.ptr, stride, w, h, ev, st, fmt=fmt, placement=placement)
        assert rc == 0, (what, gpu.last_error())
        assert carried >= 1 and direct == 0, (what, carried, direct)
        gpu.check(L.mvfx_event_synchronize(ev.h))
        assert np.array_equal(dout.download().reshape(h, stride), exp), what
    # a batch of two frames with the bit set: one launch on the stream
    w, h = 256, 64
    srcs = [frames.random_frame(0x5EED0E40 + k, w, h) for k in range(2)]
    din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in srcs]
    dout = [gpu.DeviceBuffer(f.nbytes) for f in srcs]
    fi = (gpu.Frame * 2)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in din])
    fo = (gpu.Frame * 2)(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in dout])
    ev = Event(gpu)
    gpu.check(L.mvfx_thread_set_options(gpu.OPT_DIRECT_DISPATCH))
    try:
        gpu.check(L.mvfx_thread_set_completion_event(ev.h))
        gpu.check(L.mvfx_colorlut_transform_frames(d33.h, fi, fo, 2, st))
        assert L.mvfx_thread_clear_completion_event() == 1 and L.mvfx_event_is_direct(ev.h) == 0
        # and without a completion event: the stream
        fi1, fo1 = gpu.make_frame(din[0].ptr, w, h, w * 4, "RGBA"), gpu.make_frame(dout[0].ptr, w, h, w * 4, "RGBA")
        gpu.check(L.mvfx_colorlut_transform_frame(d33.h, ctypes.byref(fi1), ctypes.byref(fo1), st))
    finally:
        L.mvfx_thread_set_options(0)
    gpu.check(L.mvfx_stream_synchronize(st))
    for k in range(2):
        exp = np.empty_like(srcs[k])
        assert o33.apply(srcs[k], w * 4, exp, w * 4, w, h, "RGBA") == 0
        assert np.array_equal(dout[k].download().reshape(h, w * 4), exp)


def test_filter_then_colorlut_on_one_lane_queue(gpu, luts):
    """hsvfilter in place, then colorlut reading that frame, both as lane packets on the SAME queue: in order, no wait in between -- the colorlut packet's
    acquire sees what the filter's write-through stores left in memory"""
    from tests.test_direct_dispatch_gpu import BENCH, direct_filter
    dev, o = luts["analytic33"]
    L = gpu.lib()
    w, h = 3840, 2160
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    for rep in range(8):
        f = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED0E80 + rep)).reshape(h, w * 4)
        mid = f.copy()
        assert orc.hsvfilter(mid, w, w * 4, "RGBA", BENCH) == 0
        exp = np.empty_like(mid)
        assert o.apply(mid, w * 4, exp, w * 4, w, h, "RGBA") == 0
        a = gpu.DeviceBuffer(f.nbytes).upload(f)
        b = gpu.DeviceBuffer(f.nbytes)
        e1, e2 = Event(gpu), Event(gpu)
        rc, _, direct = direct_filter(gpu, a.ptr, w, h, w * 4, "RGBA", BENCH, e1, stream=st)
        assert rc == 0 and direct == 1
        rc, _, direct = direct_lut(gpu, dev, a.ptr, w * 4, b.ptr, w * 4, w, h, e2, st, wg_window=bool(rep & 1), only=True)
        assert rc == 0 and direct == 1
        assert L.mvfx_event_direct_queue(e1.h) == L.mvfx_event_direct_queue(e2.h)
        gpu.check(L.mvfx_event_synchronize(e2.h))
        assert L.mvfx_event_query(e1.h) == 1  # in order: the filter finished first
        assert np.array_equal(b.download().reshape(h, w * 4), exp), rep


@pytest.mark.parametrize("wg_window", [True, False], ids=["workgroup_window", "wave_windows"])
def test_unordered_frames_in_flight_each_with_its_own_fence(gpu, luts, wg_window):
    """MVFX_OPT_DIRECT_UNORDERED: forty independent frame pairs sent as fast as the thread can, alternating between the lane's queues, no barrier bit on
    any packet -- several kernels in flight on a queue at once; every frame's fence fires when ITS kernel is done, every frame has the oracle's bytes"""
    dev, o = luts["analytic33"]
    L = gpu.lib()
    w, h, n = 1920, 1080, 40
    sts = [ctypes.c_void_p(L.mvfx_thread_stream_n(0)), ctypes.c_void_p(L.mvfx_thread_stream_n(1))]
    srcs = [np.ascontiguousarray(frames.natural_like(w, h, 0x5EED1000 + k)).reshape(h, w * 4) for k in range(4)]
    exps = []
    for f in srcs:
        e = np.empty_like(f)
        assert o.apply(f, w * 4, e, w * 4, w, h, "RGBA") == 0
        exps.append(e)
    din = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in srcs]
    dout = [gpu.DeviceBuffer(srcs[0].nbytes) for _ in range(n)]
    evs = [Event(gpu) for _ in range(n)]
    for k in range(n):
        rc, carried, direct = direct_lut(gpu, dev, din[k % 4].ptr, w * 4, dout[k].ptr, w * 4, w, h, evs[k], sts[k & 1], wg_window=wg_window, only=True, unordered=True)
        assert rc == 0 and carried == 1 and direct == 1, (k, rc, carried, direct, gpu.last_error())
    for k in reversed(range(n)):  # (any order: each fence stands for its own frame)
        gpu.check(L.mvfx_event_synchronize(evs[k].h))
        assert np.array_equal(dout[k].download().reshape(h, w * 4), exps[k % 4]), k


def test_an_in_order_packet_behind_unordered_ones_waits_for_them(gpu, luts):
    """unordered colorlut A -> B, then -- no host wait -- an IN-ORDER colorlut B -> C on the same lane queue: the second packet carries the barrier bit and
    starts when everything in front of it has finished, the unordered packet included; C = LUT(LUT(A))"""
    dev, o = luts["analytic33"]
    L = gpu.lib()
    w, h = 3840, 2160
    st = ctypes.c_void_p(L.mvfx_thread_stream())
    for rep in range(6):
        a = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED1100 + rep)).reshape(h, w * 4)
        b = np.empty_like(a)
        c = np.empty_like(a)
        assert o.apply(a, w * 4, b, w * 4, w, h, "RGBA") == 0
        assert o.apply(b, w * 4, c, w * 4, w, h, "RGBA") == 0
        da = gpu.DeviceBuffer(a.nbytes).upload(a)
        db = gpu.DeviceBuffer(a.nbytes)
        dc = gpu.DeviceBuffer(a.nbytes)
        e1, e2 = Event(gpu), Event(gpu)
        rc, _, direct = direct_lut(gpu, dev, da.ptr, w * 4, db.ptr, w * 4, w, h, e1, st, wg_window=bool(rep & 1), only=True, unordered=True)
        assert rc == 0 and direct == 1
        rc, _, direct = direct_lut(gpu, dev, db.ptr, w * 4, dc.ptr, w * 4, w, h, e2, st, wg_window=bool(rep & 2), only=True)
        assert rc == 0 and direct == 1
        gpu.check(L.mvfx_event_synchronize(e2.h))
        assert L.mvfx_event_query(e1.h) == 1
        assert np.array_equal(dc.download().reshape(h, w * 4), c), rep
        assert np.array_equal(db.download().reshape(h, w * 4), b), rep


def test_freeing_the_lut_waits_for_the_lane(gpu):
    """mvfx_cube_lut_free with lane kernels of the LUT still in flight: hipFree waits for HIP streams only, so the free quiesces the lane's queues first
    (direct_quiesce) -- every fence has fired when it returns, and the frames hold the oracle's bytes (the tables were not pulled from under a kernel)"""
    text = cubes.analytic_3d(33)
    o = orc.CubeLut(text)
    dev = gpu.CubeLut(text)
    L = gpu.lib()
    w, h, n = 3840, 2160, 24
    sts = [ctypes.c_void_p(L.mvfx_thread_stream_n(0)), ctypes.c_void_p(L.mvfx_thread_stream_n(1))]
    src = np.ascontiguousarray(frames.natural_like(w, h, 0x5EED1300)).reshape(h, w * 4)
    exp = np.empty_like(src)
    assert o.apply(src, w * 4, exp, w * 4, w, h, "RGBA") == 0
    din = gpu.DeviceBuffer(src.nbytes).upload(src)
    dout = [gpu.DeviceBuffer(src.nbytes) for _ in range(n)]
    evs = [Event(gpu) for _ in range(n)]
    for k in range(n):
        rc, _, direct = direct_lut(gpu, dev, din.ptr, w * 4, dout[k].ptr, w * 4, w, h, evs[k], sts[k & 1], wg_window=bool(k & 2), only=True, unordered=bool(k & 1))
        assert rc == 0 and direct == 1, (k, rc, direct)
    dev.free()
    assert all(L.mvfx_event_query(e.h) == 1 for e in evs)
    for k in (0, 1, n - 2, n - 1):
        assert np.array_equal(dout[k].download().reshape(h, w * 4), exp), k
