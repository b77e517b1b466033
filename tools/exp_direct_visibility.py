#!/usr/bin/env python3
"""tools/exp_direct_visibility.py -- round 6 diagnostic: one exhaustive frame through the direct lane, then read back (a) at once, (b) after 5 ms:
how many bytes differ from the oracle, and are the wrong ones the INPUT's (stores not yet in memory when the completion signal fired)?"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import _pkg
    from tests import frames
    from tests import oracle_binding as orc
    vfx = _pkg.vfx
    L = vfx.lib()
    vfx.check(L.mvfx_set_device(0))
    ex = frames.exhaustive_rgbx()
    st = (90.0, 1.25, -0.05, 0.9, 0.02)
    want = ex.copy()
    orc.hsvfilter(want, 4096, 4096 * 4, "RGBA", st)
    ev = ctypes.c_void_p()
    vfx.check(L.mvfx_event_create(ctypes.byref(ev)))
    for delay in (0.0, 0.005):
        for rep in range(3):
            buf = vfx.DeviceBuffer(ex.nbytes).upload(ex)
            f = vfx.make_frame(buf.ptr, 4096, 4096, 4096 * 4, "RGBA")
            s = vfx.HsvFilterSettings(*st)
            vfx.check(L.mvfx_thread_set_options(vfx.OPT_DIRECT_DISPATCH))
            vfx.check(L.mvfx_thread_set_completion_event(ev))
            vfx.check(L.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(s), None))
            L.mvfx_thread_clear_completion_event()
            L.mvfx_thread_set_options(0)
            direct = L.mvfx_event_is_direct(ev)
            vfx.check(L.mvfx_event_synchronize(ev))
            if delay:
                time.sleep(delay)
            got = buf.download().reshape(want.shape)
            bad = got != want
            nb = int(np.count_nonzero(bad))
            still_input = int(np.count_nonzero(bad & (got == ex)))
            rows = np.nonzero(bad.any(axis=1))[0]
            print(f"release={os.environ.get('MVFX_DIRECT_RELEASE', '0')} delay {delay * 1e3:.0f} ms direct={direct}: {nb} bytes differ, {still_input} of them still hold the input; "
                  f"rows {rows[:4].tolist()}..{rows[-4:].tolist() if len(rows) else []} ({len(rows)} rows)", flush=True)


if __name__ == "__main__":
    main()
