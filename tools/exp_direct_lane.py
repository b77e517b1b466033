#!/usr/bin/env python3
"""tools/exp_direct_lane.py -- round 6: ONE thread, one single-frame hsvfilter call per 4K RGBA buffer (the element's contract): two alternating HIP
streams (round 5's element path) against the direct-dispatch lane (MVFX_OPT_DIRECT_DISPATCH: AQL packets without the barrier bit on the library's
own queue, a fence per frame).  16 distinct videotestsrc frames (531 MB), cached and non-temporal accesses; median of 5 x 3000 frames."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 3840, 2160


def main():
    import torch
    import _pkg
    from tests import frames as _frames
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    fpt = int(sys.argv[1]) if len(sys.argv) > 1 else 16   # distinct frames in rotation (33 MB each)
    vts, _ = _frames.videotestsrc_smpte(W, H, min(fpt, 16))
    base = torch.from_numpy(vts.reshape(min(fpt, 16), -1)).to(dev)
    pool = base.repeat((fpt + base.shape[0] - 1) // base.shape[0], 1)[:fpt].contiguous()
    torch.cuda.synchronize()
    fr = (vfx.Frame * fpt)(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
    n = 3000
    print(f"# {fpt} frames in rotation = {fpt * W * H * 4 / 1e6:.0f} MB")
    for rep in range(1):
        for name, opt in (("cached", 0), ("non-temporal", vfx.OPT_NONTEMPORAL)):
            secs = (ctypes.c_double * 5)()
            per = (ctypes.c_double * 1)()
            rc = bench.mvfxbench_hsvfilter_streams_rot(0, 1, 2, 600, n, 5, fr, fpt, None, 0, ctypes.byref(settings), opt, secs, per)
            assert rc == 0, (rc, vfx.last_error())
            a = n / sorted(secs)[2]
            took = ctypes.c_uint64()
            rc = bench.mvfxbench_hsvfilter_direct(0, 600, n, 5, fr, fpt, ctypes.byref(settings), opt, secs, ctypes.byref(took))
            assert rc == 0, (rc, vfx.last_error())
            b = n / sorted(secs)[2]
            print(f"{name:>13}: two streams {a:7.0f} fps ({a * 2 * W * H * 4 / 8e12:.3f})   direct lane {b:7.0f} fps ({b * 2 * W * H * 4 / 8e12:.3f}), "
                  f"{took.value} of {5 * n} launches through the lane", flush=True)


if __name__ == "__main__":
    main()
