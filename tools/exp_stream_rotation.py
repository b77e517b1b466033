#!/usr/bin/env python3
"""tools/exp_stream_rotation.py -- one video stream = one streaming thread, but consecutive buffers are independent frames: what does
a thread gain by rotating its single-frame launches over 2 or 4 private HIP streams (mvfx_thread_stream_n) instead of one?
N host threads x S streams per thread x single-frame mvfx_hsvfilter_transform_frame_ip on 4K RGBA videotestsrc-like frames."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H = 3840, 2160
    fb = W * H * 4
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    from tests import frames as _frames
    base = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
    print("threads x streams/thread: frames/s, fraction of 8 TB/s (median of 3 repetitions; nt = non-temporal accesses)")
    for nt in (0, 1):
        opts = vfx.OPT_NONTEMPORAL if nt else 0
        for n in (1, 2, 4, 16):
            fpt = 12 if n <= 4 else 6
            pool = base.unsqueeze(0).repeat(n * fpt, 1).contiguous()
            torch.cuda.synchronize()
            fr = (vfx.Frame * (n * fpt))(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(n * fpt)])
            row = []
            for s in (1, 2, 4):
                secs = (ctypes.c_double * 3)()
                per = (ctypes.c_double * n)()
                for launches in (max(300, 4000 // n), max(100, 1200 // n)):
                    rc = bench.mvfxbench_hsvfilter_streams_rot(0, n, s, 20, launches, 3, fr, fpt, None, 0, ctypes.byref(settings), opts, secs, per)
                    assert rc == 0, (rc, vfx.last_error())
                fps = n * launches / sorted(secs)[1]
                row.append(f"{s} stream(s): {fps:8.0f} = {fps * 2 * fb / 8e12:.3f}")
            print(f"nt={nt} threads {n:2d}   " + "   ".join(row), flush=True)
            del pool


if __name__ == "__main__":
    main()
