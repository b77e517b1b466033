#!/usr/bin/env python3
"""tools/exp_boundary_cost.py -- round 6: what does a kernel boundary cost the single-frame hsvfilter path when two kernels are always co-resident?
The kernel trace (tools/r6_single_trace.sh) shows two launches overlapping completely (duration 25.2 us, start to start 12.6 us) and still 12.6 us
per frame against 11.5 in a 16-frame launch.  Is the loss per PACKET (the acquire / release cache maintenance of every dispatch: an L2 write-back
walk and an invalidate on eight XCDs, whoever else is running) or per streaming kernel (fill / drain)?  One thread, two alternating streams, 4K RGBA:
  single        one frame per launch
  pair          two frames per launch
  pair+tiny     two frames per launch and, behind each, a launch on a 4 x 1 frame (a packet with no work) on the same stream
  pair+tiny/o   ... on the other stream
  single+tiny   one frame per launch + a tiny launch on the same stream
Each: median of 5 x 3000 frames."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import _pkg
    from tests import frames as _frames
    vfx = _pkg.vfx
    lib = vfx.lib()
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    vfx.check(lib.mvfx_thread_set_options(vfx.OPT_NONTEMPORAL))
    W, H, N = 3840, 2160, 16
    vts, _ = _frames.videotestsrc_smpte(W, H, N)
    pool = torch.from_numpy(vts.reshape(N, -1)).to(dev).contiguous()
    tiny = torch.zeros(64, dtype=torch.uint8, device=dev)
    fr = (vfx.Frame * N)(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)])
    ft = vfx.make_frame(tiny.data_ptr(), 4, 1, 16, "RGBA")
    s = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    st = [ctypes.c_void_p(lib.mvfx_thread_stream_n(k)) for k in range(2)]
    single = lambda i, q: lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(fr[i % N]), ctypes.byref(s), q)
    pair = lambda i, q: lib.mvfx_hsvfilter_transform_frames_ip(ctypes.cast(ctypes.byref(fr[(2 * i) % N]), ctypes.POINTER(vfx.Frame)), 2, ctypes.byref(s), q)
    tiny_l = lambda q: lib.mvfx_hsvfilter_transform_frame_ip(ctypes.byref(ft), ctypes.byref(s), q)
    modes = {
        "single": (1, lambda i: single(i, st[i & 1])),
        "pair": (2, lambda i: pair(i, st[i & 1])),
        "pair+tiny": (2, lambda i: (pair(i, st[i & 1]), tiny_l(st[i & 1]))),
        "pair+tiny/o": (2, lambda i: (pair(i, st[i & 1]), tiny_l(st[(i + 1) & 1]))),
        "single+tiny": (1, lambda i: (single(i, st[i & 1]), tiny_l(st[i & 1]))),
        "single+2tiny": (1, lambda i: (single(i, st[i & 1]), tiny_l(st[i & 1]), tiny_l(st[i & 1]))),
    }
    for rep in range(2):
        for name, (fpl, step) in modes.items():
            n = 3000 // fpl
            for i in range(600 // fpl):
                step(i)
            torch.cuda.synchronize()
            ts = []
            for r in range(5):
                t0 = time.perf_counter()
                for i in range(n):
                    step(i)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[2]
            print(f"{name:>13}: {dt / (n * fpl) * 1e6:6.2f} us per frame  ({n * fpl / dt:7.0f} frames/s = {n * fpl / dt * 2 * W * H * 4 / 8e12:.3f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
