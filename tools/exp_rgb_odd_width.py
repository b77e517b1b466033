#!/usr/bin/env python3
"""tools/exp_rgb_odd_width.py -- round 6: hsvfilter on 16 x 1366 x 768 RGB frames per launch (a width that is not a multiple of four):
the typed kernel with a per-row tail against the VALU kernel those frames took before (MVFX_OPT_HSV_VALU_UNORM).  3 + 3 algorithmic B/px."""
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
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    for W, H in ((1366, 768), (854, 480), (1920, 1080)):
        stride = (3 * W + 3) // 4 * 4
        nb, pool = 16, 24
        src = torch.randint(0, 256, (pool * nb, stride * H), dtype=torch.uint8, device=dev)
        fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, stride, "RGB") for i in range(nb)]) for b in range(pool)]
        s = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
        st = torch.cuda.current_stream(dev)
        sp = ctypes.c_void_p(st.cuda_stream)
        for name, typed, nt in (("typed + row tail", True, False), ("typed, write-through", True, True), ("VALU kernel", False, False)):
            vfx.check(lib.mvfx_thread_set_options(vfx.options(typed=typed, nontemporal=nt).word))
            for i in range(400):
                vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(fi[i % pool], nb, ctypes.byref(s), sp))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 400
            e0.record(st)
            for i in range(n):
                vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(fi[i % pool], nb, ctypes.byref(s), sp))
            e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"{W}x{H} RGB x {nb} per launch, {name:>21}: {us:7.2f} us per launch, {nb / (us * 1e-6):9.0f} fps, frac of 8 TB/s {nb * 6 * W * H / (us * 1e-6) / 8e12:.3f}", flush=True)
        lib.mvfx_thread_set_options(0)


if __name__ == "__main__":
    main()
