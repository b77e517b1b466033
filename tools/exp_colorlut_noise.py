#!/usr/bin/env python3
"""tools/exp_colorlut_noise.py -- colorlut 33^3 on 4K frames of smooth gradients + uniform noise of +-A codes per channel: the x-prelerped
window kernel (automatic choice) against round 2's cell-window kernel (placement 5), 16 frames per launch.  Where does a pixel that
leaves the wave's window start to cost more on the new kernel (two 24-byte entries of a 6.9 MB table) than on the old one (one 96-byte
cell of a 3.45 MB table)?"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import _pkg
    from tests import cubes
    vfx = _pkg.vfx
    lib = vfx.lib()
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H, N = 3840, 2160, 16
    lut = vfx.CubeLut(cubes.analytic_3d(33))
    x = torch.linspace(0, 1, W, device=dev).view(1, W)
    y = torch.linspace(0, 1, H, device=dev).view(H, 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    print("noise +-A: frames/s auto (x-prelerped) | placement 5 (cell window)   [16 x 4K per launch]")
    for amp in (0, 1, 3, 5, 8, 12, 16, 24, 48, 128):
        src = torch.empty((N, W * H * 4), dtype=torch.uint8, device=dev)
        for k in range(N):
            ph = 0.37 * k
            img = torch.stack([(0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)).expand(H, W), (0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)).expand(H, W),
                               (0.5 + 0.45 * torch.cos(4.0 * x * y + ph)).expand(H, W), torch.ones((H, W), device=dev)], dim=-1) * 255.0
            if amp:
                noise = torch.randint(-amp, amp + 1, img.shape, device=dev, generator=gen).float()
                noise[..., 3] = 0
                img = img + noise
            src[k] = img.clamp(0, 255).to(torch.uint8).view(-1)
        dst = torch.empty_like(src)
        fi = (vfx.Frame * N)(*[vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)])
        fo = (vfx.Frame * N)(*[vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)])
        row = []
        for placement in (0, 5):
            vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=placement).word))
            for _ in range(5):
                vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi, fo, N, sptr))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = 30
            for _ in range(it):
                vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi, fo, N, sptr))
            torch.cuda.synchronize()
            row.append(N * it / (time.perf_counter() - t0))
        vfx.check(lib.mvfx_thread_set_options(0))
        print(f"A = {amp:3d}: {row[0]:9.0f} | {row[1]:9.0f}   ratio {row[0] / row[1]:.2f}", flush=True)


if __name__ == "__main__":
    main()
