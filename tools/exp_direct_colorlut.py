#!/usr/bin/env python3
"""tools/exp_direct_colorlut.py -- round 6: ONE thread, one single-frame colorlut call per 4K RGBA buffer pair (the element's contract), 33^3 LUT:
ordinary launches on two alternating HIP streams against the direct-dispatch lane (MVFX_OPT_DIRECT_DISPATCH) in queue order and without the
barrier bit (MVFX_OPT_DIRECT_UNORDERED: the frames are independent), per content (videotestsrc,
natural-like, natural-like + noise) and per window kernel (content probe, workgroup window forced, per-wave windows forced).  12 frame pairs in
rotation (0.8 GB: HBM-resident); median of 5 x 2000 frames; fraction of 8 TB/s at 4 + 4 bytes per pixel.
Arguments (all optional, for traces): --contents natural-like --kernels "wave windows" --modes unordered   (comma separated subsets)."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 3840, 2160


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contents", default="")
    ap.add_argument("--kernels", default="")
    ap.add_argument("--modes", default="streams,in_order,unordered")
    args = ap.parse_args()
    import torch
    import _pkg
    from tests import cubes, frames as _frames
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    lut = vfx.CubeLut(cubes.analytic_3d(33))
    fpt, n = 12, 2000
    contents = {}
    vts, _ = _frames.videotestsrc_smpte(W, H, fpt)
    contents["videotestsrc"] = vts.reshape(fpt, -1)
    nat = np.stack([np.ascontiguousarray(_frames.natural_like(W, H, 0x5EED0F00 + k)).reshape(-1) for k in range(fpt)])
    contents["natural-like"] = nat
    rng = np.random.default_rng(0x5EED0F20)
    noisy = nat.reshape(fpt, H, W, 4).copy()
    noisy[..., :3] = np.clip(noisy[..., :3].astype(np.int16) + rng.integers(-8, 9, (fpt, H, W, 3), dtype=np.int16), 0, 255).astype(np.uint8)
    contents["natural +-8"] = noisy.reshape(fpt, -1)
    print(f"# {fpt} frame pairs in rotation = {2 * fpt * W * H * 4 / 1e6:.0f} MB")
    modes = [m for m in args.modes.split(",") if m]
    for cname, host in contents.items():
        if args.contents and cname not in args.contents.split(","):
            continue
        src = torch.from_numpy(host).to(dev)
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        fi = (vfx.Frame * fpt)(*[vfx.make_frame(src[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
        fo = (vfx.Frame * fpt)(*[vfx.make_frame(dst[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
        for kname, kopt in (("probe", 0), ("workgroup window", vfx.OPT_LUT_WG_WINDOW), ("wave windows", vfx.options(placement=7).word)):
            if args.kernels and kname not in args.kernels.split(","):
                continue
            out = []
            for mode, direct in (("streams", 0), ("in_order", vfx.OPT_DIRECT_DISPATCH), ("unordered", vfx.OPT_DIRECT_DISPATCH | vfx.OPT_DIRECT_UNORDERED)):
                if mode not in modes:
                    out.append((0.0, 0))
                    continue
                secs = (ctypes.c_double * 5)()
                took = ctypes.c_uint64()
                rc = bench.mvfxbench_colorlut_direct(0, 400, n, 5, lut.h, fi, fo, fpt, kopt | direct, secs, ctypes.byref(took))
                assert rc == 0, (rc, vfx.last_error())
                out.append((n / sorted(secs)[2], took.value))
            (a, _), (b, took), (c, took_u) = out
            print(f"{cname:>13} {kname:>17}: two streams {a:7.0f} fps ({a * 2 * W * H * 4 / 8e12:.3f})   lane in order {b:7.0f} ({b * 2 * W * H * 4 / 8e12:.3f})   "
                  f"lane unordered {c:7.0f} ({c * 2 * W * H * 4 / 8e12:.3f}); {took} + {took_u} of 2 x {5 * n} through the lane", flush=True)


if __name__ == "__main__":
    main()
