#!/usr/bin/env python3
"""tools/exp_colorlut_variants.py [--noise 0 3 5 8 16] [--random] [--check]: every gst-plugin-rs_amd/build_ab/lib_*.so (tools/ab_colorlut.sh)
through the colorlut 33^3 noise sweep of bench.py -- 16 x 4K natural-like frames per launch, HIP events around 30 launches -- one child
process per library.  --check compares every variant's output on the +-8 frames with the first library's (bit-exact or it says so)."""
import argparse
import ctypes
import glob
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    import _pkg
    import bench
    from tests import cubes
    vfx = _pkg.vfx
    lib = vfx.lib()
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H, N = 3840, 2160, 16
    lut = vfx.CubeLut(cubes.analytic_3d(args.size))
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    vfx.check(lib.mvfx_thread_set_options(vfx.options(placement=args.placement).word))
    src = torch.empty((2 * N, W * H * 4), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    fi = [(vfx.Frame * N)(*[vfx.make_frame(src[b * N + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)]) for b in range(2)]
    fo = [(vfx.Frame * N)(*[vfx.make_frame(dst[b * N + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(N)]) for b in range(2)]

    def measure():
        for i in range(6):
            vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[i & 1], fo[i & 1], N, sptr))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 0.0
        for _ in range(args.reps):
            e0.record()
            for i in range(30):
                vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[i & 1], fo[i & 1], N, sptr))
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 30 * N / (e0.elapsed_time(e1) * 1e-3))
        return best
    # settle the clocks
    src.random_(0, 256, generator=gen)
    for i in range(40):
        vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[i & 1], fo[i & 1], N, sptr))
    torch.cuda.synchronize()
    out = []
    for amp in args.noise:
        gen.manual_seed(1000 + amp)
        bench.fill_frames(torch, dev, gen, src, "natural", W, H, first_frame=0, noise=amp)
        out.append(f"+-{amp}: {measure() / 1e3:6.1f}k")
        if args.check and amp == args.noise[-1]:
            torch.cuda.synchronize()
            out.append("md5 " + hashlib.md5(dst[:4].cpu().numpy().tobytes()).hexdigest()[:10])
    if args.random:
        gen.manual_seed(99)
        src.random_(0, 256, generator=gen)
        out.append(f"random: {measure() / 1e3:6.1f}k")
        if args.check:
            out.append("md5 " + hashlib.md5(dst[:2].cpu().numpy().tobytes()).hexdigest()[:10])
    print(f"{os.environ.get('MVFX_VARIANT', '?'):28s} " + "  ".join(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", type=int, nargs="*", default=[0, 3, 5, 8, 16])
    ap.add_argument("--random", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--size", type=int, default=33)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--placement", type=int, default=0, help="MVFX_OPT_LUT_PLACEMENT of the measured calls (7: round 4's per-wave windows)")
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--tiles", type=int, nargs="*", default=[0], help="MVFX_XTILE_TILES values to run every library with (0 = the launcher's choice)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = sorted(glob.glob(os.path.join(ROOT, "gst-plugin-rs_amd", "build_ab", "lib_*.so")))
    libs = [os.path.join(ROOT, "gst-plugin-rs_amd", "libmi355vfx.so")] + libs
    for so in libs:
        name = os.path.basename(so)[4:-3] if "build_ab" in so else "shipped"
        if args.only and name not in args.only:
            continue
        for tiles in args.tiles:
            name_ = name + (f"@T{tiles}" if tiles else "") + (f"@P{args.placement}" if args.placement else "")
            env = dict(os.environ, MVFX_LIB=so, MVFX_VARIANT=name_)
            if tiles:
                env["MVFX_XTILE_TILES"] = str(tiles)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:] if a != "--child"], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith(name_)]
            print(lines[-1] if lines else f"{name_}: FAILED rc {r.returncode}\n{r.stdout[-1500:]}", flush=True)


if __name__ == "__main__":
    main()
