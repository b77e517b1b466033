#!/usr/bin/env python3
"""tools/exp_single_shape.py -- round 6 (NEEDS the MVFX_EXP_SINGLE hook of commit b1acdc1: it was removed with the negative result; kept as the record of what was measured): the launch shape of ONE 4K RGBA frame per hsvfilter call (the element's contract, hsvfilter/imp.rs:322-326).
One host thread, single-frame mvfx_hsvfilter_transform_frame_ip calls rotating over 1 / 2 / 3 private streams, 16 distinct 33 MB frames (531 MB:
nothing stays in the Infinity Cache), non-temporal accesses.  MVFX_EXP_SINGLE=tile,iters[,maxgrid] picks the experimental shape (hsv_kernels.hip):
`tile` 16-byte groups per lane with the loads issued together, `iters` adjacent chunks per workgroup, grid capped at `maxgrid` workgroups.
Each cell: median of 5 repetitions of 3000 frames.   python tools/exp_single_shape.py [shape ...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 3840, 2160


def cell(streams, n_frames=3000):
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    from tests import frames as _frames
    fpt = 16
    vts, _ = _frames.videotestsrc_smpte(W, H, fpt)
    pool = torch.from_numpy(vts.reshape(fpt, -1)).to(dev).contiguous()
    torch.cuda.synchronize()
    fr = (vfx.Frame * fpt)(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
    out = []
    for st in streams:
        secs = (ctypes.c_double * 5)()
        per = (ctypes.c_double * 1)()
        rc = bench.mvfxbench_hsvfilter_streams_rot(0, 1, st, 600, n_frames, 5, fr, fpt, None, 0, ctypes.byref(settings), vfx.OPT_NONTEMPORAL, secs, per)
        assert rc == 0, (rc, vfx.last_error())
        out.append(n_frames / sorted(secs)[2])
    return out


def main():
    if os.environ.get("MVFX_EXP_CHILD"):
        print(" ".join(f"{v:.0f}" for v in cell([int(x) for x in sys.argv[1].split(",")])))
        return
    shapes = sys.argv[1:] or ["", "2,1", "2,2", "2,3", "2,4", "2,8", "1,2", "1,4", "1,8", "2,2,1024", "2,1,2048", "1,2,2048"]
    streams = "1,2,3"
    print(f"# shape (tile,iters[,maxgrid]; '' = the shipped launch) -> frames/s with {streams} streams per thread (fraction of 8 TB/s)")
    for rep in range(2):
        for shape in shapes:
            env = dict(os.environ, MVFX_EXP_CHILD="1")
            env.pop("MVFX_EXP_SINGLE", None)
            if shape:
                env["MVFX_EXP_SINGLE"] = shape
            r = subprocess.run([sys.executable, os.path.abspath(__file__), streams], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            if r.returncode != 0:
                print(f"{shape or 'shipped':>10}: FAILED {r.stderr[-300:]}")
                continue
            v = [float(x) for x in r.stdout.strip().splitlines()[-1].split()]
            print(f"{shape or 'shipped':>10}: " + "   ".join(f"{s} str {x:7.0f} ({x * 2 * W * H * 4 / 8e12:.3f})" for s, x in zip(streams.split(","), v)), flush=True)


if __name__ == "__main__":
    main()
