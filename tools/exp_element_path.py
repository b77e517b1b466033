#!/usr/bin/env python3
"""tools/exp_element_path.py -- round 4: what ONE streaming thread can get out of single-frame hsvfilter calls (the element's contract:
one call per buffer, hsvfilter/imp.rs:322-326).  4K RGBA videotestsrc-like frames, non-temporal accesses, libmvfxbench.so:
  * streams per thread 1 / 2 / 3 / 4 (the thread alternates its launches over its private streams);
  * MVFXBENCH_SPLIT = 1 / 2 / 4: every call becomes that many launches on horizontal bands of the frame, spread over the streams;
  * the same calls on 64 x 64 frames: the rate at which one thread can issue launches at all (the CPU / runtime ceiling).
Each cell: median of 5 repetitions of 2000 frames."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cell(split, streams, w, h, n_frames=2000):
    import torch
    import _pkg
    vfx = _pkg.vfx
    lib = vfx.lib()
    bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    fb = w * h * 4
    settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
    from tests import frames as _frames
    base = torch.from_numpy(_frames.smpte_like(w, h).reshape(-1)).to(dev)
    fpt = 16
    pool = base.unsqueeze(0).repeat(fpt, 1).contiguous()
    torch.cuda.synchronize()
    fr = (vfx.Frame * fpt)(*[vfx.make_frame(pool[i].data_ptr(), w, h, w * 4, "RGBA") for i in range(fpt)])
    secs = (ctypes.c_double * 5)()
    per = (ctypes.c_double * 1)()
    rc = bench.mvfxbench_hsvfilter_streams_rot(0, 1, streams, 400, n_frames, 5, fr, fpt, None, 0, ctypes.byref(settings), vfx.OPT_NONTEMPORAL, secs, per)
    assert rc == 0, (rc, vfx.last_error())
    fps = n_frames / sorted(secs)[2]
    return fps, fps * 2 * fb / 8e12


def main():
    if len(sys.argv) > 1:  # child: MVFXBENCH_SPLIT is read once per process
        split, streams, w, h = (int(x) for x in sys.argv[1:5])
        fps, frac = cell(split, streams, w, h)
        print(f"{fps:.0f} {frac:.4f}")
        return
    for w, h, label in ((3840, 2160, "3840x2160"), (64, 64, "64x64 (launch rate only)")):
        print(f"# {label}: bands per call x streams per thread -> frames/s (fraction of 8 TB/s)")
        for split in (1, 2, 4):
            row = []
            for streams in (1, 2, 3, 4):
                env = dict(os.environ, MVFXBENCH_SPLIT=str(split))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(split), str(streams), str(w), str(h)], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
                out = r.stdout.strip().splitlines()[-1].split() if r.returncode == 0 and r.stdout.strip() else ["nan", "nan"]
                row.append(f"{streams} streams {float(out[0]):8.0f} ({out[1]})")
            print(f"bands {split}:  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
