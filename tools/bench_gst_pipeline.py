#!/usr/bin/env python3
"""tools/bench_gst_pipeline.py -- pipeline-level frames/s of the real GStreamer elements, host path
(system-memory buffers: H2D + kernel + D2H inside every element) vs device path
(`memory:HIPMemory`: one hipupload, N elements in HBM, one hipdownload).  The source is a looped
raw frame from RAM (multifilesrc-free: videotestsrc pattern=black is the cheapest generator in the
image), so the numbers bound the element chain, not the generator.  Run on the GPU box."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cubes, gst_env  # noqa: E402

LAUNCH = gst_env.tool("gst-launch-1.0")


def run(pipeline, tmp):
    t0 = time.perf_counter()
    r = gst_env.run([LAUNCH, "-q"] + pipeline.split(), tmp, timeout=600)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(r.stdout)
    return dt


def main():
    tmp = tempfile.mkdtemp()
    cube = os.path.join(tmp, "look.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(17))
    for (w, h, n) in ((1920, 1080, 300), (3840, 2160, 120)):
        src = f"videotestsrc pattern=black num-buffers={n} ! video/x-raw,format=RGBx,width={w},height={h}"
        chain = ("hsvfilter hue-shift=45 ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 "
                 f"value-ref=0.6 value-var=0.4 ! {{caps}} ! colorlut location={cube}")
        base = run(f"{src} ! fakesink", tmp)
        host = run(f"{src} ! {chain.format(caps='video/x-raw,format=RGBA')} ! fakesink", tmp)
        hip = run(f"{src} ! hipupload ! {chain.format(caps='video/x-raw(memory:HIPMemory),format=RGBA')} ! hipdownload ! fakesink", tmp)
        print(json.dumps({"frame": f"{w}x{h}", "buffers": n, "source_only_fps": round(n / base, 1),
                          "host_path_fps": round(n / host, 1), "hipmemory_path_fps": round(n / hip, 1),
                          "chain": "hsvfilter ! hsvdetector ! colorlut(17^3)"}), flush=True)


if __name__ == "__main__":
    main()
