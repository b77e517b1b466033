#!/usr/bin/env python3
"""tools/soak_lane_chain.py [buffers] -- round 6: long runs of device-only chains whose elements all take the direct-dispatch lane (hsvfilter in place,
colorlut with and without the barrier bit, hsvdetector), 4K RGBA on recycled pools of several sizes (odd pools make an element meet its own dispatch on
the lane's other queue), with and without a queue element between the stages.  What it checks is that every run ENDS (no fence is waited for that
never fires, no packet is lost) and how fast; the bytes are the pipeline tests' business (tests/test_gst_pipelines_gpu.py)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests import cubes, gst_env
    launch = gst_env.tool("gst-launch-1.0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    tmp = tempfile.mkdtemp(prefix="soak_lane_")
    cube = os.path.join(tmp, "look.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(33))
    caps = "video/x-raw(memory:HIPMemory),format=RGBA,width=3840,height=2160"
    chains = {
        "hsvfilter ! colorlut": f"hsvfilter hue-shift=90 ! colorlut location={cube}",
        "hsvfilter ! colorlut ! hsvdetector": f"hsvfilter hue-shift=90 ! colorlut location={cube} ! video/x-raw(memory:HIPMemory),format=RGBx ! hsvdetector hue-ref=120 hue-var=40",
        "colorlut ! queue ! hsvfilter": f"colorlut location={cube} ! queue max-size-buffers=4 ! hsvfilter hue-shift=-123.4",
        "hsvfilter ! queue ! colorlut ! queue ! colorlut": f"hsvfilter ! queue max-size-buffers=3 ! colorlut location={cube} ! queue max-size-buffers=3 ! colorlut location={cube}",
    }
    for pool in ("5", "12"):
        for name, chain in chains.items():
            pipe = f"hiptestsrc num-buffers={n} refresh=false ! {caps} ! {chain} ! fakesink sync=false"
            t0 = time.perf_counter()
            r = gst_env.run([launch, "-q"] + pipe.split(), tmp, timeout=600, extra_env={"MVFX_HIP_POOL_MIN": pool})
            dt = time.perf_counter() - t0
            status = "ok" if r.returncode == 0 else f"FAILED rc {r.returncode}: {r.stdout[-400:]}"
            print(f"pool {pool:>2}  {name:<48} {n} buffers in {dt:6.1f} s = {n / dt:8.0f} fps  {status}", flush=True)


if __name__ == "__main__":
    main()
