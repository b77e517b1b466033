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
    size = "width=3840,height=2160"
    hip = "video/x-raw(memory:HIPMemory)"
    chains = {  # name: (source format, chain)
        "hsvfilter ! colorlut": ("RGBA", f"hsvfilter hue-shift=90 ! colorlut location={cube}"),
        "hsvfilter ! hsvdetector ! colorlut": ("RGBx", f"hsvfilter hue-shift=90 ! hsvdetector hue-ref=120 hue-var=40 ! {hip},format=RGBA ! colorlut location={cube}"),
        "colorlut ! queue ! hsvfilter": ("RGBA", f"colorlut location={cube} ! queue max-size-buffers=4 ! hsvfilter hue-shift=-123.4"),
        "hsvfilter ! queue ! colorlut ! queue ! colorlut": ("RGBA", f"hsvfilter ! queue max-size-buffers=3 ! colorlut location={cube} ! queue max-size-buffers=3 ! colorlut location={cube}"),
        "hsvfilter ! queue ! hsvdetector ! queue ! colorlut": ("RGBx", f"hsvfilter ! queue max-size-buffers=2 ! hsvdetector ! queue max-size-buffers=2 ! {hip},format=RGBA ! colorlut location={cube}"),
        # a consumer that cannot take the lane (hipdownload: a copy on a stream) close behind lane elements: its waits send the producers back to their
        # streams for a while (mvfx_direct_discouraged), then they try again
        "hsvfilter ! queue ! hipdownload": ("RGBA", "hsvfilter hue-shift=45 ! queue max-size-buffers=3 ! hipdownload"),
        "hsvfilter ! colorlut ! queue ! hipdownload": ("RGBA", f"hsvfilter ! colorlut location={cube} ! queue max-size-buffers=3 ! hipdownload"),
        # two readers of one block on two threads behind a tee, both lane elements
        "hsvfilter ! tee ! 2 x (queue ! hsvdetector)": ("RGBx", "hsvfilter hue-shift=30 ! tee name=t t. ! queue max-size-buffers=3 ! hsvdetector ! fakesink sync=false "
                                                                "t. ! queue max-size-buffers=3 ! hsvdetector hue-ref=200"),
    }
    only = os.environ.get("ONLY", "")
    for pool in os.environ.get("POOLS", "5,12").split(","):
        for name, (fmt, chain) in chains.items():
            if only and only not in name:
                continue
            pipe = f"hiptestsrc num-buffers={n} refresh=false ! {hip},format={fmt},{size} ! {chain} ! fakesink sync=false"
            t0 = time.perf_counter()
            r = gst_env.run([launch, "-q"] + pipe.split(), tmp, timeout=600, extra_env=dict({"MVFX_HIP_POOL_MIN": pool}, **{k: v for k, v in os.environ.items() if k.startswith("MVFX_DIRECT") or k in ("MVFX_LANE_STATS", "GPU_MAX_HW_QUEUES", "MVFX_LANE_PARK_AFTER")}))
            dt = time.perf_counter() - t0
            status = "ok" if r.returncode == 0 else f"FAILED rc {r.returncode}: {r.stdout[-400:]}"
            print(f"pool {pool:>2}  {name:<48} {n} buffers in {dt:6.1f} s = {n / dt:8.0f} fps  {status}", flush=True)
            for ln in r.stdout.splitlines():
                if ln.startswith("mvfx lane:"):
                    print("        " + ln, flush=True)


if __name__ == "__main__":
    main()
