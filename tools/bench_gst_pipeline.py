#!/usr/bin/env python3
"""tools/bench_gst_pipeline.py -- pipeline-level frames/s of the real GStreamer elements at 3840x2160:

  host chain:  hiptestsrc ! hsvfilter ! hsvdetector ! colorlut ! fakesink               (system-memory buffers: every element
                                                                                         does H2D + kernel + D2H, as a drop-in
                                                                                         for the reference's CPU elements must)
  HIP chain:   hiptestsrc ! hipupload ! hsvfilter ! hsvdetector ! colorlut ! hipdownload ! fakesink
                                                                                        (memory:HIPMemory between the elements:
                                                                                         one upload, three kernels ordered by
                                                                                         fences, one download)
  device only: hiptestsrc(memory:HIPMemory) ! hsvfilter ! hsvdetector ! colorlut ! fakesink   (no PCIe at all)

`hiptestsrc` hands out pre-painted videotestsrc-smpte frames (no per-frame generator cost; videotestsrc needs longer to paint
a 4K frame than the whole chain needs to filter it; on memory:HIPMemory caps every frame is refreshed from a device-resident
master by one asynchronous device-to-device copy, so in-place filters always see the same input).  Every pipeline runs twice, with N1 and N2 buffers; frames/s =
(N2 - N1) / (t2 - t1), which removes process start-up, plugin loading, LUT parsing and the first-frame allocations.
Run on the GPU box:  python tools/bench_gst_pipeline.py [--width 3840 --height 2160]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cubes, gst_env  # noqa: E402

LAUNCH = gst_env.tool("gst-launch-1.0")


def run(pipeline, tmp, extra_env=None):
    t0 = time.perf_counter()
    r = gst_env.run([LAUNCH, "-q"] + pipeline.split(), tmp, timeout=900, extra_env=extra_env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:])
    return dt


def fps(template, tmp, n1, n2, extra_env=None):
    run(template.format(n=n1), tmp, extra_env)  # page cache, registry
    t1 = min(run(template.format(n=n1), tmp, extra_env) for _ in range(2))
    t2 = min(run(template.format(n=n2), tmp, extra_env) for _ in range(2))
    return (n2 - n1) / (t2 - t1), t1, t2


def rate(template, tmp, n1, n2, extra_env=None):
    """frames/s of ONE gst-launch run of n2 buffers, measured inside the process by hiptestsrc between buffer n1 and the last one
    (MVFX_TESTSRC_RATE: two instants at which a recycled block is back with all downstream work on it finished)."""
    import re
    env = dict(extra_env or {}, MVFX_TESTSRC_RATE=str(n1))
    r = gst_env.run([LAUNCH, "-q"] + template.format(n=n2).split(), tmp, timeout=900, extra_env=env)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:])
    m = re.findall(r"hiptestsrc \S+: ([0-9.]+) buffers/s between buffer", r.stdout)
    if not m:
        raise RuntimeError("no rate line: " + r.stdout[-1000:])
    return sum(float(x) for x in m)  # one line per source (branches)


def branches_main(args):
    """--branches N: N independent `hiptestsrc ! hsvfilter ! fakesink` streams in ONE process (each its own streaming thread and
    HIP stream, each hsvfilter with its own hue-shift), frames born in HBM.  Launch model of the elements: a launch per buffer
    (MVFX_COMBINE unset) against the launch combiner (MVFX_COMBINE=1: the frames of the N threads leave as batched launches with
    per-frame settings).  refresh=false: the source fills every pool block once, so the only HBM traffic is the filter's 8 B/px."""
    import re
    tmp = tempfile.mkdtemp()
    w, h, n = args.width, args.height, args.branches
    caps = f"video/x-raw(memory:HIPMemory),format=RGBA,width={w},height={h},framerate=30/1"
    out = {"frame": f"{w}x{h}", "branches": n, "n1": args.n1, "n2": args.n2, "bytes_per_frame": 2 * w * h * 4}
    for refresh in (("false",) if args.quick else ("false", "true")):
        tpl = " ".join(f"hiptestsrc num-buffers={{n}} refresh={refresh} ! {caps} ! hsvfilter hue-shift={(17 * k) % 360 - 120} saturation-mul={1 + 0.02 * k} "
                       "! fakesink sync=false" for k in range(n))
        for combine in (("0",) if args.quick else ("0", "1", "2")):
            env = {"MVFX_COMBINE": combine, "MVFX_COMBINE_STATS": "1"}
            # ten times the buffers of the single-chain runs: 16 branches at tens of thousands of frames per second finish 500 buffers each
            # inside the noise of a process start
            mul = 1 if args.quick else 10
            key = f"refresh_{refresh}_combine_{combine}"
            if args.quick == 2:  # one run, rate taken inside the process (bench.py's sub-line)
                run(tpl.format(n=min(args.n1, 2000)), tmp, env)  # page cache, registry
                out[key + "_fps"] = round(rate(tpl, tmp, args.n1, args.n2, env), 1)
                out[key + "_frac_of_8TBs"] = round(out[key + "_fps"] * 2 * w * h * 4 / 8e12, 4)
                if args.others and n == 1:  # the two out-of-place elements the same way (one run each)
                    cube = os.path.join(tmp, "look.cube")
                    with open(cube, "w") as f:
                        f.write(cubes.analytic_3d(args.lut))
                    for name, fmt, desc in (("hsvdetector", "RGBx", "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"),
                                            ("colorlut", "RGBA", f"colorlut location={cube}")):
                        c2 = f"video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h},framerate=30/1"
                        t2 = f"hiptestsrc num-buffers={{n}} refresh={refresh} ! {c2} ! {desc} ! fakesink sync=false"
                        run(t2.format(n=2000), tmp, env)
                        out[name + "_fps"] = round(rate(t2, tmp, args.n1, args.n2 * 2 // 3, env), 1)
                continue
            v, t1, t2 = fps(tpl, tmp, args.n1 * mul, args.n2 * mul, env)
            out[key + "_fps"] = round(v * n, 1)
            out[key + "_frac_of_8TBs"] = round(v * n * 2 * w * h * 4 / 8e12, 4)
            if combine != "0":
                r = gst_env.run([LAUNCH, "-q"] + tpl.format(n=args.n2 * 10).split(), tmp, timeout=900, extra_env=env)
                m = re.search(r"mvfx combiner device 0: (\d+) launches for (\d+) frames \(([0-9.]+) frames per launch\), ([0-9.]+) us", r.stdout)
                if m:
                    out[key + "_frames_per_launch"] = float(m.group(3))
                    out[key + "_submit_to_launch_us"] = float(m.group(4))
    print(json.dumps(out), flush=True)


def element_main(args):
    """--element NAME: `hiptestsrc refresh=false ! NAME ! fakesink` on memory:HIPMemory, one streaming thread, with the element's pair
    launches on (default) and off (MVFX_ELEMENT_PAIR=0)."""
    tmp = tempfile.mkdtemp()
    w, h = args.width, args.height
    cube = os.path.join(tmp, "look.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(args.lut))
    desc = {"hsvfilter": ("RGBA", "hsvfilter hue-shift=90", 8),
            "hsvdetector": ("RGBx", "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4", 8),
            "colorlut": ("RGBA", f"colorlut location={cube}", 8)}[args.element]
    caps = f"video/x-raw(memory:HIPMemory),format={desc[0]},width={w},height={h},framerate=30/1"
    out = {"element": args.element, "frame": f"{w}x{h}", "n1": args.n1, "n2": args.n2}
    # one run per figure, the rate taken inside the process (rate()); interleaved repeats, the median is reported
    for refresh in ("false", "true"):
        tpl = f"hiptestsrc num-buffers={{n}} refresh={refresh} ! {caps} ! {desc[1]} ! fakesink sync=false"
        run(tpl.format(n=2000), tmp)  # page cache, registry
        runs = {"1": [], "0": []}
        for rep in range(args.repeats):
            for pair in (("0", "1") if rep % 2 else ("1", "0")):
                runs[pair].append(round(rate(tpl, tmp, args.n1, args.n2, {"MVFX_ELEMENT_PAIR": pair}), 1))
        for pair in ("1", "0"):
            med = sorted(runs[pair])[len(runs[pair]) // 2]
            key = f"refresh_{refresh}_pair_{pair}"
            out[key + "_fps_runs"] = runs[pair]
            out[key + "_fps"] = med
            out[key + "_frac_of_8TBs"] = round(med * desc[2] * w * h / 8e12, 4)
    print(json.dumps(out), flush=True)


def honest_main(args):
    """--honest 1 (round 5, VERDICT r4 missing-2 / W5): the real element on blocks that do NOT sit in the 256 MB Infinity Cache.
    `hiptestsrc refresh=false ! hsvfilter ! fakesink` with MVFX_HIP_POOL_MIN=12: twelve 33 MB blocks in rotation (398 MB), every buffer's
    kernel reads its block from HBM; the same with the pool's usual four blocks (133 MB: cache resident -- reported without a roofline
    fraction); and with a device consumer behind the filter (`! hsvdetector ! fakesink`, RGBx -> RGBA, twelve-block pools on both sides).
    The elements run as shipped: one launch per buffer (MVFX_ELEMENT_PAIR unset)."""
    tmp = tempfile.mkdtemp()
    w, h = args.width, args.height
    size = f"width={w},height={h},framerate=30/1"
    out = {"frame": f"{w}x{h}", "n1": args.n1, "n2": args.n2, "pool_blocks_hbm": 12, "pool_blocks_cache": 4}
    flt = f"hiptestsrc num-buffers={{n}} refresh=false ! video/x-raw(memory:HIPMemory),format=RGBA,{size} ! hsvfilter hue-shift=90 saturation-mul=1.25 saturation-off=-0.05 value-mul=0.9 value-off=0.02 ! fakesink sync=false"
    det = (f"hiptestsrc num-buffers={{n}} refresh=false ! video/x-raw(memory:HIPMemory),format=RGBx,{size} ! hsvfilter hue-shift=90 saturation-mul=1.25 saturation-off=-0.05 value-mul=0.9 value-off=0.02 ! "
           "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! fakesink sync=false")
    run(flt.format(n=min(args.n1, 2000)), tmp)  # page cache, registry
    reps = max(1, args.repeats)
    med = lambda v: sorted(v)[len(v) // 2]
    a = [rate(flt, tmp, args.n1, args.n2, {"MVFX_HIP_POOL_MIN": "12"}) for _ in range(reps)]
    b = [rate(flt, tmp, args.n1, args.n2, {"MVFX_HIP_POOL_MIN": "4"})]
    c = [rate(det, tmp, args.n1, args.n2 * 2 // 3, {"MVFX_HIP_POOL_MIN": "12"})]
    if args.dump_dir:
        # for bench.py's `verified`: the source's frame as system memory, and three buffers of the timed pipeline's element (same caps, same
        # properties, refresh=true so that every buffer is the filtered master frame) downloaded into a file; bench.py compares them with the oracle
        os.makedirs(args.dump_dir, exist_ok=True)
        run(f"hiptestsrc num-buffers=1 ! video/x-raw,format=RGBA,{size} ! filesink location={args.dump_dir}/in.raw", tmp)
        run(flt.format(n=3).replace(" refresh=false", "").replace("fakesink sync=false", f"hipdownload ! filesink location={args.dump_dir}/out.raw"), tmp)
        out["dump"] = {"in": f"{args.dump_dir}/in.raw", "out": f"{args.dump_dir}/out.raw", "frames_out": 3, "width": w, "height": h}
    if args.lut_element:
        # colorlut as the single device element on the same rotation (twelve input blocks + twelve output blocks): 4 + 4 algorithmic B/px
        cube = os.path.join(tmp, "look.cube")
        with open(cube, "w") as f:
            f.write(cubes.analytic_3d(args.lut))
        lut = f"hiptestsrc num-buffers={{n}} refresh=false ! video/x-raw(memory:HIPMemory),format=RGBA,{size} ! colorlut location={cube} ! fakesink sync=false"
        run(lut.format(n=min(args.n1, 2000)), tmp)
        l = [rate(lut, tmp, args.n1, args.n2 * 2 // 3, {"MVFX_HIP_POOL_MIN": "12"}) for _ in range(reps)]
        out["colorlut_hbm_resident_fps"] = round(med(l), 1)
        out["colorlut_hbm_resident_frac_of_8TBs"] = round(med(l) * 2 * w * h * 4 / 8e12, 4)
    out["hsvfilter_hbm_resident_fps"] = round(med(a), 1)
    out["hsvfilter_hbm_resident_runs"] = [round(x, 1) for x in a]
    out["hsvfilter_hbm_resident_frac_of_8TBs"] = round(med(a) * 2 * w * h * 4 / 8e12, 4)
    out["hsvfilter_cache_resident_fps"] = round(med(b), 1)
    out["hsvfilter_then_hsvdetector_hbm_resident_fps"] = round(med(c), 1)
    out["hsvfilter_then_hsvdetector_frac_of_8TBs"] = round(med(c) * 4 * w * h * 4 / 8e12, 4)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lut-element", type=int, default=0, help="--honest: also the colorlut element alone on the HBM-resident rotation")
    ap.add_argument("--dump-dir", default="", help="--honest: also write the source frame and three filtered buffers of the element there (bench.py compares them with the oracle)")
    ap.add_argument("--honest", type=int, default=0, help="1: the single element on a rotation larger than the Infinity Cache, and with a device consumer (bench.py's sub-line)")
    ap.add_argument("--element", default="", help="hsvfilter | hsvdetector | colorlut: the single element on device memory, pair launches on / off")
    ap.add_argument("--only", default="", help="comma-separated pipeline names of the chain mode (default: all)")
    ap.add_argument("--branches", type=int, default=0, help="N parallel hiptestsrc ! hsvfilter ! fakesink streams in one process (launch combiner A/B)")
    ap.add_argument("--quick", type=int, default=0, help="--branches: only refresh=false, no combiner, n1 / n2 taken literally; 2: ONE run, the rate taken inside the process by hiptestsrc (bench.py's sub-line)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--n1", type=int, default=60)
    ap.add_argument("--n2", type=int, default=460)
    ap.add_argument("--lut", type=int, default=33)
    ap.add_argument("--others", type=int, default=0, help="--branches 1 --quick 2: also hsvdetector and colorlut as the single device element")
    ap.add_argument("--repeats", type=int, default=3, help="--element: interleaved repeats per configuration (the median is reported)")
    args = ap.parse_args()
    if args.honest:
        return honest_main(args)
    if args.element:
        return element_main(args)
    if args.branches > 0:
        return branches_main(args)
    tmp = tempfile.mkdtemp()
    cube = os.path.join(tmp, "look.cube")
    with open(cube, "w") as f:
        f.write(cubes.analytic_3d(args.lut))
    w, h = args.width, args.height
    size = f"width={w},height={h},framerate=30/1"
    chain = ("hsvfilter hue-shift=45 ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 "
             f"value-ref=0.6 value-var=0.4 ! {{rgba}} ! colorlut location={cube}")
    sys_rgbx, hip_rgbx = f"video/x-raw,format=RGBx,{size}", f"video/x-raw(memory:HIPMemory),format=RGBx,{size}"
    pipes = {
        "source_only_system": f"hiptestsrc num-buffers={{n}} ! {sys_rgbx} ! fakesink sync=false",
        "host_chain": f"hiptestsrc num-buffers={{n}} ! {sys_rgbx} ! {chain.format(rgba='video/x-raw,format=RGBA')} ! fakesink sync=false",
        "hip_chain": (f"hiptestsrc num-buffers={{n}} ! {sys_rgbx} ! hipupload ! "
                      f"{chain.format(rgba='video/x-raw(memory:HIPMemory),format=RGBA')} ! hipdownload ! fakesink sync=false"),
        # the usual GStreamer decoupling: upload, filters and download on three streaming threads (= three HIP streams,
        # ordered by the fences on the device blocks), so the two PCIe directions and the kernels overlap
        "hip_chain_queues": (f"hiptestsrc num-buffers={{n}} ! {sys_rgbx} ! hipupload ! queue max-size-buffers=4 ! "
                             f"{chain.format(rgba='video/x-raw(memory:HIPMemory),format=RGBA')} ! queue max-size-buffers=4 ! hipdownload "
                             "! fakesink sync=false"),
        "host_chain_queues": (f"hiptestsrc num-buffers={{n}} ! {sys_rgbx} ! queue max-size-buffers=4 ! "
                              + chain.format(rgba='video/x-raw,format=RGBA').replace(" ! ", " ! queue max-size-buffers=4 ! ")
                              + " ! fakesink sync=false"),
        "device_only_chain": (f"hiptestsrc num-buffers={{n}} ! {hip_rgbx} ! "
                              f"{chain.format(rgba='video/x-raw(memory:HIPMemory),format=RGBA')} ! fakesink sync=false"),
    }
    out = {"frame": f"{w}x{h}", "chain": f"hsvfilter ! hsvdetector ! colorlut({args.lut}^3)", "n1": args.n1, "n2": args.n2}
    only = [x for x in args.only.split(",") if x]
    for name, tpl in pipes.items():
        if only and name not in only:
            continue
        # the device-only chain runs ~10 k frames/s: 400 frames are 40 ms, inside the noise of a process start -- 10x the frames there
        n1, n2 = (args.n1 * 10, args.n2 * 10) if name == "device_only_chain" else (args.n1, args.n2)
        v, t1, t2 = fps(tpl, tmp, n1, n2)
        out[name + "_fps"] = round(v, 1)
        out[name + "_seconds"] = [round(t1, 3), round(t2, 3)]
    if only:
        print(json.dumps(out), flush=True)
        return
    # the same HIP chain with pageable staging (what round 1 shipped): upload source and download target malloc'ed
    v, t1, t2 = fps(pipes["hip_chain"], tmp, args.n1, args.n2, {"MVFX_HIP_PAGEABLE": "1"})
    out["hip_chain_pageable_staging_fps"] = round(v, 1)
    out["hip_over_host"] = round(out["hip_chain_fps"] / out["host_chain_fps"], 2)
    out["hip_over_host_with_queues"] = round(out["hip_chain_queues_fps"] / out["host_chain_queues_fps"], 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
