#!/usr/bin/env python3
"""tools/check_lane_threads.py [library.so]: libmvfxbench.so's mvfxbench_lane_threads (two threads, three fences each, 6000 one-frame lane calls each) in a
child process with a 60 s timeout -- with the shipped library, and (argument) with another build of libmi355vfx.so preloaded in front of it.  Used once in
round 6 to show that tests/test_direct_dispatch_gpu.py::test_two_threads_few_fences deadlocks on the lane as it was before the fix."""
import os
import subprocess
import sys

CODE = """
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import _pkg
from tests import frames
gpu = _pkg.vfx
gpu.check(gpu.lib().mvfx_set_device(0))
bench = ctypes.CDLL("gst-plugin-rs_amd/libmvfxbench.so")
w, h, threads, per, launches = [int(x) for x in os.environ.get("SHAPE", "1920,1080,2,2,6000").split(",")]
nev = int(os.environ.get("EVENTS", "3"))
host = [frames.random_frame(0x5EED1400 + k, w, h) for k in range(threads * per)]
bufs = [gpu.DeviceBuffer(f.nbytes).upload(f) for f in host]
fr = (gpu.Frame * (threads * per))(*[gpu.make_frame(b.ptr, w, h, w * 4, "RGBA") for b in bufs])
s = gpu.HsvFilterSettings(0.0, 1.0, 0.0, 1.0, 0.0)
took = ctypes.c_uint64()
rc = bench.mvfxbench_lane_threads(0, threads, nev, launches, fr, per, ctypes.byref(s), ctypes.byref(took))
print("rc", rc, "took", took.value)
"""

for lib in [""] + sys.argv[1:]:
    env = dict(os.environ)
    if lib:
        env["LD_PRELOAD"] = os.path.abspath(lib)
        env["MVFX_LIB"] = os.path.abspath(lib)
    try:
        r = subprocess.run([sys.executable, "-c", CODE], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
        print(lib or "shipped", "->", (r.stdout.strip().splitlines() or ["(no output)"])[-1], flush=True)
    except subprocess.TimeoutExpired:
        print(lib or "shipped", "-> TIMEOUT after 60 s (deadlock)", flush=True)
