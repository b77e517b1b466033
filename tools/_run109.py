import os, sys, tempfile, resource, subprocess, time
sys.path.insert(0, os.getcwd())
from tests import gst_env
tmp = tempfile.mkdtemp()
L = gst_env.tool("gst-launch-1.0")
w, h, n = 3840, 2160, 200000
det = "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
for name, fmt, chain in (("hsvfilter", "RGBA", "hsvfilter hue-shift=90"), ("hsvdetector", "RGBx", det)):
    for pair in ("1", "0"):
        caps = f"video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h},framerate=30/1"
        cmd = f"hiptestsrc num-buffers={n} refresh=false ! {caps} ! {chain} ! fakesink sync=false"
        r0 = resource.getrusage(resource.RUSAGE_CHILDREN); t0 = time.perf_counter()
        r = gst_env.run([L, "-q"] + cmd.split(), tmp, timeout=300, extra_env={"MVFX_ELEMENT_PAIR": pair})
        dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        print(f"{name} pair={pair}: wall {dt:.2f} s, user {r1.ru_utime - r0.ru_utime:.2f} s, sys {r1.ru_stime - r0.ru_stime:.2f} s, "
              f"vol ctx {r1.ru_nvcsw - r0.ru_nvcsw}, invol ctx {r1.ru_nivcsw - r0.ru_nivcsw}", flush=True)
