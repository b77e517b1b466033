"""Finds which launch mode of 16 hsvfilter branches crashes with MVFX_ELEMENT_STREAMS=2 and prints a backtrace (rocgdb, batch mode)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gst_env  # noqa: E402

LAUNCH = gst_env.tool("gst-launch-1.0")
tmp = tempfile.mkdtemp()
caps = "video/x-raw(memory:HIPMemory),format=RGBA,width=3840,height=2160,framerate=30/1"
streams = sys.argv[1] if len(sys.argv) > 1 else "2"
for refresh in ("true",):
    for combine in ("1",):
        for n in (6000,):
            tpl = " ".join(f"hiptestsrc num-buffers={n} refresh={refresh} ! {caps} ! hsvfilter hue-shift={(17 * k) % 360 - 120} "
                           "! fakesink sync=false" for k in range(16))
            env = {"MVFX_COMBINE": combine, "MVFX_ELEMENT_STREAMS": streams}
            rcs = []
            for rep in range(3):
                r = gst_env.run([LAUNCH, "-q"] + tpl.split(), tmp, timeout=600, extra_env=env)
                rcs.append(r.returncode)
            print(f"refresh={refresh} combine={combine} n={n}: rcs {rcs}", flush=True)
            if any(rcs):
                print(r.stdout[-1500:], flush=True)
                import subprocess
                so = os.path.join(tmp, "libsegvtrace.so")
                subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-g", os.path.join(ROOT, "tools", "segv_trace.c"), "-o", so])
                os.environ["MVFX_GST_LD_PRELOAD"] = so  # gst_env passes it on as LD_PRELOAD
                for attempt in range(12):
                    g = gst_env.run([LAUNCH, "-q", "-f"] + tpl.split(), tmp, timeout=900, extra_env=env)
                    print("attempt", attempt, "rc", g.returncode, flush=True)
                    if g.returncode != 0:
                        print(g.stdout[-8000:], flush=True)
                        break
                sys.exit(0)
print("no crash")
