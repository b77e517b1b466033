"""Runs a scenario of tests/gst_worker.py (in-process GStreamer through PyGObject) in a child process with the plugin
environment of tests/gst_env.py.  PyGObject is the system interpreter's; the GStreamer it binds is the image's conda build:
its typelibs and libraries are put on the search paths and the system libstdc++ is preloaded (libamdhip64 needs a newer one
than conda ships)."""
import json
import os
import subprocess
import sys
import tempfile

from tests import gst_env

TYPELIBS = "/opt/conda/lib/girepository-1.0"
STDCXX = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"


def available():
    if not gst_env.available() or not os.path.exists(os.path.join(TYPELIBS, "Gst-1.0.typelib")) or not os.path.exists(STDCXX):
        return False
    try:
        import gi  # noqa: F401
    except ImportError:
        return False
    return True


def run(scenario, arg="", timeout=300):
    tmp = tempfile.mkdtemp()
    e = gst_env.env(tmp)
    e["GI_TYPELIB_PATH"] = TYPELIBS
    e["LD_LIBRARY_PATH"] = "/opt/conda/lib:" + e.get("LD_LIBRARY_PATH", "")
    e["LD_PRELOAD"] = STDCXX
    e["MVFX_WORKER_TMP"] = tmp
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gst_worker.py")
    cmd = [sys.executable, worker, scenario, str(arg)]
    if os.environ.get("MVFX_GST_LD_PRELOAD"):  # `make asan-test`: the sanitizer runtime first, and no ASLR (see gst_env.run)
        import platform
        import shutil
        e["LD_PRELOAD"] = os.environ["MVFX_GST_LD_PRELOAD"] + " " + STDCXX
        if shutil.which("setarch"):
            cmd = ["setarch", platform.machine(), "-R"] + cmd
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise AssertionError(f"worker {scenario} {arg}: rc {r.returncode}\n{r.stdout[-3000:]}")
