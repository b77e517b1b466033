"""Locates the GStreamer tools of the image (conda GStreamer 1.14 under /opt/conda) and our
plugins; shared by the CPU surface test and the GPU pipeline tests."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MVFX_GST_PLUGIN_DIR / MVFX_GST_LD_PRELOAD: `make asan-test` points the same tests at the sanitizer build of the plugins
PLUGIN_DIR = os.environ.get("MVFX_GST_PLUGIN_DIR") or os.path.join(ROOT, "gst-plugin-rs_amd", "gst-plugins")
PLUGINS = ["libgsthsv.so", "libgstcolorlut.so", "libgstrsvideofx.so", "libgstmi355hip.so", "libgstimagers.so"]


def tool(name):
    for cand in (os.path.join("/opt/conda/bin", name), shutil.which(name)):
        if cand and os.path.exists(cand):
            return cand
    return None


def available():
    return tool("gst-launch-1.0") is not None and all(os.path.exists(os.path.join(PLUGIN_DIR, p)) for p in PLUGINS)


def env(tmpdir):
    e = dict(os.environ)
    e["PATH"] = "/opt/conda/bin:" + e.get("PATH", "")
    if os.path.isdir("/opt/conda/lib/gstreamer-1.0"):
        e["GST_PLUGIN_SYSTEM_PATH"] = "/opt/conda/lib/gstreamer-1.0"
    e["GST_PLUGIN_PATH"] = PLUGIN_DIR
    e["GST_REGISTRY"] = os.path.join(str(tmpdir), "registry.bin")
    e["GST_REGISTRY_FORK"] = "no"
    # the reference's CI runs its tests with G_DEBUG=fatal_warnings (ci/run-cargo-test.sh:25): so do these -- a g_warning() or
    # g_critical() anywhere in a pipeline aborts the tool and fails the test
    e.setdefault("G_DEBUG", "fatal-warnings")
    e.pop("LD_PRELOAD", None)
    if os.environ.get("MVFX_GST_LD_PRELOAD"):
        e["LD_PRELOAD"] = os.environ["MVFX_GST_LD_PRELOAD"]
    return e


def run(args, tmpdir, timeout=120, extra_env=None):
    e = env(tmpdir)
    if os.environ.get("MVFX_GST_LD_PRELOAD") and shutil.which("setarch"):
        # gcc 11's libasan cannot place its shadow memory under 32-bit mmap randomisation: run the tool without ASLR
        import platform
        args = ["setarch", platform.machine(), "-R"] + list(args)
    if extra_env:
        e.update(extra_env)
    return subprocess.run(args, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
