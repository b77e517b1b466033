"""SURVEY 5 "compile host code with -fsanitize=address,undefined in a CI target": `make -C gst-plugin-rs_amd asan-test`
builds the .cube parser / MMCQ fuzz driver and the element layer (libmvfxgst.so + the four plugins) with ASan + UBSan on
the CPU box and runs the fuzz driver and the CPU element tests (gst-inspect surface, roundedcorners pipelines) against
the instrumented plugins.  Never the GPU build."""
import os
import shutil
import subprocess

import pytest

from tests import gst_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not gst_env.available() or shutil.which("g++") is None or shutil.which("setarch") is None
                    or os.environ.get("MVFX_GST_LD_PRELOAD"), reason="needs GStreamer dev files + g++ (and is not itself run under asan)")
@pytest.mark.timeout(600)
def test_sanitizer_build_of_the_host_code_is_clean():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gst-plugin-rs_amd"), "asan-test", "FUZZ_ITERS=15000"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=580)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "fuzz ok:" in r.stdout and " passed" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout
