"""The C ABI from a plain C99 client (no Python, no C++): builds tests/c_abi_client.c against
include/mi355vfx.h + libmi355vfx.so and runs it.  CPU: the host-only entry points work and the
compute entry points refuse (NO_DEVICE); GPU: the pixels match the reference's known answers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "gst-plugin-rs_amd")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_client")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_client.c"),
           "-L" + LIBDIR, "-lmi355vfx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_client_without_gpu(vfx, tmp_path):
    if vfx.lib().mvfx_device_count() > 0:
        pytest.skip("GPU present: covered by test_c_client_on_gpu")
    r = subprocess.run([_build(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("NO_DEVICE"), r.stdout


@pytest.mark.gpu
def test_c_client_on_gpu(gpu, tmp_path):
    r = subprocess.run([_build(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
