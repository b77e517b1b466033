import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The checker is test infrastructure; (re)build it if it is missing (gcc only, <2 s).
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def vfx():
    import _pkg
    return _pkg.vfx


@pytest.fixture(scope="session")
def gpu(vfx):
    """Device 0 through the C ABI; GPU tests fail (not skip) if the HIP path is unusable."""
    n = vfx.lib().mvfx_device_count()
    assert n >= 1, "no HIP device: -m gpu tests need the MI355X box"
    vfx.check(vfx.lib().mvfx_set_device(0))
    return vfx
