"""roundedcorners alpha mask on the CPU box: mvfx_roundedcorners_mask_host replays
video/videofx/src/border/imp.rs:57-149 through the system libcairo (the library the reference itself
calls), touches no device, and is byte-identical to the committed libcairo goldens for every case,
2*radius > min(width, height) included."""
import ctypes
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def masks():
    return np.load(os.path.join(GOLDEN, "roundedcorners_masks.npz"))


def _cases():
    with np.load(os.path.join(GOLDEN, "roundedcorners_masks.npz")) as g:
        return [k for k in g.files if k != "cairo_version"]


def test_cairo_is_loadable(vfx, masks):
    ver = vfx.lib().mvfx_roundedcorners_cairo_version()
    assert ver, "libcairo.so.2 not loadable: the reference's mask cannot be reproduced on this box"
    # goldens are cairo-version dependent, like the reference's own output
    assert ver.decode().split(".")[:2] == bytes(masks["cairo_version"]).decode().split(".")[:2]


@pytest.mark.parametrize("case", _cases())
def test_mask_host_equals_cairo_golden(vfx, masks, case):
    gold = masks[case]
    w, h, r = (int(t[1:]) for t in case.split("_"))
    rows, stride = gold.shape
    buf = np.full(gold.size, 0x77, np.uint8)
    vfx.check(vfx.lib().mvfx_roundedcorners_mask_host(buf.ctypes.data_as(ctypes.c_void_p), w, h, stride, r))
    assert np.array_equal(buf.reshape(rows, stride), gold)


def test_mask_host_argument_errors(vfx):
    lib = vfx.lib()
    buf = np.zeros(64 * 48, np.uint8)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    assert lib.mvfx_roundedcorners_mask_host(None, 64, 48, 64, 4) == vfx.ERR_INVALID_ARGUMENT
    assert lib.mvfx_roundedcorners_mask_host(p, 64, 48, 32, 4) == vfx.ERR_INVALID_ARGUMENT      # stride < width
    assert lib.mvfx_roundedcorners_mask_host(p, 0, 48, 64, 4) == vfx.ERR_INVALID_ARGUMENT
    # cairo itself refuses a stride that is not a multiple of 4, as it does for the reference
    big = np.zeros(66 * 48, np.uint8)
    rc = lib.mvfx_roundedcorners_mask_host(big.ctypes.data_as(ctypes.c_void_p), 64, 48, 66, 4)
    assert rc == vfx.ERR_INVALID_ARGUMENT and "cairo image surface" in vfx.last_error()
    # radius 0 never reaches cairo: any stride, 0xFF everywhere (border/imp.rs:123-128)
    vfx.check(lib.mvfx_roundedcorners_mask_host(big.ctypes.data_as(ctypes.c_void_p), 64, 48, 66, 0))
    assert (big == 0xFF).all()
