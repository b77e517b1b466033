"""CPU checks of the drop-in boundary: libmi355vfx.so loads without a GPU, exports every symbol
include/mi355vfx.h declares, the Python binding table covers them all, enum values agree with the
oracle's, and compute entry points fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np

from tests import oracle_binding as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    with open(os.path.join(ROOT, "include", "mi355vfx.h")) as f:
        return f.read()


def _declared_functions():
    text = re.sub(r"/\*.*?\*/", "", _header(), flags=re.S)
    return sorted(set(re.findall(r"\b(mvfx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(vfx):
    handle = ctypes.CDLL(vfx.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} is declared in include/mi355vfx.h but not exported"


def test_binding_table_matches_header(vfx):
    assert sorted(vfx.SIGNATURES) == _declared_functions()


def test_format_enum_matches_oracle_and_header(vfx):
    text = _header()
    for name, val in vfx.FORMATS.items():
        m = re.search(rf"MVFX_FORMAT_{name.upper()}\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == val, name
        assert orc.FORMATS[name] == val
    with open(os.path.join(ROOT, "oracle", "oracle.h")) as f:
        otext = f.read()
    for name, val in vfx.FORMATS.items():
        m = re.search(rf"ORC_FORMAT_{name.upper()}\s*=\s*(\d+)", otext)
        assert m and int(m.group(1)) == val, name


def test_status_strings(vfx):
    lib = vfx.lib()
    assert lib.mvfx_abi_version() == 2
    assert lib.mvfx_status_string(0) == b"ok"
    for code in range(-10, 0):
        assert lib.mvfx_status_string(code) not in (b"ok", b"unknown status")


def test_no_cpu_fallback_without_device(vfx):
    """On a box without a GPU every compute entry point must fail with NO_DEVICE, never compute."""
    lib = vfx.lib()
    if lib.mvfx_device_count() > 0:
        return  # GPU box: covered by the -m gpu tests
    frame = np.arange(64, dtype=np.uint8)
    before = frame.copy()
    f = vfx.make_frame(frame.ctypes.data, 4, 4, 16, "RGBA")
    rc = lib.mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(vfx.HsvFilterSettings.default()))
    assert rc == vfx.ERR_NO_DEVICE and "no CPU fallback" in vfx.last_error()
    assert np.array_equal(frame, before)
    out = np.zeros(64, np.uint8)
    fo = vfx.make_frame(out.ctypes.data, 4, 4, 16, "RGBA")
    fi = vfx.make_frame(frame.ctypes.data, 4, 4, 16, "RGBx")
    rc = lib.mvfx_hsvdetector_transform_frame_host(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(vfx.HsvDetectorSettings.default()))
    assert rc == vfx.ERR_NO_DEVICE
    lut = vfx.CubeLut("LUT_1D_SIZE 2\n0 0 0\n1 1 1\n")  # parsing is host-only and works
    fi = vfx.make_frame(frame.ctypes.data, 4, 4, 16, "RGBA")
    rc = lib.mvfx_colorlut_transform_frame_host(lut.h, ctypes.byref(fi), ctypes.byref(fo))
    assert rc == vfx.ERR_NO_DEVICE
    assert not out.any()
    p = ctypes.c_void_p()
    assert lib.mvfx_device_alloc(ctypes.byref(p), 16) == vfx.ERR_NO_DEVICE


def test_argument_validation_needs_no_device(vfx):
    lib = vfx.lib()
    s = vfx.HsvFilterSettings.default()
    assert lib.mvfx_hsvfilter_transform_frame_ip_host(None, ctypes.byref(s)) == vfx.ERR_INVALID_ARGUMENT
    a = np.zeros(64, np.uint8)
    f = vfx.make_frame(a.ctypes.data, 4, 4, 16, "I420")
    assert lib.mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(s)) == vfx.ERR_UNSUPPORTED_FORMAT
    assert lib.mvfx_thread_set_options(0x1000) == vfx.ERR_INVALID_ARGUMENT                       # unknown bit
    assert lib.mvfx_thread_set_options(vfx.OPT_HSV_LITERAL | vfx.OPT_HSV_FORCE_FAST) == vfx.ERR_INVALID_ARGUMENT
    assert lib.mvfx_thread_set_options(7 << vfx.OPT_LUT_PLACEMENT_SHIFT) == 0   # placement 7 (round 5): round 4's per-wave windows
    assert lib.mvfx_thread_set_options(vfx.OPT_LUT_WG_WINDOW) == 0 and lib.mvfx_thread_options() == vfx.OPT_LUT_WG_WINDOW
    assert lib.mvfx_thread_set_options(0) == 0
    assert lib.mvfx_cube_lut_content_verdict(None, None) == 0                    # diagnostic accessor: no LUT, no verdict
    assert lib.mvfx_thread_options() == 0


def test_kernel_options_are_thread_local(vfx):
    """The calling thread is the context: an element on another streaming thread never sees this thread's word."""
    import threading
    lib = vfx.lib()
    seen = {}
    with vfx.options(nontemporal=True, typed=False, placement=3):
        assert lib.mvfx_thread_options() == (vfx.OPT_NONTEMPORAL | vfx.OPT_HSV_VALU_UNORM | (3 << vfx.OPT_LUT_PLACEMENT_SHIFT))

        def other():
            seen["before"] = lib.mvfx_thread_options()
            lib.mvfx_thread_set_options(vfx.OPT_HSV_LITERAL)
            seen["after"] = lib.mvfx_thread_options()
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"before": 0, "after": vfx.OPT_HSV_LITERAL}
        assert lib.mvfx_thread_options() & vfx.OPT_NONTEMPORAL
    assert lib.mvfx_thread_options() == 0
