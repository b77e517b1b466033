"""ctypes binding of libmi355vfx.so (the C ABI declared in include/mi355vfx.h).

This package is test/bench plumbing around the product: the product itself is the HIP
library under ``csrc/`` plus the C++ element layer under ``host/``.  Nothing here computes
pixels; every call goes through the C ABI and therefore through the HIP kernels.  There is
no CPU fallback: if ``libmi355vfx.so`` is missing, importing :func:`lib` raises.

The directory name (``gst-plugin-rs_amd``) is not a valid Python identifier, so the repo
root's ``_pkg.py`` loads this package under the module name ``gst_plugin_rs_amd``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_size_t,
                    c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVFX_LIB: another build of the same library (kernel A/B runs, tools/ab_bench.sh); never a different backend
LIB_PATH = os.environ.get("MVFX_LIB") or os.path.join(_HERE, "libmi355vfx.so")

# mvfx_format (include/mi355vfx.h)
FORMATS = {
    "RGBx": 0, "xRGB": 1, "BGRx": 2, "xBGR": 3, "RGBA": 4, "ARGB": 5, "BGRA": 6, "ABGR": 7,
    "RGB": 8, "BGR": 9, "RGBA64_LE": 10, "RGBA64_BE": 11, "I420": 12, "A420": 13, "RGB10A2_LE": 14, "NV12": 15,
}
FORMAT_NAMES = {v: k for k, v in FORMATS.items()}
BYTES_PER_PIXEL = {0: 4, 1: 4, 2: 4, 3: 4, 4: 4, 5: 4, 6: 4, 7: 4, 8: 3, 9: 3, 10: 8, 11: 8, 14: 4}

# mvfx_status
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_UNSUPPORTED_FORMAT = -2
ERR_NOT_NEGOTIATED = -3
ERR_DEVICE = -4
ERR_NO_DEVICE = -5
ERR_PARSE = -6
ERR_IO = -7
ERR_REFERENCE_PANIC = -8
ERR_NO_LUT = -9
ERR_OUT_OF_MEMORY = -10


# mvfx_thread_set_options bits
OPT_NONTEMPORAL = 0x01
OPT_HSV_LITERAL = 0x02
OPT_HSV_FORCE_FAST = 0x04
OPT_HSV_VALU_UNORM = 0x08
OPT_LUT_PLACEMENT_SHIFT = 4
OPT_SSIM_F64 = 0x80
OPT_LUT_WG_WINDOW = 0x100
OPT_DIRECT_DISPATCH = 0x200
OPT_DIRECT_ONLY = 0x400
OPT_DIRECT_UNORDERED = 0x800
ERR_DIRECT_UNAVAILABLE = -11


class options:
    """``with vfx.options(variant=1, typed=False): ...`` -- sets the calling thread's kernel options
    (mvfx_thread_set_options) for the block and restores the previous word.  variant: 0 auto, 1 literal,
    2 force strength-reduced; placement: colorlut LUT placement 0..6 (include/mi355vfx.h: 0 auto, 1 node layout in global/L2,
    2 LDS, 3 cell-packed global, 4 literal kernels, 5 tile kernel, 6 baked table of all 2^24 colours, 7 round 4's per-wave windows);
    wg_window: colorlut always through the workgroup-window kernel instead of asking the content probe; direct: one-frame hsvfilter calls may go
    out on the library's own queue (MVFX_OPT_DIRECT_DISPATCH: needs a completion event on the thread); unordered: colorlut's lane packets without the
    barrier bit (MVFX_OPT_DIRECT_UNORDERED)."""

    def __init__(self, variant=0, nontemporal=False, typed=True, placement=0, ssim_f64=False, wg_window=False, direct=False, unordered=False):
        self.word = ((OPT_DIRECT_DISPATCH if direct else 0) | (OPT_DIRECT_UNORDERED if unordered else 0) | (OPT_LUT_WG_WINDOW if wg_window else 0) | (OPT_SSIM_F64 if ssim_f64 else 0) | (OPT_NONTEMPORAL if nontemporal else 0) | (OPT_HSV_LITERAL if variant == 1 else 0) |
                     (OPT_HSV_FORCE_FAST if variant == 2 else 0) | (0 if typed else OPT_HSV_VALU_UNORM) |
                     (placement << OPT_LUT_PLACEMENT_SHIFT))

    def __enter__(self):
        self.prev = lib().mvfx_thread_options()
        check(lib().mvfx_thread_set_options(self.word))
        return self

    def __exit__(self, *exc):
        lib().mvfx_thread_set_options(self.prev)
        return False


class Frame(Structure):
    """struct mvfx_frame"""
    _fields_ = [("data", c_void_p), ("width", c_uint32), ("height", c_uint32),
                ("stride", c_uint32), ("format", c_int32)]


class PlanarFrame(Structure):
    """struct mvfx_planar_frame"""
    _fields_ = [("data", c_void_p * 4), ("stride", c_uint32 * 4), ("width", c_uint32),
                ("height", c_uint32), ("format", c_int32)]


class HsvFilterSettings(Structure):
    """struct mvfx_hsvfilter_settings == hsvfilter/imp.rs:32-39"""
    _fields_ = [("hue_shift", c_float), ("saturation_mul", c_float), ("saturation_off", c_float),
                ("value_mul", c_float), ("value_off", c_float)]

    @classmethod
    def default(cls):  # hsvfilter/imp.rs:25-29
        return cls(0.0, 1.0, 0.0, 1.0, 0.0)


class HsvDetectorSettings(Structure):
    """struct mvfx_hsvdetector_settings == hsvdetector/imp.rs:34-42"""
    _fields_ = [("hue_ref", c_float), ("hue_var", c_float), ("saturation_ref", c_float),
                ("saturation_var", c_float), ("value_ref", c_float), ("value_var", c_float)]

    @classmethod
    def default(cls):  # hsvdetector/imp.rs:26-31
        return cls(0.0, 10.0, 0.0, 0.15, 0.0, 0.3)


class MvfxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"mvfx status {status}: {message}")
        self.status = status
        self.message = message


_lib = None

# name -> (restype, argtypes); every symbol include/mi355vfx.h declares (tests/test_abi.py
# cross-checks this table against the header text)
SIGNATURES = {
    "mvfx_abi_version": (c_int, []),
    "mvfx_last_error": (c_char_p, []),
    "mvfx_status_string": (c_char_p, [c_int]),
    "mvfx_device_count": (c_int, []),
    "mvfx_set_device": (c_int, [c_int]),
    "mvfx_current_device": (c_int, []),
    "mvfx_stream_synchronize": (c_int, [c_void_p]),
    "mvfx_device_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "mvfx_device_free": (c_int, [c_void_p]),
    "mvfx_copy_to_device": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_copy_to_host": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_copy_device_to_device": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_thread_stream": (c_void_p, []),
    "mvfx_thread_stream_n": (c_void_p, [c_uint32]),
    "mvfx_event_create": (c_int, [POINTER(c_void_p)]),
    "mvfx_event_destroy": (c_int, [c_void_p]),
    "mvfx_event_record": (c_int, [c_void_p, c_void_p]),
    "mvfx_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "mvfx_event_synchronize": (c_int, [c_void_p]),
    "mvfx_event_query": (c_int, [c_void_p]),
    "mvfx_event_is_direct": (c_int, [c_void_p]),
    "mvfx_event_direct_queue": (c_int, [c_void_p]),
    "mvfx_direct_queue_of_stream": (c_int, [c_void_p]),
    "mvfx_direct_queue_wait_event": (c_int, [c_int, c_void_p]),
    "mvfx_direct_lane_park": (c_int, []),
    "mvfx_thread_set_completion_event": (c_int, [c_void_p]),
    "mvfx_thread_clear_completion_event": (c_int, []),
    "mvfx_host_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "mvfx_host_free": (c_int, [c_void_p]),
    "mvfx_copy_to_device_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_copy_to_host_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_copy_device_to_device_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvfx_thread_set_options": (c_int, [c_uint32]),
    "mvfx_thread_options": (c_uint32, []),
    "mvfx_hsvfilter_transform_frame_ip": (c_int, [POINTER(Frame), POINTER(HsvFilterSettings), c_void_p]),
    "mvfx_hsvfilter_transform_frames_ip": (c_int, [POINTER(Frame), c_uint32, POINTER(HsvFilterSettings), c_void_p]),
    "mvfx_hsvfilter_transform_frames_ip_settings": (c_int, [POINTER(Frame), c_uint32, POINTER(HsvFilterSettings), c_void_p]),
    "mvfx_hsvfilter_transform_frame_ip_combined": (c_int, [POINTER(Frame), POINTER(HsvFilterSettings), c_void_p]),
    "mvfx_hsvfilter_transform_frame_ip_fenced": (c_int, [POINTER(Frame), POINTER(HsvFilterSettings), c_void_p, POINTER(c_void_p)]),
    "mvfx_combiner_stats": (c_int, [c_int, POINTER(c_uint64), POINTER(c_uint64)]),
    "mvfx_combiner_average_wait_us": (ctypes.c_double, [c_int]),
    "mvfx_hsvfilter_transform_frame_ip_host": (c_int, [POINTER(Frame), POINTER(HsvFilterSettings)]),
    "mvfx_hsvdetector_transform_frame": (c_int, [POINTER(Frame), POINTER(Frame), POINTER(HsvDetectorSettings), c_void_p]),
    "mvfx_hsvdetector_transform_frames": (c_int, [POINTER(Frame), POINTER(Frame), c_uint32, POINTER(HsvDetectorSettings), c_void_p]),
    "mvfx_hsvdetector_transform_frame_host": (c_int, [POINTER(Frame), POINTER(Frame), POINTER(HsvDetectorSettings)]),
    "mvfx_hsv_from_frame": (c_int, [POINTER(Frame), c_void_p, c_void_p]),
    "mvfx_selftest_typed_unorm8": (c_int, [POINTER(c_uint32), POINTER(c_uint32)]),
    "mvfx_cube_lut_parse": (c_int, [c_char_p, c_size_t, POINTER(c_void_p)]),
    "mvfx_cube_lut_parse_file": (c_int, [c_char_p, POINTER(c_void_p)]),
    "mvfx_cube_lut_free": (None, [c_void_p]),
    "mvfx_cube_lut_is_3d": (c_int, [c_void_p]),
    "mvfx_cube_lut_size": (c_uint32, [c_void_p]),
    "mvfx_cube_lut_content_verdict": (c_int, [c_void_p, POINTER(c_uint32)]),
    "mvfx_cube_lut_device_copies": (c_int, [c_void_p]),
    "mvfx_cube_lut_domain": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float)]),
    "mvfx_cube_lut_rgba": (POINTER(c_float), [c_void_p]),
    "mvfx_cube_lut_table_1d": (POINTER(c_float), [c_void_p, c_int]),
    "mvfx_colorlut_transform_frame": (c_int, [c_void_p, POINTER(Frame), POINTER(Frame), c_void_p]),
    "mvfx_colorlut_transform_frames": (c_int, [c_void_p, POINTER(Frame), POINTER(Frame), c_uint32, c_void_p]),
    "mvfx_colorlut_transform_frame_host": (c_int, [c_void_p, POINTER(Frame), POINTER(Frame)]),
    "mvfx_cube_lut_write": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "mvfx_free_text": (None, [c_void_p]),
    "mvfx_overlay_blend": (c_int, [POINTER(Frame), POINTER(Frame), c_int32, c_int32, c_float, c_void_p]),
    "mvfx_overlay_blend_host": (c_int, [POINTER(Frame), POINTER(Frame), c_int32, c_int32, c_float]),
    "mvfx_colordetect_histogram": (c_int, [POINTER(Frame), c_uint32, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "mvfx_colordetect_histogram_frames": (c_int, [POINTER(Frame), c_uint32, c_uint32, c_void_p, c_void_p]),
    "mvfx_mmcq_palette_from_histogram": (c_int, [c_void_p, POINTER(c_uint32), c_uint32, POINTER(c_uint32), POINTER(c_uint32)]),
    "mvfx_colordetect_palette": (c_int, [POINTER(Frame), c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_uint32), c_void_p]),
    "mvfx_colordetect_palette_host": (c_int, [POINTER(Frame), c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_uint32)]),
    "mvfx_css_color_similar": (c_char_p, [ctypes.c_uint8, ctypes.c_uint8, ctypes.c_uint8]),
    "mvfx_blockhash_sums": (c_int, [POINTER(Frame), c_uint32, c_uint32, c_void_p, c_void_p]),
    "mvfx_blockhash_sums_band": (c_int, [POINTER(Frame), c_uint32, c_uint32, c_void_p, c_void_p]),
    "mvfx_blockhash_sums_pads": (c_int, [POINTER(Frame), c_uint32, c_uint32, c_uint32, c_void_p, c_void_p]),
    "mvfx_blockhash_bits": (c_int, [POINTER(c_uint32), c_uint32, c_uint32, POINTER(c_uint64)]),
    "mvfx_hash_distance": (c_uint32, [c_uint64, c_uint64]),
    "mvfx_blockhash": (c_int, [POINTER(Frame), POINTER(c_uint64), c_void_p]),
    "mvfx_blockhash_host": (c_int, [POINTER(Frame), POINTER(c_uint64)]),
    "mvfx_videocompare_distance": (c_int, [POINTER(Frame), POINTER(Frame), POINTER(ctypes.c_double), c_void_p]),
    "mvfx_comm_unique_id": (c_int, [c_void_p]),
    "mvfx_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "mvfx_comm_destroy": (c_int, [c_void_p]),
    "mvfx_comm_rank": (c_int, [c_void_p]),
    "mvfx_comm_world": (c_int, [c_void_p]),
    "mvfx_comm_allreduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_void_p]),
    "mvfx_videocompare_sharded_distances": (c_int, [c_void_p, POINTER(Frame), c_uint32, c_uint32, c_uint32, POINTER(ctypes.c_double),
                                                    POINTER(c_uint64), c_void_p]),
    "mvfx_videocompare_sharded_dssim": (c_int, [c_void_p, POINTER(Frame), POINTER(Frame), c_uint32, c_uint32, POINTER(ctypes.c_double), c_void_p]),
    "mvfx_image_hash": (c_int, [POINTER(Frame), ctypes.c_int32, POINTER(c_uint64), POINTER(c_uint32), c_void_p]),
    "mvfx_image_hash_host": (c_int, [POINTER(Frame), ctypes.c_int32, POINTER(c_uint64), POINTER(c_uint32)]),
    "mvfx_image_gray_resize_lanczos3": (c_int, [POINTER(Frame), c_uint32, c_uint32, c_void_p, c_void_p]),
    "mvfx_videocompare_distance_algo": (c_int, [POINTER(Frame), POINTER(Frame), ctypes.c_int32, POINTER(ctypes.c_double), c_void_p]),
    "mvfx_ssim_partial_sums": (c_int, [POINTER(Frame), POINTER(Frame), c_uint32, c_uint32, POINTER(ctypes.c_double),
                                       POINTER(ctypes.c_double), POINTER(c_uint32), c_void_p]),
    "mvfx_ssim_partial_deviation": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_void_p]),
    "mvfx_ssim_combine": (ctypes.c_double, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_uint32]),
    "mvfx_ssim_distance": (c_int, [POINTER(Frame), POINTER(Frame), POINTER(ctypes.c_double), c_void_p]),
    "mvfx_ssim_distance_host": (c_int, [POINTER(Frame), POINTER(Frame), POINTER(ctypes.c_double)]),
    "mvfx_roundedcorners_mask": (c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]),
    "mvfx_roundedcorners_mask_host": (c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_uint32]),
    "mvfx_roundedcorners_cairo_version": (c_char_p, []),
    "mvfx_convert_i420_to_rgba": (c_int, [POINTER(PlanarFrame), POINTER(Frame), ctypes.c_int32, c_void_p]),
    "mvfx_convert_rgba_to_i420": (c_int, [POINTER(Frame), POINTER(PlanarFrame), ctypes.c_int32, c_void_p]),
    "mvfx_convert_nv12_to_rgba": (c_int, [POINTER(PlanarFrame), POINTER(Frame), ctypes.c_int32, c_void_p]),
    "mvfx_convert_rgba_to_nv12": (c_int, [POINTER(Frame), POINTER(PlanarFrame), ctypes.c_int32, c_void_p]),
    "mvfx_hsvfilter_transform_i420": (c_int, [POINTER(PlanarFrame), POINTER(PlanarFrame), POINTER(HsvFilterSettings), ctypes.c_int32, c_void_p]),
    "mvfx_convert_i420_to_rgba_frames": (c_int, [POINTER(PlanarFrame), POINTER(Frame), c_uint32, ctypes.c_int32, c_void_p]),
    "mvfx_convert_rgba_to_i420_frames": (c_int, [POINTER(Frame), POINTER(PlanarFrame), c_uint32, ctypes.c_int32, c_void_p]),
    "mvfx_hsvdetector_transform_i420": (c_int, [POINTER(PlanarFrame), POINTER(Frame), POINTER(HsvDetectorSettings), ctypes.c_int32, c_void_p]),
    "mvfx_colorlut_transform_i420": (c_int, [c_void_p, POINTER(PlanarFrame), POINTER(PlanarFrame), ctypes.c_int32, c_void_p]),
    "mvfx_roundedcorners_compose_a420": (c_int, [POINTER(PlanarFrame), c_void_p, c_uint32, POINTER(PlanarFrame), c_void_p]),
}


def lib() -> ctypes.CDLL:
    """Loads libmi355vfx.so (built in-tree by `make -C gst-plugin-rs_amd`); fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() or "
                "`make -C gst-plugin-rs_amd`. There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int) -> int:
    if status != OK:
        raise MvfxError(status, lib().mvfx_last_error().decode("utf-8", "replace"))
    return status


def last_error() -> str:
    return lib().mvfx_last_error().decode("utf-8", "replace")


def make_frame(ptr: int, width: int, height: int, stride: int, fmt) -> Frame:
    f = FORMATS[fmt] if isinstance(fmt, str) else int(fmt)
    return Frame(c_void_p(ptr), width, height, stride, f)


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through the C ABI (no torch needed)."""

    def __init__(self, nbytes: int):
        p = c_void_p()
        check(lib().mvfx_device_alloc(ctypes.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, host) -> "DeviceBuffer":
        import numpy as np
        a = np.ascontiguousarray(host)
        assert a.nbytes <= self.nbytes
        check(lib().mvfx_copy_to_device(c_void_p(self.ptr), a.ctypes.data_as(c_void_p), a.nbytes, None))
        return self

    def download(self, nbytes: int | None = None, dtype="uint8"):
        import numpy as np
        n = self.nbytes if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        check(lib().mvfx_copy_to_host(out.ctypes.data_as(c_void_p), c_void_p(self.ptr), n, None))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            lib().mvfx_device_free(c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- thin, numpy-friendly wrappers used by tests / smoke / bench ---------------------------

def hsvfilter_host(frame_bytes, width, height, stride, fmt, settings: HsvFilterSettings):
    """In-place hsvfilter on a host numpy buffer through the *_host entry point."""
    import numpy as np
    assert frame_bytes.dtype == np.uint8 and frame_bytes.flags["C_CONTIGUOUS"]
    f = make_frame(frame_bytes.ctypes.data, width, height, stride, fmt)
    check(lib().mvfx_hsvfilter_transform_frame_ip_host(ctypes.byref(f), ctypes.byref(settings)))
    return frame_bytes


def hsvfilter_device(ptr, width, height, stride, fmt, settings: HsvFilterSettings, stream=None):
    f = make_frame(ptr, width, height, stride, fmt)
    check(lib().mvfx_hsvfilter_transform_frame_ip(ctypes.byref(f), ctypes.byref(settings), stream))


def hsvfilter_device_batch(ptrs, width, height, stride, fmt, settings: HsvFilterSettings, stream=None):
    arr = (Frame * len(ptrs))(*[make_frame(p, width, height, stride, fmt) for p in ptrs])
    check(lib().mvfx_hsvfilter_transform_frames_ip(arr, len(ptrs), ctypes.byref(settings), stream))


def hsvdetector_host(in_bytes, in_stride, in_fmt, out_bytes, out_stride, out_fmt, width, height,
                     settings: HsvDetectorSettings):
    fi = make_frame(in_bytes.ctypes.data, width, height, in_stride, in_fmt)
    fo = make_frame(out_bytes.ctypes.data, width, height, out_stride, out_fmt)
    check(lib().mvfx_hsvdetector_transform_frame_host(ctypes.byref(fi), ctypes.byref(fo), ctypes.byref(settings)))
    return out_bytes


class CubeLut:
    """mvfx_cube_lut handle (colorlut's parsed .cube); parse errors raise MvfxError."""

    def __init__(self, text=None, path=None):
        h = c_void_p()
        if path is not None:
            check(lib().mvfx_cube_lut_parse_file(os.fsencode(path), ctypes.byref(h)))
        else:
            raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
            check(lib().mvfx_cube_lut_parse(raw, len(raw), ctypes.byref(h)))
        self.h = h

    @property
    def is_3d(self):
        return bool(lib().mvfx_cube_lut_is_3d(self.h))

    @property
    def size(self):
        return lib().mvfx_cube_lut_size(self.h)

    def domain(self):
        import numpy as np
        sc = (c_float * 3)()
        of = (c_float * 3)()
        check(lib().mvfx_cube_lut_domain(self.h, sc, of))
        return np.array(list(sc), dtype=np.float32), np.array(list(of), dtype=np.float32)

    def rgba(self):
        import numpy as np
        p = lib().mvfx_cube_lut_rgba(self.h)
        return np.ctypeslib.as_array(p, shape=(self.size ** 3 * 4,)).reshape(-1, 4).copy()

    def table(self, c):
        import numpy as np
        p = lib().mvfx_cube_lut_table_1d(self.h, c)
        return np.ctypeslib.as_array(p, shape=(self.size,)).copy()

    def apply_host(self, src, src_stride, dst, dst_stride, width, height, fmt):
        fi = make_frame(src.ctypes.data, width, height, src_stride, fmt)
        fo = make_frame(dst.ctypes.data, width, height, dst_stride, fmt)
        check(lib().mvfx_colorlut_transform_frame_host(self.h, ctypes.byref(fi), ctypes.byref(fo)))
        return dst

    def apply_device(self, src_ptr, src_stride, dst_ptr, dst_stride, width, height, fmt, stream=None):
        fi = make_frame(src_ptr, width, height, src_stride, fmt)
        fo = make_frame(dst_ptr, width, height, dst_stride, fmt)
        check(lib().mvfx_colorlut_transform_frame(self.h, ctypes.byref(fi), ctypes.byref(fo), stream))

    def write(self) -> str:
        """mvfx_cube_lut_write: the LUT as .cube text (round-trips bit for bit through the parser)"""
        text = c_void_p()
        n = c_size_t()
        check(lib().mvfx_cube_lut_write(self.h, ctypes.byref(text), ctypes.byref(n)))
        try:
            return ctypes.string_at(text.value, n.value).decode("utf-8")
        finally:
            lib().mvfx_free_text(text)

    def free(self):
        if getattr(self, "h", None):
            lib().mvfx_cube_lut_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


ALL_SAMPLES = (1 << 64) - 1
COLORDETECT_RECORD_WORDS = 32768 + 8


def colordetect_palette_host(frame_bytes, width, height, stride, fmt, quality=10, max_colors=2):
    """-> (palette [0xRRGGBB...], dominant css name) like ColorDetect::detect_color (imp.rs:57-86)"""
    f = make_frame(frame_bytes.ctypes.data, width, height, stride, fmt)
    pal = (c_uint32 * 256)()
    n = c_uint32()
    check(lib().mvfx_colordetect_palette_host(ctypes.byref(f), quality, max_colors, pal, ctypes.byref(n)))
    palette = [pal[i] for i in range(n.value)]
    p0 = palette[0]
    name = lib().mvfx_css_color_similar((p0 >> 16) & 255, (p0 >> 8) & 255, p0 & 255).decode()
    return palette, name


def blockhash_host(frame_bytes, width, height, stride, fmt):
    f = make_frame(frame_bytes.ctypes.data, width, height, stride, fmt)
    h = c_uint64()
    check(lib().mvfx_blockhash_host(ctypes.byref(f), ctypes.byref(h)))
    return h.value


def make_i420(base_ptr: int, width: int, height: int, y_stride: int, c_stride: int, u_offset: int, v_offset: int) -> PlanarFrame:
    """mvfx_planar_frame view of one I420 frame inside a device buffer"""
    f = PlanarFrame()
    f.data[0], f.data[1], f.data[2] = base_ptr, base_ptr + u_offset, base_ptr + v_offset
    f.stride[0], f.stride[1], f.stride[2] = y_stride, c_stride, c_stride
    f.width, f.height, f.format = width, height, FORMATS["I420"]
    return f


DTYPE_U32, DTYPE_U64, DTYPE_F64 = 0, 1, 2
REDUCE_SUM, REDUCE_MIN, REDUCE_MAX = 0, 1, 2


class Comm:
    """mvfx_comm: the library's own RCCL communicator (one process per GPU).  `broadcast_id(id_bytes_or_None) -> bytes` hands rank 0's
    128-byte id to every rank (torch.distributed.broadcast_object_list in the bench, a file in the tests)."""

    def __init__(self, rank: int, world: int, broadcast_id):
        ident = None
        if rank == 0:
            buf = (ctypes.c_uint8 * 128)()
            check(lib().mvfx_comm_unique_id(buf))
            ident = bytes(buf)
        ident = broadcast_id(ident)
        h = c_void_p()
        check(lib().mvfx_comm_create((ctypes.c_uint8 * 128).from_buffer_copy(ident), rank, world, ctypes.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    def allreduce(self, ptr: int, count: int, dtype=DTYPE_U32, op=REDUCE_SUM, stream=None):
        check(lib().mvfx_comm_allreduce(self.h, c_void_p(ptr), count, dtype, op, stream))

    def destroy(self):
        if getattr(self, "h", None):
            lib().mvfx_comm_destroy(self.h)
            self.h = None


def videocompare_sharded_distances(comm, bands, full_height, band_first_row, stream=None, want_hashes=False):
    """bands: ctypes array of Frame, this rank's rows of every pad's frame -> distances of pads 1.. to pad 0 (and the hashes)"""
    n = len(bands)
    out = (ctypes.c_double * max(n - 1, 1))()
    hashes = (c_uint64 * n)() if want_hashes else None
    check(lib().mvfx_videocompare_sharded_distances(comm.h if comm is not None else None, bands, n, full_height, band_first_row, out,
                                                    hashes, stream))
    d = [out[i] for i in range(n - 1)]
    return (d, [hashes[i] for i in range(n)]) if want_hashes else d


def videocompare_sharded_dssim(comm, frame_a, frame_b, row_begin, row_end, stream=None):
    """hash-algo=dssim of a pair whose rows are shared out over the ranks of `comm` (None: one GPU): this rank maps rows
    [row_begin, row_end) of the whole frames it holds; both all-reduces happen inside the library (RCCL)."""
    out = ctypes.c_double()
    check(lib().mvfx_videocompare_sharded_dssim(comm.h if comm is not None else None, ctypes.byref(frame_a), ctypes.byref(frame_b), row_begin,
                                                row_end, ctypes.byref(out), stream))
    return out.value


HASH_ALGOS = {"mean": 0, "gradient": 1, "vertgradient": 2, "doublegradient": 3, "blockhash": 4, "dssim": 5}


def image_hash_host(frame_bytes, width, height, stride, fmt, algo):
    """hash-algo mean / gradient / vertgradient / doublegradient of a host frame -> (hash bits, n_bits)"""
    f = make_frame(frame_bytes.ctypes.data, width, height, stride, fmt)
    h = c_uint64()
    n = c_uint32()
    check(lib().mvfx_image_hash_host(ctypes.byref(f), HASH_ALGOS[algo], ctypes.byref(h), ctypes.byref(n)))
    return h.value, n.value


def ssim_distance_host(a_bytes, b_bytes, width, height, stride_a, stride_b, fmt):
    """`hash-algo=dssim` distance of two host frames (videocompare/hashed_image.rs:49-59,72-75)."""
    a = (ctypes.c_uint8 * len(a_bytes)).from_buffer_copy(bytes(a_bytes))
    b = (ctypes.c_uint8 * len(b_bytes)).from_buffer_copy(bytes(b_bytes))
    fa = make_frame(ctypes.addressof(a), width, height, stride_a, fmt)
    fb = make_frame(ctypes.addressof(b), width, height, stride_b, fmt)
    out = ctypes.c_double(0.0)
    check(lib().mvfx_ssim_distance_host(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(out)))
    return out.value


def ssim_partial_sums(frame_a: Frame, frame_b: Frame, row_begin: int, row_end: int, stream=None):
    """Pass 1 of the shardable SSIM distance: (sums[5], counts[5], n_scales) of rows [row_begin,row_end)."""
    sums = (ctypes.c_double * 5)()
    counts = (ctypes.c_double * 5)()
    n = c_uint32(0)
    check(lib().mvfx_ssim_partial_sums(ctypes.byref(frame_a), ctypes.byref(frame_b), row_begin, row_end, sums, counts,
                                       ctypes.byref(n), stream))
    return list(sums), list(counts), n.value


def ssim_partial_deviation(mean, stream=None):
    """Pass 2: per-scale sum of |map - mean| over the band given to `ssim_partial_sums` on this thread."""
    m = (ctypes.c_double * 5)(*mean)
    out = (ctypes.c_double * 5)()
    check(lib().mvfx_ssim_partial_deviation(m, out, stream))
    return list(out)


def ssim_combine(mean, mad, n_scales):
    return lib().mvfx_ssim_combine((ctypes.c_double * 5)(*mean), (ctypes.c_double * 5)(*mad), n_scales)
