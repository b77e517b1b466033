"""ctypes binding of oracle/liboracle.so -- the CPU checker.  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32,
                    c_uint64, c_void_p)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "liboracle.so")

FORMATS = {
    "RGBx": 0, "xRGB": 1, "BGRx": 2, "xBGR": 3, "RGBA": 4, "ARGB": 5, "BGRA": 6, "ABGR": 7,
    "RGB": 8, "BGR": 9, "RGBA64_LE": 10, "RGBA64_BE": 11, "I420": 12, "A420": 13, "RGB10A2_LE": 14, "NV12": 15,
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(SO)
        u8p = POINTER(c_uint8)
        f32p = POINTER(c_float)
        L.orc_hsv_from_rgb.argtypes = [u8p, f32p]
        L.orc_hsv_from_bgr.argtypes = [u8p, f32p]
        L.orc_hsv_to_rgb.argtypes = [f32p, u8p]
        L.orc_hsv_to_bgr.argtypes = [f32p, u8p]
        L.orc_hsvfilter_transform_frame_ip.argtypes = [c_void_p, c_size_t, c_uint32, c_uint32, c_int, f32p]
        L.orc_hsvfilter_transform_frame_ip.restype = c_int
        L.orc_hsvdetector_transform_frame.argtypes = [c_void_p, c_size_t, c_uint32, c_int, c_void_p,
                                                      c_size_t, c_uint32, c_int, c_uint32, f32p]
        L.orc_hsvdetector_transform_frame.restype = c_int
        L.orc_hsv_from_rgb_frame.argtypes = [c_void_p, c_size_t, c_void_p]
        L.orc_cube_parse.argtypes = [c_char_p, c_size_t, c_char_p, c_size_t]
        L.orc_cube_parse.restype = c_void_p
        L.orc_cube_free.argtypes = [c_void_p]
        L.orc_cube_is_3d.argtypes = [c_void_p]
        L.orc_cube_is_3d.restype = c_int
        L.orc_cube_size.argtypes = [c_void_p]
        L.orc_cube_size.restype = c_uint32
        for n in ("orc_cube_domain_scale", "orc_cube_domain_offset", "orc_cube_rgba"):
            getattr(L, n).argtypes = [c_void_p]
            getattr(L, n).restype = f32p
        L.orc_cube_table_1d.argtypes = [c_void_p, c_int]
        L.orc_cube_table_1d.restype = f32p
        L.orc_colorlut_transform_frame.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p,
                                                   c_size_t, c_uint32, c_uint32, c_uint32, c_int]
        L.orc_colorlut_transform_frame.restype = c_int
        L.orc_colordetect_histogram.argtypes = [c_void_p, c_size_t, c_int, c_uint32, c_void_p,
                                                POINTER(c_uint32), POINTER(c_uint64)]
        L.orc_colordetect_histogram.restype = c_int
        L.orc_colordetect_palette.argtypes = [c_void_p, c_size_t, c_int, c_uint32, c_uint32, POINTER(c_uint32)]
        L.orc_colordetect_palette.restype = c_int
        L.orc_mmcq_from_histogram.argtypes = [c_void_p, POINTER(c_uint32), c_uint32, POINTER(c_uint32)]
        L.orc_mmcq_from_histogram.restype = c_int
        L.orc_css_color_similar.argtypes = [c_uint8, c_uint8, c_uint8]
        L.orc_css_color_similar.restype = c_char_p
        L.orc_blockhash_sums.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, POINTER(c_uint32)]
        L.orc_blockhash_sums.restype = c_int
        L.orc_blockhash_bits.argtypes = [POINTER(c_uint32), c_uint32, c_uint32]
        L.orc_blockhash_bits.restype = c_uint64
        L.orc_blockhash.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, POINTER(c_uint64)]
        L.orc_blockhash.restype = c_int
        L.orc_gray_resize_lanczos3.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_uint32, c_uint32, c_void_p]
        L.orc_gray_resize_lanczos3.restype = c_int
        L.orc_image_hash.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_int, POINTER(c_uint64), POINTER(c_uint32)]
        L.orc_image_hash.restype = c_int
        L.orc_convert_i420_to_rgba.argtypes = [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_uint32]
        L.orc_convert_i420_to_rgba.restype = c_int
        L.orc_convert_rgba_to_i420.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_uint32]
        L.orc_convert_rgba_to_i420.restype = c_int
        L.orc_convert_rgba_to_nv12.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_void_p, c_uint32, c_uint32]
        L.orc_convert_rgba_to_nv12.restype = c_int
        L.orc_convert_nv12_to_rgba.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_uint32]
        L.orc_convert_nv12_to_rgba.restype = c_int
        L.orc_hamming64.argtypes = [c_uint64, c_uint64]
        L.orc_hamming64.restype = c_uint32
        L.orc_ssim_distance.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_int,
                                        POINTER(ctypes.c_double), POINTER(ctypes.c_double)]
        L.orc_ssim_distance.restype = c_int
        L.orc_ssim_band.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_uint32, c_uint32,
                                    POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                    POINTER(c_int)]
        L.orc_ssim_band.restype = c_int
        L.orc_ssim_combine.argtypes = [POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_int]
        L.orc_ssim_combine.restype = ctypes.c_double
        _lib = L
    return _lib


def _fmt(f):
    return FORMATS[f] if isinstance(f, str) else int(f)


def from_rgb(rgb, bgr=False):
    a = (c_uint8 * 3)(*rgb)
    o = (c_float * 3)()
    (lib().orc_hsv_from_bgr if bgr else lib().orc_hsv_from_rgb)(a, o)
    return np.array(list(o), dtype=np.float32)


def to_rgb(hsv, bgr=False):
    a = (c_float * 3)(*[float(x) for x in hsv])
    o = (c_uint8 * 3)()
    (lib().orc_hsv_to_bgr if bgr else lib().orc_hsv_to_rgb)(a, o)
    return tuple(o)


def hsvfilter(frame: np.ndarray, width, stride, fmt, settings):
    """In place on a C-contiguous uint8 array holding stride*height bytes. Returns status."""
    assert frame.dtype == np.uint8 and frame.flags["C_CONTIGUOUS"]
    s = (c_float * 5)(*[float(x) for x in settings])
    return lib().orc_hsvfilter_transform_frame_ip(frame.ctypes.data, frame.nbytes, width, stride, _fmt(fmt), s)


def hsvdetector(inp: np.ndarray, in_stride, in_fmt, out: np.ndarray, out_stride, out_fmt, width, settings):
    s = (c_float * 6)(*[float(x) for x in settings])
    return lib().orc_hsvdetector_transform_frame(inp.ctypes.data, inp.nbytes, in_stride, _fmt(in_fmt),
                                                 out.ctypes.data, out.nbytes, out_stride, _fmt(out_fmt),
                                                 width, s)


def hsv_from_rgbx(rgbx: np.ndarray):
    n = rgbx.nbytes // 4
    out = np.empty((n, 3), dtype=np.float32)
    lib().orc_hsv_from_rgb_frame(rgbx.ctypes.data, n, out.ctypes.data)
    return out


class CubeLut:
    def __init__(self, text):
        if isinstance(text, str):
            text = text.encode("utf-8")
        err = ctypes.create_string_buffer(256)
        self.h = lib().orc_cube_parse(text, len(text), err, 256)
        self.error = err.value.decode() if not self.h else None

    @property
    def ok(self):
        return bool(self.h)

    @property
    def is_3d(self):
        return bool(lib().orc_cube_is_3d(self.h))

    @property
    def size(self):
        return lib().orc_cube_size(self.h)

    @property
    def domain_scale(self):
        p = lib().orc_cube_domain_scale(self.h)
        return np.array([p[i] for i in range(3)], dtype=np.float32)

    @property
    def domain_offset(self):
        p = lib().orc_cube_domain_offset(self.h)
        return np.array([p[i] for i in range(3)], dtype=np.float32)

    def rgba(self):
        n = self.size ** 3 * 4
        return np.ctypeslib.as_array(lib().orc_cube_rgba(self.h), shape=(n,)).reshape(-1, 4).copy()

    def table(self, c):
        return np.ctypeslib.as_array(lib().orc_cube_table_1d(self.h, c), shape=(self.size,)).copy()

    def apply(self, src: np.ndarray, src_stride, dst: np.ndarray, dst_stride, width, height, fmt):
        return lib().orc_colorlut_transform_frame(self.h, src.ctypes.data, src.nbytes, src_stride,
                                                  dst.ctypes.data, dst.nbytes, dst_stride, width, height, _fmt(fmt))

    def __del__(self):
        try:
            if self.h:
                lib().orc_cube_free(self.h)
        except Exception:
            pass


def colordetect_histogram(pixels: np.ndarray, fmt, quality):
    hist = np.zeros(32768, dtype=np.int32)
    mm = (c_uint32 * 6)()
    n = c_uint64()
    rc = lib().orc_colordetect_histogram(pixels.ctypes.data, pixels.nbytes, _fmt(fmt), quality,
                                         hist.ctypes.data, mm, ctypes.byref(n))
    return rc, hist, list(mm), n.value


def colordetect_palette(pixels: np.ndarray, fmt, quality, max_colors):
    out = (c_uint32 * 256)()
    rc = lib().orc_colordetect_palette(pixels.ctypes.data, pixels.nbytes, _fmt(fmt), quality, max_colors, out)
    return rc, [out[i] for i in range(max(rc, 0))]


def mmcq_from_histogram(hist: np.ndarray, minmax, max_colors):
    out = (c_uint32 * 256)()
    mm = (c_uint32 * 6)(*minmax)
    h = np.ascontiguousarray(hist, dtype=np.int32)
    rc = lib().orc_mmcq_from_histogram(h.ctypes.data, mm, max_colors, out)
    return rc, [out[i] for i in range(max(rc, 0))]


def css_similar(r, g, b):
    return lib().orc_css_color_similar(r, g, b).decode()


def blockhash_sums(data: np.ndarray, width, height, stride, fmt):
    s = (c_uint32 * 64)()
    rc = lib().orc_blockhash_sums(data.ctypes.data, width, height, stride, _fmt(fmt), s)
    return rc, np.array(list(s), dtype=np.uint32)


def blockhash_bits(sums, width, height):
    s = (c_uint32 * 64)(*[int(x) for x in sums])
    return lib().orc_blockhash_bits(s, width, height)


def blockhash(data: np.ndarray, width, height, stride, fmt):
    h = c_uint64()
    rc = lib().orc_blockhash(data.ctypes.data, width, height, stride, _fmt(fmt), ctypes.byref(h))
    return rc, h.value


HASH_ALGOS = {"mean": 0, "gradient": 1, "vertgradient": 2, "doublegradient": 3, "blockhash": 4, "dssim": 5}


def gray_resize_lanczos3(data: np.ndarray, width, height, stride, fmt, nw, nh):
    out = np.zeros((nh, nw), np.uint8)
    rc = lib().orc_gray_resize_lanczos3(data.ctypes.data, width, height, stride, _fmt(fmt), nw, nh, out.ctypes.data)
    return rc, out


def image_hash(data: np.ndarray, width, height, stride, fmt, algo):
    """(rc, hash bits as int, n_bits) for algo in mean / gradient / vertgradient / doublegradient"""
    h = c_uint64()
    n = c_uint32()
    rc = lib().orc_image_hash(data.ctypes.data, width, height, stride, _fmt(fmt), HASH_ALGOS[algo], ctypes.byref(h), ctypes.byref(n))
    return rc, h.value, n.value


def i420_layout(w, h):
    """GstVideoInfo layout of I420: (y stride, chroma stride, y rows, chroma rows, u offset, v offset, size)"""
    ru = lambda v, a: (v + a - 1) // a * a
    ys, cs = ru(w, 4), ru(ru(w, 2) // 2, 4)
    yr, cr = ru(h, 2), ru(h, 2) // 2
    return ys, cs, yr, cr, ys * yr, ys * yr + cs * cr, ys * yr + 2 * cs * cr


def convert_i420_to_rgba(raw: np.ndarray, w, h, standard=0):
    """raw: one I420 frame in the GstVideoInfo layout -> (rc, h x w*4 RGBA)"""
    ys, cs, yr, cr, uo, vo, size = i420_layout(w, h)
    assert raw.size >= size
    out = np.zeros((h, w * 4), np.uint8)
    base = raw.ctypes.data
    rc = lib().orc_convert_i420_to_rgba(base, base + uo, base + vo, ys, cs, cs, w, h, standard, out.ctypes.data, w * 4)
    return rc, out


def convert_rgba_to_i420(px: np.ndarray, w, h, stride, standard=0):
    """-> (rc, Y[h,w], U[RU2(h)/2, RU2(w)/2], V[...]) tightly packed"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    Y = np.zeros((h, w), np.uint8)
    U = np.zeros((ch, cw), np.uint8)
    V = np.zeros((ch, cw), np.uint8)
    rc = lib().orc_convert_rgba_to_i420(px.ctypes.data, stride, w, h, standard, Y.ctypes.data, U.ctypes.data, V.ctypes.data, w, cw, cw)
    return rc, Y, U, V


def convert_rgba_to_nv12(px: np.ndarray, w, h, stride, standard=0):
    """-> (rc, Y[h,w], UV[RU2(h)/2, 2 * RU2(w)/2]) tightly packed"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    Y = np.zeros((h, w), np.uint8)
    UV = np.zeros((ch, 2 * cw), np.uint8)
    rc = lib().orc_convert_rgba_to_nv12(px.ctypes.data, stride, w, h, standard, Y.ctypes.data, UV.ctypes.data, w, 2 * cw)
    return rc, Y, UV


def nv12_layout(w, h):
    """GstVideoInfo layout of NV12: (y stride, uv stride, y rows, uv rows, uv offset, size)"""
    ru = lambda v, a: (v + a - 1) // a * a
    ys, uvs, yr = ru(w, 4), ru(ru(w, 2), 4), ru(h, 2)
    return ys, uvs, yr, yr // 2, ys * yr, ys * yr + uvs * (yr // 2)


def convert_nv12_to_rgba(raw: np.ndarray, w, h, standard=0):
    """raw: one NV12 frame in the GstVideoInfo layout -> (rc, h x w*4 RGBA)"""
    ys, uvs, yr, cr, uvo, size = nv12_layout(w, h)
    assert raw.size >= size
    out = np.zeros((h, w * 4), np.uint8)
    base = raw.ctypes.data
    rc = lib().orc_convert_nv12_to_rgba(base, base + uvo, ys, uvs, w, h, standard, out.ctypes.data, w * 4)
    return rc, out


def hamming(a, b):
    return lib().orc_hamming64(a, b)


def ssim_distance(a: np.ndarray, b: np.ndarray, width, height, stride_a, stride_b, fmt):
    d = ctypes.c_double()
    per = (ctypes.c_double * 5)()
    rc = lib().orc_ssim_distance(a.ctypes.data, b.ctypes.data, width, height, stride_a, stride_b, _fmt(fmt), ctypes.byref(d), per)
    return rc, d.value, [per[i] for i in range(5)]


def ssim_band(a: np.ndarray, b: np.ndarray, width, height, stride_a, stride_b, fmt, row_begin, row_end, mean=None):
    sums = (ctypes.c_double * 5)()
    counts = (ctypes.c_double * 5)()
    n = c_int(0)
    m = (ctypes.c_double * 5)(*mean) if mean is not None else None
    rc = lib().orc_ssim_band(a.ctypes.data, b.ctypes.data, width, height, stride_a, stride_b, _fmt(fmt), row_begin, row_end,
                             m, sums, counts, ctypes.byref(n))
    assert rc == 0, rc
    return list(sums), list(counts), n.value


def ssim_combine(mean, mad, n_scales):
    return lib().orc_ssim_combine((ctypes.c_double * 5)(*mean), (ctypes.c_double * 5)(*mad), n_scales)


def overlay_blend(dest: np.ndarray, width, height, stride, fmt, overlay: np.ndarray, ow, oh, x, y, global_alpha=1.0):
    """In place on `dest`; overlay is BGRA with stride ow*4.  Returns the status."""
    L = lib()
    L.orc_overlay_blend.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_uint32, c_uint32, c_uint32,
                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_float]
    L.orc_overlay_blend.restype = c_int
    return L.orc_overlay_blend(dest.ctypes.data, width, height, stride, _fmt(fmt), overlay.ctypes.data, ow, oh, ow * 4, x, y, global_alpha)
