#!/usr/bin/env python3
"""Generates the roundedcorners golden masks with the SAME C library the reference calls
(libcairo, through cairo-rs): replays the exact call sequence of
video/videofx/src/border/imp.rs:57-106 (draw_rounded_corners) and :108-149 (generate_alpha_mask)
on an A8 surface of stride round_up_4(width) and round_up_2(height) rows.

    python tests/golden/make_cairo_masks.py        # needs libcairo (1.16.0 in the build image)

Output: tests/golden/roundedcorners_masks.npz (compressed; masks are mostly 0/255) with one
array per case named w{W}_h{H}_r{R}, plus `cairo_version`."""
import ctypes
import ctypes.util
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(32, 24, 8), (64, 48, 10), (641, 481, 33), (640, 480, 0), (33, 17, 40), (1920, 1080, 1),
         (1920, 1080, 50), (1920, 1080, 540), (3840, 2160, 100),
         # 2*radius > min(width, height): cairo draws the self-overlapping path; the element must not refuse it
         (64, 48, 30), (1920, 1080, 700), (100, 101, 4000)]


def load_cairo():
    for cand in ("/opt/conda/lib/libcairo.so.2", ctypes.util.find_library("cairo"), "libcairo.so.2"):
        if not cand:
            continue
        try:
            return ctypes.CDLL(cand)
        except OSError:
            pass
    raise SystemExit("libcairo not found")


def make_mask(c, w, h, radius):
    stride = (w + 3) // 4 * 4            # GstVideoInfo.stride[3] of A420
    rows = (h + 1) & ~1                  # border/imp.rs:469-470
    buf = np.zeros(stride * rows, dtype=np.uint8)
    if radius == 0:                      # border/imp.rs:123-128
        buf[:] = 0xFF
        return buf.reshape(rows, stride)
    c.cairo_image_surface_create_for_data.restype = ctypes.c_void_p
    c.cairo_image_surface_create_for_data.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    c.cairo_create.restype = ctypes.c_void_p
    c.cairo_create.argtypes = [ctypes.c_void_p]
    for name, n in (("cairo_arc", 5), ("cairo_set_source_rgb", 3), ("cairo_set_source_rgba", 4), ("cairo_set_line_width", 1)):
        getattr(c, name).argtypes = [ctypes.c_void_p] + [ctypes.c_double] * n
    for name in ("cairo_new_sub_path", "cairo_close_path", "cairo_fill_preserve", "cairo_stroke", "cairo_destroy",
                 "cairo_surface_flush", "cairo_surface_destroy"):
        getattr(c, name).argtypes = [ctypes.c_void_p]
    surf = c.cairo_image_surface_create_for_data(buf.ctypes.data, 2, w, h, stride)  # CAIRO_FORMAT_A8 = 2
    cr = c.cairo_create(surf)
    r = float(radius)
    deg = math.pi / 180.0
    fw, fh = float(w), float(h)
    c.cairo_new_sub_path(cr)
    c.cairo_arc(cr, fw - r, r, r, -90.0 * deg, 0.0 * deg)
    c.cairo_arc(cr, fw - r, fh - r, r, 0.0 * deg, 90.0 * deg)
    c.cairo_arc(cr, r, fh - r, r, 90.0 * deg, 180.0 * deg)
    c.cairo_arc(cr, r, r, r, 180.0 * deg, 270.0 * deg)
    c.cairo_close_path(cr)
    c.cairo_set_source_rgb(cr, 0.0, 0.0, 0.0)
    c.cairo_fill_preserve(cr)
    c.cairo_set_source_rgba(cr, 0.0, 0.0, 0.0, 1.0)
    c.cairo_set_line_width(cr, 1.0)
    c.cairo_stroke(cr)
    c.cairo_destroy(cr)
    c.cairo_surface_flush(surf)
    c.cairo_surface_destroy(surf)
    return buf.reshape(rows, stride)


def main():
    c = load_cairo()
    c.cairo_version_string.restype = ctypes.c_char_p
    out = {"cairo_version": np.frombuffer(c.cairo_version_string(), dtype=np.uint8)}
    for (w, h, r) in CASES:
        m = make_mask(c, w, h, r)
        out[f"w{w}_h{h}_r{r}"] = m
        partial = int(np.count_nonzero((m > 0) & (m < 255)))
        print(f"{w}x{h} r={r}: stride {m.shape[1]} rows {m.shape[0]} partial {partial} first row {m[0, :12].tolist()}")
    np.savez_compressed(os.path.join(HERE, "roundedcorners_masks.npz"), **out)


if __name__ == "__main__":
    main()
