#!/usr/bin/env python3
"""Golden vectors for imagersoverlay's per-frame work (`composition.blend(frame)`, video/image/src/overlay/imp.rs:703-727),
made with the REAL library the element calls: gst_video_overlay_composition_blend of libgstvideo (1.14.0 in the build
image, /opt/conda/lib), driven through ctypes.  One unscaled BGRA rectangle per case (what load_image builds,
imp.rs:241-283), optional global alpha (imp.rs:183-185), positions that clip against every frame edge.

    python tests/golden/make_overlay_blend_golden.py      # writes tests/golden/overlay_blend_kat.npz

Fixture = inputs + expected outputs (data only)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = "/opt/conda/lib/"
os.environ.setdefault("GST_PLUGIN_SYSTEM_PATH", "/opt/conda/lib/gstreamer-1.0")
os.environ.setdefault("GST_REGISTRY", "/tmp/gst-golden-registry.bin")
gst = ctypes.CDLL(LIBDIR + "libgstreamer-1.0.so.0")
gv = ctypes.CDLL(LIBDIR + "libgstvideo-1.0.so.0")
vp = ctypes.c_void_p
gst.gst_buffer_new_allocate.restype = vp
gst.gst_buffer_new_allocate.argtypes = [vp, ctypes.c_size_t, vp]
gst.gst_buffer_fill.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t]
gst.gst_buffer_extract.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t]
gst.gst_mini_object_unref.argtypes = [vp]
gst.gst_version_string.restype = ctypes.c_char_p
gv.gst_video_format_from_string.argtypes = [ctypes.c_char_p]
gv.gst_video_info_init.argtypes = [vp]
gv.gst_video_info_set_format.argtypes = [vp, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
gv.gst_video_frame_map.argtypes = [vp, vp, vp, ctypes.c_int]
gv.gst_video_frame_unmap.argtypes = [vp]
gv.gst_buffer_add_video_meta.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
gv.gst_buffer_add_video_meta.restype = vp
gv.gst_video_overlay_rectangle_new_raw.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_int]
gv.gst_video_overlay_rectangle_new_raw.restype = vp
gv.gst_video_overlay_rectangle_set_global_alpha.argtypes = [vp, ctypes.c_float]
gv.gst_video_overlay_composition_new.argtypes = [vp]
gv.gst_video_overlay_composition_new.restype = vp
gv.gst_video_overlay_composition_blend.argtypes = [vp, vp]


def blend(dest, fmt, w, h, overlay, ow, oh, x, y, global_alpha):
    """dest: uint8 [h, GStreamer stride]; overlay: uint8 [oh, ow*4] BGRA.  Returns the blended frame."""
    fid = gv.gst_video_format_from_string(fmt.encode())
    info = ctypes.create_string_buffer(1024)
    gv.gst_video_info_init(info)
    gv.gst_video_info_set_format(info, fid, w, h)
    buf = gst.gst_buffer_new_allocate(None, dest.nbytes, None)
    gst.gst_buffer_fill(buf, 0, dest.ctypes.data, dest.nbytes)
    obuf = gst.gst_buffer_new_allocate(None, overlay.nbytes, None)
    gst.gst_buffer_fill(obuf, 0, overlay.ctypes.data, overlay.nbytes)
    gv.gst_buffer_add_video_meta(obuf, 0, gv.gst_video_format_from_string(b"BGRA"), ow, oh)
    rect = gv.gst_video_overlay_rectangle_new_raw(obuf, x, y, ow, oh, 0)
    if global_alpha != 1.0:
        gv.gst_video_overlay_rectangle_set_global_alpha(rect, global_alpha)
    comp = gv.gst_video_overlay_composition_new(rect)
    frame = ctypes.create_string_buffer(4096)
    assert gv.gst_video_frame_map(frame, info, buf, 3)
    assert gv.gst_video_overlay_composition_blend(comp, frame)
    gv.gst_video_frame_unmap(frame)
    out = np.empty_like(dest)
    gst.gst_buffer_extract(buf, 0, out.ctypes.data, out.nbytes)
    for o in (comp, rect, obuf, buf):
        gst.gst_mini_object_unref(o)
    return out


CASES = []  # (format, w, h, ow, oh, x, y, global_alpha)
for fmt in ("RGBA", "BGRA", "ARGB", "ABGR", "RGBx", "BGRx", "xRGB", "xBGR", "RGB", "BGR"):
    CASES += [(fmt, 64, 48, 24, 16, 7, 5, 1.0), (fmt, 64, 48, 24, 16, -9, 40, 0.6)]
CASES += [("RGBA", 61, 37, 80, 50, -10, -6, 1.0), ("BGRA", 32, 32, 8, 8, 31, 31, 0.999), ("RGB", 41, 23, 16, 16, 30, -8, 0.3),
          ("RGBA", 40, 30, 16, 16, 100, 5, 1.0), ("RGBA", 256, 256, 256, 256, 0, 0, 1.0), ("RGBA", 64, 64, 64, 64, 0, 0, 0.5),
          ("BGRx", 48, 32, 20, 20, 3, 3, 0.004), ("ARGB", 48, 32, 20, 20, 3, 3, 0.0039)]


def main():
    gst.gst_init(None, None)
    rng = np.random.default_rng(0x0B1E)
    out = {"gst_version": np.frombuffer(gst.gst_version_string(), dtype=np.uint8)}
    names = []
    for k, (fmt, w, h, ow, oh, x, y, ga) in enumerate(CASES):
        bpp = 3 if fmt in ("RGB", "BGR") else 4
        stride = (w * bpp + 3) // 4 * 4
        dest = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        ov = rng.integers(0, 256, (oh, ow * 4), dtype=np.uint8)
        if (w, h, ow) == (256, 256, 256):  # every (source alpha, destination alpha) pair; patterned colours (compressible)
            xx, yy = np.meshgrid(np.arange(256), np.arange(256))
            o4, d4 = ov.reshape(oh, ow, 4), dest.reshape(h, w, 4)
            for c, (ka, kb) in enumerate(((7, 13), (29, 3), (11, 37))):
                o4[..., c] = ((xx >> 4) * ka + (yy >> 5) * kb + 17 * c) & 255   # constant over 16 x 32 blocks: the fixture stays small
                d4[..., c] = ((xx >> 5) * kb + (yy >> 4) * ka + 101 * c) & 255
            o4[..., 3] = xx
            d4[..., 3] = yy
        else:
            a = ov.reshape(oh, ow, 4)[..., 3]
            a[rng.random(a.shape) < 0.2] = 0
            a[rng.random(a.shape) < 0.2] = 255
        res = blend(dest, fmt, w, h, ov, ow, oh, x, y, ga)
        name = f"c{k:02d}"
        names.append(name)
        out[name + "_meta"] = np.array([w, h, stride, ow, oh, x, y], dtype=np.int64)
        out[name + "_fmt"] = np.frombuffer(fmt.encode(), dtype=np.uint8)
        out[name + "_alpha"] = np.array([ga], dtype=np.float32)
        out[name + "_dest"], out[name + "_overlay"], out[name + "_expect"] = dest, ov, res
        print(name, fmt, w, h, ow, oh, x, y, ga, "changed bytes", int(np.count_nonzero(res != dest)))
    out["cases"] = np.frombuffer(",".join(names).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "overlay_blend_kat.npz"), **out)


if __name__ == "__main__":
    main()
