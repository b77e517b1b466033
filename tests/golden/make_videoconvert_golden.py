#!/usr/bin/env python3
"""Writes tests/golden/videoconvert_kat.npz: outputs of the image's OWN GStreamer 1.14.0 `videoconvert` for
I420 -> RGBA and RGBA -> I420 on seeded random frames (inputs are regenerated from the seeds by the tests; small
cases keep the full output, large ones a sha256).  This is the real element the reference's colorlut example wraps
around the filter (colorlut/imp.rs:17-19), run here through gst-launch-1.0 + rawvideoparse -- not a restatement."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import frames  # noqa: E402


def ru(v, a):
    return (v + a - 1) // a * a


def i420_layout(w, h):
    """GstVideoInfo layout of I420: strides RU4(w), RU4(RU2(w)/2); plane rows RU2(h), RU2(h)/2"""
    ys, cs = ru(w, 4), ru(ru(w, 2) // 2, 4)
    yr, cr = ru(h, 2), ru(h, 2) // 2
    return ys, cs, yr, cr


def seeded_i420(seed, w, h):
    ys, cs, yr, cr = i420_layout(w, h)
    raw = frames.splitmix64_bytes(seed, ys * yr + 2 * cs * cr)
    return raw, ys, cs, yr, cr


def gst(cmd, tmp):
    env = dict(os.environ)
    env["PATH"] = "/opt/conda/bin:" + env["PATH"]
    env["GST_PLUGIN_SYSTEM_PATH"] = "/opt/conda/lib/gstreamer-1.0"
    env["GST_REGISTRY"] = os.path.join(tmp, "reg.bin")
    r = subprocess.run(cmd.split(), env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]


def main():
    out = {}
    meta = []
    with tempfile.TemporaryDirectory() as tmp:
        for (seed, w, h) in [(0x5EED0B01, 64, 32), (0x5EED0B02, 66, 34), (0x5EED0B03, 65, 33), (0x5EED0B04, 16, 578),
                             (0x5EED0B05, 1280, 720), (0x5EED0B06, 3840, 2160)]:
            raw, ys, cs, yr, cr = seeded_i420(seed, w, h)
            open(f"{tmp}/in.i420", "wb").write(raw.tobytes())
            gst(f"gst-launch-1.0 -q filesrc location={tmp}/in.i420 blocksize={raw.size} ! rawvideoparse format=i420 width={w} height={h} "
                f"! videoconvert ! video/x-raw,format=RGBA ! filesink location={tmp}/out.rgba", tmp)
            rgba = np.fromfile(f"{tmp}/out.rgba", dtype=np.uint8)
            assert rgba.size == w * h * 4
            key = f"i420_to_rgba_{w}x{h}"
            meta.append((key, seed, w, h, hashlib.sha256(rgba.tobytes()).hexdigest()))
            if rgba.size <= 64 * 1024:
                out[key] = rgba.reshape(h, w * 4)
        # even sizes and (round 3) odd ones: the element replicates the last column / row to the next even size
        for (seed, w, h) in [(0x5EED0C01, 64, 32), (0x5EED0C02, 2, 2), (0x5EED0C03, 6, 600), (0x5EED0C04, 16, 578),
                             (0x5EED0C05, 1280, 720), (0x5EED0C06, 3840, 2160), (0x5EED0C07, 65, 33), (0x5EED0C08, 3, 3),
                             (0x5EED0C09, 5, 7), (0x5EED0C0A, 7, 601), (0x5EED0C0B, 66, 33), (0x5EED0C0C, 65, 34), (0x5EED0C0D, 1, 1),
                             (0x5EED0C0E, 641, 481), (0x5EED0C0F, 1919, 1079)]:
            px = frames.random_frame(seed, w, h)
            open(f"{tmp}/in.rgba", "wb").write(px.tobytes())
            gst(f"gst-launch-1.0 -q filesrc location={tmp}/in.rgba blocksize={px.size} ! rawvideoparse format=rgba width={w} height={h} "
                f"! videoconvert ! video/x-raw,format=I420 ! filesink location={tmp}/out.i420", tmp)
            ys, cs, yr, cr = i420_layout(w, h)
            d = np.fromfile(f"{tmp}/out.i420", dtype=np.uint8)
            assert d.size == ys * yr + 2 * cs * cr
            # keep only the picture area (row padding is not defined)
            Y = d[:ys * yr].reshape(yr, ys)[:h, :w]
            U = d[ys * yr: ys * yr + cs * cr].reshape(cr, cs)[:(h + 1) // 2, :(w + 1) // 2]
            V = d[ys * yr + cs * cr:].reshape(cr, cs)[:(h + 1) // 2, :(w + 1) // 2]
            packed = np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])
            key = f"rgba_to_i420_{w}x{h}"
            meta.append((key, seed, w, h, hashlib.sha256(packed.tobytes()).hexdigest()))
            if packed.size <= 64 * 1024:
                out[key] = packed
        # RGBA -> NV12 (round 3): Y plane + interleaved UV plane (strides RU4(w) and RU4(RU2(w)), RU2(h) and RU2(h)/2 rows)
        for (seed, w, h) in [(0x5EED0D01, 64, 32), (0x5EED0D02, 65, 33), (0x5EED0D03, 16, 578), (0x5EED0D04, 7, 601), (0x5EED0D05, 8, 2160),
                             (0x5EED0D06, 1280, 720), (0x5EED0D07, 3840, 2160)]:
            px = frames.random_frame(seed, w, h)
            open(f"{tmp}/in.rgba", "wb").write(px.tobytes())
            gst(f"gst-launch-1.0 -q filesrc location={tmp}/in.rgba blocksize={px.size} ! rawvideoparse format=rgba width={w} height={h} "
                f"! videoconvert ! video/x-raw,format=NV12 ! filesink location={tmp}/out.nv12", tmp)
            ys, uvs, yr, cr = ru(w, 4), ru(ru(w, 2), 4), ru(h, 2), ru(h, 2) // 2
            d = np.fromfile(f"{tmp}/out.nv12", dtype=np.uint8)
            assert d.size == ys * yr + uvs * cr
            Y = d[:ys * yr].reshape(yr, ys)[:h, :w]
            UV = d[ys * yr:].reshape(cr, uvs)[:(h + 1) // 2, :2 * ((w + 1) // 2)]
            packed = np.concatenate([Y.reshape(-1), UV.reshape(-1)])
            key = f"rgba_to_nv12_{w}x{h}"
            meta.append((key, seed, w, h, hashlib.sha256(packed.tobytes()).hexdigest()))
            if packed.size <= 64 * 1024:
                out[key] = packed
        # NV12 -> RGBA (round 3): the element's generic path (chroma interpolated, horizontally then vertically)
        for (seed, w, h) in [(0x5EED0E01, 64, 32), (0x5EED0E02, 66, 34), (0x5EED0E03, 65, 33), (0x5EED0E04, 7, 5), (0x5EED0E05, 2, 2),
                             (0x5EED0E06, 4, 6), (0x5EED0E07, 16, 578), (0x5EED0E08, 8, 2160), (0x5EED0E09, 1280, 720), (0x5EED0E0A, 1, 1),
                             (0x5EED0E0B, 3840, 2160)]:
            ys, uvs, yr, cr = ru(w, 4), ru(ru(w, 2), 4), ru(h, 2), ru(h, 2) // 2
            raw = frames.splitmix64_bytes(seed, ys * yr + uvs * cr)
            open(f"{tmp}/in.nv12", "wb").write(raw.tobytes())
            gst(f"gst-launch-1.0 -q filesrc location={tmp}/in.nv12 blocksize={raw.size} ! rawvideoparse format=nv12 width={w} height={h} "
                f"! videoconvert ! video/x-raw,format=RGBA ! filesink location={tmp}/out.rgba", tmp)
            rgba = np.fromfile(f"{tmp}/out.rgba", dtype=np.uint8)
            assert rgba.size == w * h * 4
            key = f"nv12_to_rgba_{w}x{h}"
            meta.append((key, seed, w, h, hashlib.sha256(rgba.tobytes()).hexdigest()))
            if rgba.size <= 64 * 1024:
                out[key] = rgba.reshape(h, w * 4)
    out["meta"] = np.array([f"{k}|{s}|{w}|{h}|{d}" for (k, s, w, h, d) in meta])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "videoconvert_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
