#!/usr/bin/env python3
"""Generates the committed known-answer fixtures (SURVEY.md 8c "Fixtures to commit") from the
oracle (oracle/liboracle.so) and from the image's GStreamer 1.14 videotestsrc:

  hsv_kat.npz          4096 seeded pixels x 8 hsvfilter settings -> expected RGBA, and
                       x 4 hsvdetector settings -> expected alpha; the reference's 5 colour vectors
  colorlut_kat.npz     the .cube texts + 64x48 random RGBA / 32x4 RGBA64 LE+BE in -> expected out
  videotestsrc_*.bin   64x48 RGBA frames captured from videotestsrc (smpte, red, snow)
  videofx_kat.json     colordetect palette / name and blockhash of those frames + a seeded random
                       frame ("self-golden, upstream-unpinned": the crates are not in /root/reference)

    python tests/golden/make_golden.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import cubes, frames, gst_env  # noqa: E402
from tests import oracle_binding as orc  # noqa: E402

FILTER_SETTINGS = [(0.0, 1.0, 0.0, 1.0, 0.0), (90.0, 1.25, -0.05, 0.9, 0.02), (-123.4, 0.5, 0.3, 1.7, -0.2),
                   (360.0, 1.0, 0.0, 1.0, 0.0), (-360.0, 2.0, -0.5, 0.25, 0.5), (720.5, 0.7, 0.1, 1.2, -0.1),
                   (-1e6, 1.1, 0.0, 0.9, 0.0), (float("nan"), 1.0, 0.0, 1.0, 0.0)]
DETECT_SETTINGS = [(0.0, 10.0, 0.0, 0.15, 0.0, 0.3), (120.0, 40.0, 0.6, 0.4, 0.6, 0.4), (350.0, 25.0, 0.5, 0.5, 0.5, 0.5),
                   (-200.0, 180.0, 1.0, 1.0, 1.0, 1.0)]


def hsv():
    n = 4096
    px = frames.splitmix64_bytes(0x5EED0001, n * 4).reshape(1, n * 4)
    # make sure greys, primaries and near-equal channels are in the set
    special = np.array([[255, 255, 255, 1], [0, 0, 0, 2], [255, 0, 0, 3], [0, 255, 0, 4], [0, 0, 255, 5], [12, 200, 77, 9],
                        [128, 128, 127, 0], [1, 0, 0, 0], [254, 255, 255, 0], [17, 17, 17, 255]], np.uint8)
    px[0, :special.size] = special.reshape(-1)
    expected = np.empty((len(FILTER_SETTINGS), n * 4), np.uint8)
    for i, s in enumerate(FILTER_SETTINGS):
        f = px.copy()
        assert orc.hsvfilter(f, n, n * 4, "RGBA", s) == 0
        expected[i] = f[0]
    alpha = np.empty((len(DETECT_SETTINGS), n), np.uint8)
    for i, s in enumerate(DETECT_SETTINGS):
        out = np.empty_like(px)
        assert orc.hsvdetector(px, n * 4, "RGBx", out, n * 4, "RGBA", n, s) == 0
        alpha[i] = out[0, 3::4]
    np.savez_compressed(os.path.join(HERE, "hsv_kat.npz"), pixels=px[0], filter_settings=np.array(FILTER_SETTINGS, np.float32),
                        filter_expected=expected, detect_settings=np.array(DETECT_SETTINGS, np.float32), detect_alpha=alpha)


def colorlut():
    texts = {"analytic9": cubes.analytic_3d(9), "identity2": cubes.identity_3d(2, 1), "curve1d_16": cubes.curve_1d(16),
             "curve1d_domain": cubes.curve_1d(8, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9)))}
    out = {}
    rgba = frames.random_frame(0x5EED0500, 64, 48)
    wide = frames.random_frame(0x5EED0501, 32, 4, 8)
    out["rgba_in"] = rgba
    out["rgba64_in"] = wide
    for name, text in texts.items():
        with open(os.path.join(HERE, f"{name}.cube"), "w") as f:
            f.write(text)
        lut = orc.CubeLut(text)
        assert lut.ok
        o = np.empty_like(rgba)
        assert lut.apply(rgba, 256, o, 256, 64, 48, "RGBA") == 0
        out[f"{name}_rgba"] = o
        for fmt in ("RGBA64_LE", "RGBA64_BE"):
            o = np.empty_like(wide)
            assert lut.apply(wide, 256, o, 256, 32, 4, fmt) == 0
            out[f"{name}_{fmt.lower()}"] = o
    np.savez_compressed(os.path.join(HERE, "colorlut_kat.npz"), **out)


def videotestsrc():
    launch = gst_env.tool("gst-launch-1.0")
    res = {}
    if not launch:
        print("no gst-launch-1.0: keeping existing videotestsrc captures")
        return
    import tempfile
    tmp = tempfile.mkdtemp()
    for pattern in ("smpte", "red", "snow"):
        path = os.path.join(HERE, f"videotestsrc_{pattern}_64x48_RGBA.bin")
        e = gst_env.env(tmp)
        e.pop("GST_PLUGIN_PATH", None)
        subprocess.run([launch, "-q", "videotestsrc", "num-buffers=1", f"pattern={pattern}", "!",
                        "video/x-raw,format=RGBA,width=64,height=48", "!", "filesink", f"location={path}"], env=e, check=True)
        res[pattern] = path
    return res


def videofx():
    out = {"note": "self-golden from oracle/videofx_oracle.c; upstream crates (color-thief 0.2.2, color-name 1.2.0, "
                   "image_hasher 3.1.1) are not under /root/reference, parity unpinned beyond red => 'red' and identical => 0"}
    cases = {}
    for pattern in ("smpte", "red", "snow"):
        f = np.fromfile(os.path.join(HERE, f"videotestsrc_{pattern}_64x48_RGBA.bin"), np.uint8).reshape(48, 256)
        cases[f"videotestsrc_{pattern}"] = f
    cases["random_5EED0001_64x48"] = frames.random_frame(0x5EED0001, 64, 48)
    for name, f in cases.items():
        entry = {}
        for (q, mc) in ((10, 2), (1, 8)):
            rc, pal = orc.colordetect_palette(f, "RGBA", q, mc)
            entry[f"palette_q{q}_n{mc}"] = pal
            entry[f"name_q{q}_n{mc}"] = orc.css_similar((pal[0] >> 16) & 255, (pal[0] >> 8) & 255, pal[0] & 255)
        rc, h = orc.blockhash(f, 64, 48, 256, "RGBA")
        entry["blockhash"] = f"{h:016x}"
        rc, sums = orc.blockhash_sums(f, 64, 48, 256, "RGBA")
        entry["block_sums"] = [int(x) for x in sums]
        out[name] = entry
    with open(os.path.join(HERE, "videofx_kat.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    hsv()
    colorlut()
    videotestsrc()
    videofx()
    print(sorted(os.listdir(HERE)))
