"""imagersoverlay's `composition.blend(frame)` (video/image/src/overlay/imp.rs:703-727 -> libgstvideo gst_video_blend):
the C oracle against golden vectors produced by the image's REAL libgstvideo 1.14.0
(tests/golden/make_overlay_blend_golden.py): every (source alpha, destination alpha) pair, ten destination formats, global
alpha, rectangles clipped by every edge."""
import os

import numpy as np
import pytest

from tests import oracle_binding as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "overlay_blend_kat.npz")


def cases():
    with np.load(GOLDEN) as g:
        return bytes(g["cases"]).decode().split(",")


def load_case(g, name):
    w, h, stride, ow, oh, x, y = (int(v) for v in g[name + "_meta"])
    return dict(w=w, h=h, stride=stride, ow=ow, oh=oh, x=x, y=y, fmt=bytes(g[name + "_fmt"]).decode(),
                alpha=float(g[name + "_alpha"][0]), dest=g[name + "_dest"], overlay=g[name + "_overlay"], expect=g[name + "_expect"])


@pytest.mark.parametrize("name", cases())
def test_oracle_equals_libgstvideo(name):
    with np.load(GOLDEN) as g:
        c = load_case(g, name)
    got = c["dest"].copy()
    assert orc.overlay_blend(got, c["w"], c["h"], c["stride"], c["fmt"], np.ascontiguousarray(c["overlay"]), c["ow"], c["oh"], c["x"], c["y"],
                             c["alpha"]) == 0
    assert np.array_equal(got, c["expect"]), f"{name} {c['fmt']}: {np.count_nonzero(got != c['expect'])} bytes differ"


def test_unsupported_destination_format():
    d = np.zeros(64, np.uint8)
    o = np.zeros(16, np.uint8)
    assert orc.overlay_blend(d, 4, 4, 16, "I420", o, 2, 2, 0, 0) != 0
