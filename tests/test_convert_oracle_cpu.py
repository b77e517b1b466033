"""The videoconvert restatement (oracle/convert_oracle.c) against outputs of the REAL GStreamer 1.14.0 videoconvert
(tests/golden/videoconvert_kat.npz, made by tests/golden/make_videoconvert_golden.py with the image's gst-launch-1.0):
byte for byte on the small cases, sha256 on 1280x720 and 3840x2160.  This pins the oracle for SURVEY 8f-3."""
import hashlib
import os

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "videoconvert_kat.npz"))
META = [m.split("|") for m in KAT["meta"].tolist()]


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("i420_to_rgba")], ids=lambda m: m[0])
def test_i420_to_rgba_matches_gstreamer(meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    size = orc.i420_layout(w, h)[6]
    raw = frames.splitmix64_bytes(seed, size)
    rc, got = orc.convert_i420_to_rgba(raw, w, h)
    assert rc == 0
    assert hashlib.sha256(got.tobytes()).hexdigest() == digest
    if key in KAT.files:
        assert np.array_equal(got, KAT[key])


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("rgba_to_i420")], ids=lambda m: m[0])
def test_rgba_to_i420_matches_gstreamer(meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    px = frames.random_frame(seed, w, h)
    rc, Y, U, V = orc.convert_rgba_to_i420(px, w, h, w * 4)
    assert rc == 0
    packed = np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])
    assert hashlib.sha256(packed.tobytes()).hexdigest() == digest
    if key in KAT.files:
        assert np.array_equal(packed, KAT[key])


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("rgba_to_nv12")], ids=lambda m: m[0])
def test_rgba_to_nv12_matches_gstreamer(meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    px = frames.random_frame(seed, w, h)
    rc, Y, UV = orc.convert_rgba_to_nv12(px, w, h, w * 4)
    assert rc == 0
    packed = np.concatenate([Y.reshape(-1), UV.reshape(-1)])
    assert hashlib.sha256(packed.tobytes()).hexdigest() == digest
    if key in KAT.files:
        assert np.array_equal(packed, KAT[key])


@pytest.mark.parametrize("meta", [m for m in META if m[0].startswith("nv12_to_rgba")], ids=lambda m: m[0])
def test_nv12_to_rgba_matches_gstreamer(meta):
    key, seed, w, h, digest = meta[0], int(meta[1]), int(meta[2]), int(meta[3]), meta[4]
    raw = frames.splitmix64_bytes(seed, orc.nv12_layout(w, h)[5])
    rc, got = orc.convert_nv12_to_rgba(raw, w, h)
    assert rc == 0
    assert hashlib.sha256(got.tobytes()).hexdigest() == digest
    if key in KAT.files:
        assert np.array_equal(got, KAT[key])


def test_standard_override_and_errors():
    w, h = 32, 16
    px = frames.random_frame(7, w, h)
    rc, Y1, U1, V1 = orc.convert_rgba_to_i420(px, w, h, w * 4, standard=1)
    rc, Y0, U0, V0 = orc.convert_rgba_to_i420(px, w, h, w * 4, standard=0)
    assert np.array_equal(Y0, Y1) and np.array_equal(U0, U1)       # <= 576 lines: BT.601 by default
    rc, Y2, U2, V2 = orc.convert_rgba_to_i420(px, w, h, w * 4, standard=2)
    rc, Y3, U3, V3 = orc.convert_rgba_to_i420(px, w, h, w * 4, standard=3)
    assert not np.array_equal(Y0, Y2) and not np.array_equal(Y2, Y3)
    assert orc.convert_rgba_to_i420(px, w - 1, h, w * 4)[0] == 0     # odd sizes: last column / row replicated (round 3)
    # grey: Y = 16 + 219/255 * v (approx), chroma 128
    g = np.full((h, w * 4), 200, np.uint8)
    rc, Y, U, V = orc.convert_rgba_to_i420(g, w, h, w * 4)
    assert (Y == ((66 + 129 + 25) * 200 >> 8) + 16).all() and (U == 128).all() and (V == 128).all()
