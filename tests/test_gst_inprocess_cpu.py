"""Application-style use of the elements that need no GPU (roundedcorners on system memory renders its mask with cairo on
the host): caps change in mid-stream."""
import pytest

from tests import gst_inprocess

pytestmark = pytest.mark.skipif(not gst_inprocess.available(), reason="no PyGObject + GStreamer in this environment")


def test_roundedcorners_rerenders_its_mask_when_the_size_changes_in_mid_stream():
    r = gst_inprocess.run("renegotiate", "rounded")
    assert r["frames"] == 8
    assert r["sizes"] == [[96, 64], [160, 120]]
    assert r["mismatches"] == []


def test_roundedcorners_radius_changed_while_playing_including_back_to_passthrough():
    r = gst_inprocess.run("rounded_radius_change")
    assert r["frames"] == 9
    assert r["formats"] == ["A420"] * 6 + ["I420"] * 3
    assert r["mismatches"] == []
