"""Application-style use of the real elements in one process (PyGObject): property changes while PLAYING, caps changes in
mid-stream on system and device memory, and repeated NULL <-> PLAYING cycles of the device-memory chain."""
import pytest

from tests import gst_inprocess

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gst_inprocess.available(), reason="no PyGObject + GStreamer in this environment")]


@pytest.mark.parametrize("chain", ["", "hip"])
def test_hsvfilter_properties_changed_while_playing(chain):
    r = gst_inprocess.run("hsvfilter_property_change", chain)
    assert r["frames"] == 8
    assert r["mismatches"] == []


@pytest.mark.parametrize("chain", ["hsv", "hsv-hip"])
def test_size_change_in_mid_stream(chain):
    r = gst_inprocess.run("renegotiate", chain)
    assert r["frames"] == 8
    assert r["sizes"] == [[96, 64], [160, 120]]
    assert r["mismatches"] == []


@pytest.mark.parametrize("chain", ["", "hip"])
def test_hsvdetector_properties_changed_while_playing(chain):
    r = gst_inprocess.run("hsvdetector_property_change", chain)
    assert r["frames"] == 8
    assert r["mismatches"] == []
    assert r["opaque_pixels"][0] != r["opaque_pixels"][-1]   # the second set of ranges selects other pixels


@pytest.mark.parametrize("chain", ["", "hip"])
def test_imagersoverlay_properties_changed_while_playing(chain):
    r = gst_inprocess.run("overlay_property_change", chain)
    assert r["frames"] == 6
    assert r["mismatches"] == []


@pytest.mark.parametrize("chain", ["", "hip"])
def test_colorlut_location_changed_in_ready_between_two_runs(chain):
    r = gst_inprocess.run("colorlut_relocation", chain, timeout=90)
    assert r["luts_differ"]
    assert [x["frames"] for x in r["runs"]] == [3, 3]
    assert [x["mismatches"] for x in r["runs"]] == [[], []]


def test_videocompare_built_with_request_pads_reports_every_other_pad():
    r = gst_inprocess.run("videocompare_three_pads")
    first = [m for m in r["first"] if isinstance(m, dict)]
    assert len(first) == len(r["first"]) >= 1, r["first"]
    for m in first:
        assert sorted(m) == ["sink_1", "sink_2"], m      # never the reference pad itself (imp.rs:326-329)
        assert m["sink_1"] == 0.0 and m["sink_2"] > 0.0
    second = [m for m in r["second"] if isinstance(m, dict)]
    assert len(second) == len(r["second"]) >= 1, r["second"]
    assert all(sorted(m) == ["sink_1"] and m["sink_1"] == 0.0 for m in second)
    assert r["sink_pads_after_release"] == ["sink_0", "sink_1"]


def test_device_memory_chain_survives_state_cycles_without_accumulating_device_memory():
    r = gst_inprocess.run("state_cycles", 24, timeout=600)
    assert r["results"] == ["eos"]
    free = r["free_mb"]
    # after the first cycles (code objects, streams, the allocator's first blocks) free memory must stay flat
    assert min(free[6:]) > free[5] - 64.0, free


@pytest.mark.parametrize("element", ["hsvfilter", "hsvdetector"])
def test_a_failed_held_back_frame_fails_the_next_transform_call_with_flow_error(element):
    """VERDICT r4 (W-semantics): with pair launches on, a held-back frame whose launch fails must not only be posted -- the element's NEXT
    transform call returns GST_FLOW_ERROR, so the stream ends like the reference's, whose error is the flow return of the failing buffer
    itself (hsvfilter/imp.rs:322-326).  The application here keeps running after the first error message, so the bus shows all three
    stages: the posted failure of the held-back frame, the error of the next buffer's call, and the source's 'streaming stopped,
    reason error (-5)' -- GST_FLOW_ERROR came back up the stream.  Exactly two buffers entered the element, none after the error."""
    r = gst_inprocess.run("held_back_failure", element)
    errs = r["errors"]
    assert len(errs) >= 3, errs
    assert errs[0]["src"] == "e" and "held-back frame: mvfx status -8" in errs[0]["debug"] and "asserts on this" in errs[0]["message"]
    assert errs[1]["src"] == "e" and "the frame before this one failed" in errs[1]["debug"] and "asserts on this" in errs[1]["message"]
    assert any(e["src"] == "src" and "reason error (-5)" in e["debug"] for e in errs[2:]), errs
    assert r["buffers_in"] == 2 and r["buffers_out"] <= 1
