"""tools/pin_crates: the harness a maintainer with cargo runs to pin the third-party arithmetic (color-thief, color-name, image_hasher,
dssim-core) that /root/reference does not vendor.  It cannot be built here (no rustc); what CAN be checked: the versions it pins are
the ones the reference's Cargo.lock pins (recorded in SURVEY.md 8c), and the comparer runs end to end on the committed frames."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tools", "pin_crates")


def test_cargo_toml_pins_the_versions_of_the_reference_lock_file():
    text = open(os.path.join(PIN, "Cargo.toml")).read()
    for crate, version in (("color-thief", "0.2.2"), ("color-name", "1.2.0"), ("image_hasher", "3.1.1"), ("dssim-core", "3.4.0")):
        assert re.search(rf'^{re.escape(crate)} = "={re.escape(version)}"$', text, re.M), crate
    assert re.search(r'^image = \{ version = "=0\.25\.10"', text, re.M)


def test_harness_makes_the_calls_the_reference_makes():
    src = open(os.path.join(PIN, "src", "main.rs")).read()
    for call in ("get_palette(px, ColorFormat::Rgba, q, n)", "color_name::css::Color::similar(", "HasherConfig::new().hash_alg(*alg).to_hasher()",
                 "create_image_rgba(", "dssim.compare("):
        assert call in src, call


def test_comparer_runs_on_the_committed_frames():
    r = subprocess.run([sys.executable, os.path.join(PIN, "compare.py"), "--self-test"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "every compared output agrees" in r.stdout and "DIFFERS" not in r.stdout and "MISSING" not in r.stdout
    assert r.stdout.count("PINNED") > 100
