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


def test_dssim_explainer_baseline_is_the_oracle_and_a_changed_constant_is_found():
    """Round 5: when the crate's dssim differs, compare.py says WHICH choice of the restatement to look at.  Checked
    without cargo: (a) the numpy twin's baseline equals oracle/ssim_oracle.c on the committed pairs (so its variants are variants of the
    oracle), (b) a 'crate' that differs from the restatement in one constant -- here: the 3x3 kernel applied twice instead of the
    binomial window -- is named by the report."""
    sys.path.insert(0, PIN)
    sys.path.insert(0, ROOT)
    import compare
    import dssim_variants as dv
    from tests import oracle_binding as orc
    frames = compare.load_frames()
    names = list(frames)
    W, H = compare.W, compare.H
    pairs = {}
    for i, a in enumerate(names):
        for b in names[i:]:
            ours = dv.dssim(frames[a], frames[b], W, H, dv.BASELINE)
            rc, d, _ = orc.ssim_distance(frames[a], frames[b], W, H, W * 4, W * 4, "RGBA")
            assert rc == 0 and abs(ours - d) <= 1e-9 * max(abs(d), 1e-12), (a, b, ours, d)
            pairs[f"{a}|{b}"] = {"dssim": dv.dssim(frames[a], frames[b], W, H, dict(dv.BASELINE, window="gauss3x3_twice"))}
    lines = dv.explain(frames, pairs, W, H)
    assert lines[1].lstrip().startswith("MATCHES window=gauss3x3_twice"), lines[:4]
    assert "`window=gauss3x3_twice` reproduces the crate" in lines[-1]


def test_comparer_consumes_a_document_in_the_harness_output_format_end_to_end():
    """Round 6: tests/golden/pin_crates_output_format_sample.json has exactly the layout src/main.rs prints (the keys the
    harness emits per frame and per pair, hashes as hex strings, distances as integers, dssim as a float) -- values: the restatement's own, the
    dssim values with ONE constant changed.  compare.py must read it, report the hashes / palettes / names PINNED, the non-trivial dssim values
    DIFFERS, name the changed constant in its per-constant report and leave with status 1; the README's commands are the ones tested here."""
    import json
    sample = os.path.join(ROOT, "tests", "golden", "pin_crates_output_format_sample.json")
    doc = json.load(open(sample))
    src = open(os.path.join(PIN, "src", "main.rs")).read()
    # the sample's keys are the keys the harness prints
    frame_keys = set(next(iter(doc["frames"].values())))
    for q, n in ((10, 2), (1, 8), (5, 5), (10, 255)):
        assert {f"palette_q{q}_n{n}", f"name_q{q}_n{n}"} <= frame_keys
    assert 'format!("\\"palette_q{q}_n{n}\\": {}"' in src and 'format!("\\"name_q{q}_n{n}\\": \\"{}\\""' in src
    for algo in ("mean", "gradient", "vertgradient", "doublegradient", "blockhash"):
        assert algo in frame_keys and f'("{algo}", HashAlg::' in src
    assert set(next(iter(doc["pairs"].values()))) == {"dssim", "mean", "gradient", "vertgradient", "doublegradient", "blockhash"}
    assert len(doc["frames"]) == 5 and len(doc["pairs"]) == 15 and "FORMAT SAMPLE" in doc["crates"]
    r = subprocess.run([sys.executable, os.path.join(PIN, "compare.py"), sample], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
    differs = [ln for ln in r.stdout.splitlines() if ln.startswith("DIFFERS")]
    assert differs and all(" dssim:" in ln for ln in differs), differs[:3]
    assert r.stdout.count("PINNED") > 100 and "MISSING" not in r.stdout
    assert "MATCHES window=gauss3x3_twice" in r.stdout and "`window=gauss3x3_twice` reproduces the crate" in r.stdout
    readme = open(os.path.join(PIN, "README.md")).read()
    assert "cargo run --release -- ../../tests/golden > crates.json" in readme and "compare.py tools/pin_crates/crates.json --explain" in readme
