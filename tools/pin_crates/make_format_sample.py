#!/usr/bin/env python3
"""tools/pin_crates/make_format_sample.py -- writes tests/golden/pin_crates_output_format_sample.json: a document in EXACTLY the layout
src/main.rs prints (same keys, same nesting, hashes as hex strings, distances as integers, dssim as a float), so that compare.py's
consumption of the harness's output can be tested end to end without cargo.  The VALUES are not crate output: they are this repository's
restatement's (oracle/), except that every dssim value comes from the restatement with ONE constant changed (the 3 x 3 kernel applied
twice instead of the binomial window) -- the sample therefore also exercises the DIFFERS path and the per-constant report.  The header
field "crates" says so.  Re-run after a change of the restatement:  python3 tools/pin_crates/make_format_sample.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import compare  # noqa: E402
import dssim_variants as dv  # noqa: E402
from tests import oracle_binding as orc  # noqa: E402


def main():
    frames = compare.load_frames()
    W, H = compare.W, compare.H
    doc = compare.self_document()
    doc["crates"] = ("FORMAT SAMPLE, not crate output: the restatement's own values in the harness's layout; dssim from the restatement with "
                     "window=gauss3x3_twice (tools/pin_crates/make_format_sample.py)")
    for name, f in frames.items():  # the harness also prints the five hashes of every frame as hex strings of their bytes
        for algo in ("mean", "gradient", "vertgradient", "doublegradient"):
            doc["frames"][name][algo] = f"{orc.image_hash(f, W, H, W * 4, 'RGBA', algo)[1]:016x}"
    names = list(frames)
    for i, a in enumerate(names):
        for b in names[i:]:
            doc["pairs"][f"{a}|{b}"]["dssim"] = dv.dssim(frames[a], frames[b], W, H, dict(dv.BASELINE, window="gauss3x3_twice"))
    out = os.path.join(ROOT, "tests", "golden", "pin_crates_output_format_sample.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
