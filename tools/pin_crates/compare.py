#!/usr/bin/env python3
"""tools/pin_crates/compare.py crates.json -- compares what the pinned crates computed (tools/pin_crates, run on a machine with cargo)
with this repository's CPU restatement (oracle/liboracle.so through tests/oracle_binding.py) on the same frames, output by output.
Exit status 0: everything that is compared agrees (hash bits and palettes exactly, dssim within --dssim-rtol).  The report names
every output as PINNED (agrees), DIFFERS or MISSING; INTEGRATION.md "What is pinned" says what each result means for an element."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import frames as tframes  # noqa: E402
from tests import oracle_binding as orc  # noqa: E402

W, H = 64, 48


def load_frames():
    out = {}
    for n in ("red", "smpte", "snow"):
        out[f"videotestsrc_{n}"] = np.fromfile(os.path.join(ROOT, "tests", "golden", f"videotestsrc_{n}_64x48_RGBA.bin"), dtype=np.uint8).reshape(H, W * 4)
    out["random_5EED0001_64x48"] = tframes.random_frame(0x5EED0001, W, H)
    out["random_5EED0002_64x48"] = tframes.random_frame(0x5EED0002, W, H)
    return out


def self_document():
    """the document tools/pin_crates prints, filled in by the restatement itself (--self-test: exercises this script without cargo)"""
    frames = load_frames()
    doc = {"crates": "self", "frames": {}, "pairs": {}}
    for name, f in frames.items():
        d = {}
        for q, n in ((10, 2), (1, 8), (5, 5), (10, 255)):
            _, pal = orc.colordetect_palette(f, "RGBA", q, n)
            d[f"palette_q{q}_n{n}"] = [int(x) for x in pal]
            d[f"name_q{q}_n{n}"] = orc.css_similar((pal[0] >> 16) & 255, (pal[0] >> 8) & 255, pal[0] & 255).lower()
        d["blockhash"] = f"{orc.blockhash(f, W, H, W * 4, 'RGBA')[1]:016x}"
        doc["frames"][name] = d
    names = list(frames)
    for i, a in enumerate(names):
        for b in names[i:]:
            fa, fb = frames[a], frames[b]
            p = {"blockhash": orc.hamming(orc.blockhash(fa, W, H, W * 4, "RGBA")[1], orc.blockhash(fb, W, H, W * 4, "RGBA")[1]),
                 "dssim": orc.ssim_distance(fa, fb, W, H, W * 4, W * 4, "RGBA")[1]}
            for algo in ("mean", "gradient", "vertgradient", "doublegradient"):
                p[algo] = orc.hamming(orc.image_hash(fa, W, H, W * 4, "RGBA", algo)[1], orc.image_hash(fb, W, H, W * 4, "RGBA", algo)[1])
            doc["pairs"][f"{a}|{b}"] = p
    return doc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("crates_json", nargs="?")
    ap.add_argument("--self-test", action="store_true", help="compare the restatement with itself (checks this script, pins nothing)")
    ap.add_argument("--dssim-rtol", type=float, default=1e-4, help="relative tolerance of the dssim value (f32 library against the f64 restatement)")
    ap.add_argument("--explain", action="store_true",
                    help="always print the per-constant report of the dssim restatement (dssim_variants.py); printed anyway when a dssim value differs")
    args = ap.parse_args()
    if not args.self_test and not args.crates_json:
        ap.error("crates.json (the output of `cargo run --release` in tools/pin_crates) or --self-test")
    ref = self_document() if args.self_test else json.load(open(args.crates_json))
    frames = load_frames()
    bad = 0

    report_lines = []

    def report(what, ours, theirs, ok):
        nonlocal bad
        bad += 0 if ok else 1
        report_lines.append(f"{'PINNED ' if ok else 'DIFFERS'} {what}: ours {ours} crate {theirs}")
        print(report_lines[-1])

    for name, f in frames.items():
        r = ref["frames"].get(name)
        if r is None:
            print(f"MISSING {name}")
            bad += 1
            continue
        for q, n in ((10, 2), (1, 8), (5, 5), (10, 255)):
            rc, pal = orc.colordetect_palette(f, "RGBA", q, n)
            ours = [int(x) for x in pal[:rc]]
            report(f"{name} palette q={q} n={n}", ours, r[f"palette_q{q}_n{n}"], ours == r[f"palette_q{q}_n{n}"])
            if ours:
                nm = orc.css_similar((ours[0] >> 16) & 255, (ours[0] >> 8) & 255, ours[0] & 255).lower()
                report(f"{name} colour name q={q} n={n}", nm, r[f"name_q{q}_n{n}"], nm == r[f"name_q{q}_n{n}"])
        _, bh = orc.blockhash(f, W, H, W * 4, "RGBA")
        # bit / byte order of the crate's serialisation is not part of the element's behaviour: distances are compared below;
        # here equality of the 64-bit set is checked up to that order
        theirs = int(r["blockhash"], 16)
        report(f"{name} blockhash popcount", bin(bh).count("1"), bin(theirs).count("1"), bin(bh).count("1") == bin(theirs).count("1"))
    names = list(frames)
    for i, a in enumerate(names):
        for b in names[i:]:
            r = ref["pairs"].get(f"{a}|{b}")
            if r is None:
                print(f"MISSING pair {a}|{b}")
                bad += 1
                continue
            fa, fb = frames[a], frames[b]
            _, ha = orc.blockhash(fa, W, H, W * 4, "RGBA")
            _, hb = orc.blockhash(fb, W, H, W * 4, "RGBA")
            report(f"{a}|{b} blockhash distance", orc.hamming(ha, hb), r["blockhash"], orc.hamming(ha, hb) == r["blockhash"])
            for algo in ("mean", "gradient", "vertgradient", "doublegradient"):
                da = orc.hamming(orc.image_hash(fa, W, H, W * 4, "RGBA", algo)[1], orc.image_hash(fb, W, H, W * 4, "RGBA", algo)[1])
                report(f"{a}|{b} {algo} distance", da, r[algo], da == r[algo])
            rc, d, _ = orc.ssim_distance(fa, fb, W, H, W * 4, W * 4, "RGBA")
            ok = rc == 0 and (abs(d - r["dssim"]) <= args.dssim_rtol * max(abs(r["dssim"]), 1e-12) or (d == 0.0 and r["dssim"] == 0.0))
            report(f"{a}|{b} dssim", f"{d:.9e}", f"{r['dssim']:.9e}", ok)
    dssim_differs = any(ln.startswith("DIFFERS") and " dssim:" in ln for ln in report_lines)
    if args.explain or dssim_differs:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import dssim_variants
        for ln in dssim_variants.explain(frames, ref["pairs"], W, H, args.dssim_rtol):
            print(ln)
    print(f"{bad} output(s) differ or are missing" if bad else "every compared output agrees with the crates")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
