#!/usr/bin/env python3
"""Writes tests/golden/imghash_kat.json: Mean / Gradient / VertGradient / DoubleGradient hashes of seeded random
frames from the numpy-f32 restatement (tests/np_twin.py).  Self-golden: image_hasher 3.1.1 / image 0.25.10 are not
under /root/reference and cannot be run here, so these vectors pin the build against itself, not against the crates."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import frames, np_twin  # noqa: E402

cases = []
for (seed, w, h, bpp, pad) in [(0x5EED0A01, 64, 48, 4, 0), (0x5EED0A02, 100, 37, 3, 4), (0x5EED0A03, 320, 240, 4, 0),
                               (0x5EED0A04, 16, 200, 4, 16), (0x5EED0A05, 6, 7, 3, 2)]:
    stride = w * bpp + pad
    f = frames.random_frame(seed, w, h, bpp, stride)
    hashes = {}
    for algo, (nw, nh) in np_twin.HASH_RESIZE.items():
        px = np_twin.gray_resize_lanczos3(f, w, h, bpp, nw, nh)
        bits = np_twin.image_hash_bits(px, algo)
        hashes[algo] = f"{sum(1 << k for k, b in enumerate(bits) if b):016x}"
    cases.append({"seed": seed, "width": w, "height": h, "bpp": bpp, "stride": stride,
                  "format": "RGBA" if bpp == 4 else "RGB", "hash": hashes})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "imghash_kat.json")
json.dump({"note": "self-golden (numpy-f32 restatement), upstream-unpinned", "cases": cases}, open(out, "w"), indent=1)
print("wrote", out)
