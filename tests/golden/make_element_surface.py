#!/usr/bin/env python3
"""Extracts the element surface (the drop-in contract of SURVEY.md 8b) of the six elements on
the path (plus imagersoverlay) from the reference's machine-readable docs cache
(/root/reference/docs/plugins/gst_plugins_cache.json) into a small JSON fixture:
plugin name/license/description, element long-name/klass/description/author/hierarchy,
pad-template formats and every property's type/default/range/mutability.

    python tests/golden/make_element_surface.py   # needs /root/reference (build container only)
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/docs/plugins/gst_plugins_cache.json"
WANT = {"hsv": ["hsvfilter", "hsvdetector"], "colorlut": ["colorlut"],
        "rsvideofx": ["colordetect", "roundedcorners", "videocompare"], "imagers": ["imagersoverlay"]}


def formats_of(caps: str):
    m = re.search(r"format:\s*(\{[^}]*\}|\S+)", caps)
    s = m.group(1).strip("{} ")
    return [t.strip() for t in s.split(",")]


def main():
    with open(SRC) as f:
        cache = json.load(f)
    out = {}
    for plugin, elements in WANT.items():
        p = cache[plugin]
        entry = {"description": p["description"], "license": p["license"], "package": p["package"],
                 "filename": p["filename"], "elements": {}}
        for name in elements:
            e = p["elements"][name]
            entry["elements"][name] = {
                "long-name": e.get("long-name"), "klass": e["klass"], "description": e["description"],
                "author": e["author"], "hierarchy": e["hierarchy"], "rank": e["rank"],
                "pads": {pn: {"direction": pv["direction"], "presence": pv["presence"], "formats": formats_of(pv["caps"])}
                         for pn, pv in e["pad-templates"].items()},
                "properties": {pn: {k: pv[k] for k in ("type", "default", "min", "max", "mutable", "blurb", "readable", "writable") if k in pv}
                               for pn, pv in e["properties"].items()},
            }
        out[plugin] = entry
    with open(os.path.join(HERE, "element_surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote element_surface.json")


if __name__ == "__main__":
    main()
