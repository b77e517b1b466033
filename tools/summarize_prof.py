#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/gpu_profile.sh into one text summary
(per-kernel average duration from the kernel trace; per-dispatch PMC averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary for {out}")
for f in find("trace/**/*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            print(",".join([row[0][:120]] + row[1:8]))
            if i > 12:
                break
for f in find("trace/**/*kernel_trace.csv"):
    durs = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            durs[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print(f"\n## kernel trace ({os.path.relpath(f, out)}): name, calls, avg_us, min_us, max_us")
    for name, d in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"{name[:110]}, {len(d)}, {sum(d)/len(d)/1e3:.2f}, {min(d)/1e3:.2f}, {max(d)/1e3:.2f}")
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_busy"):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row.get("Kernel_Name", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"\n## {sub} ({os.path.relpath(f, out)}): per-dispatch averages")
        for name, ctrs in acc.items():
            if "hsvfilter" not in name and "colorlut" not in name and "mvfx" not in name:
                continue
            print(name[:110])
            for c, v in sorted(ctrs.items()):
                print(f"    {c}: n={len(v)} avg={sum(v)/len(v):.6g}")
