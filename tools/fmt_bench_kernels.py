#!/usr/bin/env python3
"""stdin: the JSON lines of tools/bench_kernels.py -> one aligned text line per kernel."""
import json
import sys
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("%-100s %9.4f ms %10.1f /s  %.3f of 8 TB/s" % (d["kernel"][:100], d["ms_per_call"], d["frames_per_s"], d["frac_of_8TBs"]))
