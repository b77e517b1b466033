#!/bin/bash
# tools/r6_single_trace.sh [MVFX_EXP_SINGLE value]: rocprofv3 kernel trace of tools/exp_single_frame_trace.py (2400 single-frame launches on one
# stream, then 2400 alternating between two); prints, per phase, the median kernel duration, the median start-to-start interval and how much of a
# kernel overlaps its successor.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/strace_$$
[ -n "${1:-}" ] && export MVFX_EXP_SINGLE=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d "$OUT" -o t -- python3 "$R/tools/exp_single_frame_trace.py" > "$OUT.log" 2>&1
cat "$OUT.log" | grep "us per call"
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hsvfilter4_typed" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(f"{len(rows)} hsvfilter4_typed dispatches")
for name, lo, hi in (("one stream", 500, 2300), ("two streams", 2900, 4700)):
    seg = rows[lo:hi]
    st = [int(r["Start_Timestamp"]) for r in seg]
    en = [int(r["End_Timestamp"]) for r in seg]
    dur = [e - s for s, e in zip(st, en)]
    s2s = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    gap = [st[i + 1] - en[i] for i in range(len(st) - 1)]
    e2e = sorted(en)
    e2e = [e2e[i + 1] - e2e[i] for i in range(len(e2e) - 1)]
    q = lambda v, p: sorted(v)[int(p * (len(v) - 1))] / 1e3
    print(f"{name}: kernel duration p10/p50/p90 {q(dur,.1):.2f}/{q(dur,.5):.2f}/{q(dur,.9):.2f} us; start-to-start p50 {q(s2s,.5):.2f}; end-to-end p50 {q(e2e,.5):.2f}; "
          f"next start - this end p10/p50/p90 {q(gap,.1):.2f}/{q(gap,.5):.2f}/{q(gap,.9):.2f} us (negative = overlap); mean interval {(st[-1]-st[0])/(len(st)-1)/1e3:.2f} us")
PY
rm -rf "$OUT" "$OUT.log"
