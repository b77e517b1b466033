#!/bin/bash
# tools/r6_lane_trace.sh [tool.py [args ...]]: rocprofv3 kernel trace of tools/exp_direct_lane.py (or of another tool of tools/, with its arguments);
# per kernel name: count, median duration, median start-to-start
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ltrace_$$
cd /tmp && export TMPDIR=/tmp
TOOL=${1:-exp_direct_lane.py}
shift || true
rocprofv3 --kernel-trace -f csv -d "$OUT" -o t -- python3 "$R/tools/$TOOL" "$@" > "$OUT.log" 2>&1
cat "$OUT.log" | tail -5
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no kernel trace"); sys.exit(0)
rows = list(csv.DictReader(open(fs[0])))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:60]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in by.items():
    v.sort()
    seg = v[len(v) // 2: len(v) // 2 + 2000]
    dur = sorted(e - s for s, e in seg)
    s2s = sorted(seg[i + 1][0] - seg[i][0] for i in range(len(seg) - 1))
    if len(seg) > 10:
        print(f"{k}: n={len(v)} duration p50 {dur[len(dur)//2]/1e3:.2f} us, start-to-start p50 {s2s[len(s2s)//2]/1e3:.2f} us")
PY
rm -rf "$OUT" "$OUT.log"
