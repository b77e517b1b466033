#!/bin/bash
# tools/trace_kernels.sh <kernel-substring> <python script + args...>: rocprofv3 kernel trace of the command; prints, per kernel
# name matching the substring, the median duration of each consecutive block of 4000 dispatches (phases of a micro-benchmark).
set -u
MATCH=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d "$OUT" -o t -- python3 "$R/$1" "${@:2}" > "$OUT.log" 2>&1
cd "$R"
python3 - "$OUT" "$MATCH" <<'PY'
import csv, glob, statistics, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i in range(0, len(rows), 4000):
    per = defaultdict(list)
    for r in rows[i:i + 4000]:
        per[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"dispatches {i:6d}+: " + "   ".join(f"{k} {statistics.median(v) / 1e3:.2f} us x{len(v)}" for k, v in per.items()))
PY
rm -rf "$OUT" "$OUT.log"
