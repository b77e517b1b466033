#!/bin/bash
# tools/prof_pmc.sh <tag> <kernel-name-substring> <bench_kernels selector...> -- on the GPU box: separate rocprofv3 PMC
# passes (FETCH_SIZE, WRITE_SIZE, VALUBusy+MemUnitBusy; no tracing domains alongside) of tools/bench_kernels.py; prints per-dispatch averages
# of the kernels whose name contains the substring.
set -u
TAG=$1; MATCH=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "VALUBusy MemUnitBusy"; do
    D=$(echo $C | tr " " "_")
    timeout 300 rocprofv3 --pmc $C -f csv -d "$OUT/$D" -o pmc -- python3 "$REPO/tools/bench_kernels.py" "$@" > /dev/null 2> "$OUT/$D.err"
done
cd "$REPO"
python3 - "$OUT" "$MATCH" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
out, match = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if match in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in acc.items():
        for c, v in ctrs.items():
            print(f"{name}: {c} n={len(v)} avg={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
PY
find "$OUT" -name "*.csv" -size +2M -delete
