#!/bin/bash
# tools/prof_l2.sh <tag> <kernel-substring> <bench.py args...>: L1->L2 requests and L2 hit/miss of one kernel
set -u
TAG=$1; MATCH=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ctr_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --settle-seconds 0.2 --no-cpu-baseline $*"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C -f csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/p$i.err"
done
cd "$REPO"
python3 - "$OUT" "$MATCH" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
out, match = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if match in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in acc.items():
        for c, v in ctrs.items():
            print(f"{c} n={len(v)} avg={sum(v)/len(v):.6g}")
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
