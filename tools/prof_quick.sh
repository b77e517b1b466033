#!/bin/bash
# tools/prof_quick.sh <tag> <kernel-substring> <bench.py args...>: three PMC passes (SQ, LDS+VALUBusy, TCP) of bench.py
set -u
TAG=$1; MATCH=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ctr_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --settle-seconds 0.2 --no-cpu-baseline $*"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "VALUBusy MemUnitBusy"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C -f csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/p$i.err"
done
cd "$REPO"
python3 - "$OUT" "$MATCH" "$ARGS" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
out, match, args = sys.argv[1], sys.argv[2], sys.argv[3]
print(f"# rocprofv3 --pmc passes of: python3 bench.py {args}   (kernels matching '{match}')")
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if match in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in acc.items():
        for c, v in ctrs.items():
            print(f"{name}: {c} n={len(v)} avg={sum(v)/len(v):.6g}")
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
