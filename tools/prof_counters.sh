#!/bin/bash
# tools/prof_counters.sh <tag> <kernel-name-substring> <bench.py args...> -- on the GPU box: a kernel trace plus separate
# rocprofv3 PMC passes (no tracing domains beside --pmc) of bench.py; per-dispatch averages of the kernels whose name
# contains the substring land in gpurun_out/ctr_<tag>/summary.txt.
set -u
TAG=$1; MATCH=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ctr_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --settle-seconds 0.2 --no-cpu-baseline --pct-steps 0 --stream-threads 0 --content-sweep 0 --other-configs 0 $*"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_TA_BUSY" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "FETCH_SIZE" "WRITE_SIZE" "VALUBusy MemUnitBusy"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C -f csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/p$i.err"
done
cd "$REPO"
python3 - "$OUT" "$MATCH" "$ARGS" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
out, match, args = sys.argv[1], sys.argv[2], sys.argv[3]
print(f"# rocprofv3 --pmc passes of: python3 bench.py {args}   (kernels matching '{match}')")
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for i, row in enumerate(csv.reader(open(f))):
        if i == 0 or match in row[0]:
            print(",".join([row[0][:110]] + row[1:8]))
try:
    print("# bench line under trace:", open(out + "/bench_under_trace.json").read().strip()[:600])
except OSError:
    pass
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if match in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in acc.items():
        for c, v in ctrs.items():
            print(f"{name}: {c} n={len(v)} avg={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
PY
find "$OUT" -name "*.csv" -size +1M -delete
find "$OUT" -name "*.db" -delete
