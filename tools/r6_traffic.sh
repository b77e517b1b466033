#!/bin/bash
# tools/r6_traffic.sh -- on the GPU box: per BASELINE workload a rocprofv3 kernel trace and SEPARATE --pmc FETCH_SIZE and --pmc
# WRITE_SIZE passes (no tracing domain beside --pmc) of `bench.py --workload <w>`, plus the PMC calibration probe
# (tools/probe_pmc_calib.bin: known byte counts per access pattern).  tools/r6_traffic.py turns the result into
# profiles/traffic.json (what bench.py quotes as roofline.traffic) and profiles/r5/traffic_*.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r6traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--full 1 --no-cpu-baseline --pct-steps 0 --settle-seconds 0.05 --stream-threads 0 --content-sweep 0 --other-configs 0 --noise-sweep 0 --gst-pipeline 0 --warmup 2"
# the kernel trace runs with the bench's own settle time (0.6 s of the same step before the warm-up: the clock governor needs ~0.2 s of load)
# and the default warm-up, so that the TIMED launches of the trace are the launches the driver's line is made of
TRACE="--full 1 --no-cpu-baseline --pct-steps 0 --stream-threads 0 --content-sweep 0 --other-configs 0 --noise-sweep 0 --gst-pipeline 0 --warmup 5"
run() { # <key> <bench args...>
    local K=$1; shift
    mkdir -p $O/$K
    timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/$K/trace -o trace -- python3 $REPO/bench.py $TRACE "$@" > $O/$K/trace.json 2> $O/$K/trace.err
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $C -f csv -d $O/$K/$C -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/$C.json 2> $O/$K/$C.err
    done
    # which ceiling binds (round 5: MemUnitBusy reads 0.0 for every kernel on this stack -- dropped; evidence from counters that move):
    # SQ pass 1: issued / active instruction counters and the wave-state split (WAVE_CYCLES = WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY)
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/$K/SQ -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/SQ.json 2> $O/$K/SQ.err
    timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -f csv -d $O/$K/WAVE -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/WAVE.json 2> $O/$K/WAVE.err
    timeout 400 rocprofv3 --pmc VALUBusy MemUnitStalled -f csv -d $O/$K/BUSY -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/BUSY.json 2> $O/$K/BUSY.err
    timeout 400 rocprofv3 --pmc TCC_BUSY_avr TA_BUSY_avr GRBM_GUI_ACTIVE -f csv -d $O/$K/MEM -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/MEM.json 2> $O/$K/MEM.err
    timeout 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE -f csv -d $O/$K/LDS -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/LDS.json 2> $O/$K/LDS.err
}
WL=${1:-all}
want() { [ "$WL" = all ] || case ",$WL," in *",$1,"*) true;; *) false;; esac; }  # WL: all | one name | a comma-separated list
want hsvfilter && run hsvfilter --steps 30
want hsv1080p && run hsv1080p --workload hsv1080p --steps 30
want colorlut_natural && run colorlut_natural --workload colorlut --content natural --steps 20
want colorlut_random && run colorlut_random --workload colorlut --content random --steps 10
want videofx && run videofx --workload videofx --steps 50
want videocompare_blockhash && run videocompare_blockhash --workload videocompare --hash-algo blockhash --steps 40
want videocompare_dssim && run videocompare_dssim --workload videocompare --hash-algo dssim --steps 6
want hsvfilter_rgb && run hsvfilter_rgb --workload hsvfilter_rgb --steps 20
want hsvdetector_rgb && run hsvdetector_rgb --workload hsvdetector_rgb --steps 20
if want calib; then
    mkdir -p $O/calib
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $C -f csv -d $O/calib/$C -o pmc -- $REPO/tools/probe_pmc_calib.bin > $O/calib/$C.log 2> $O/calib/$C.err
    done
fi
# the direct-dispatch lane (round 6): the one-thread leg of tools/exp_direct_lane.py (64 frames in rotation) -- kernel trace, FETCH_SIZE, WRITE_SIZE
if want lane; then
    mkdir -p $O/lane
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/lane/trace -o trace -- python3 $REPO/tools/exp_direct_lane.py 64 > $O/lane/trace.log 2> $O/lane/trace.err
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $C -f csv -d $O/lane/$C -o pmc -- python3 $REPO/tools/exp_direct_lane.py 64 > $O/lane/$C.log 2> $O/lane/$C.err
    done
    python3 - $O/lane <<'PY' > $O/lane_summary.txt 2>&1
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
print("# the direct-dispatch lane under rocprofv3 (tools/exp_direct_lane.py 64: one thread, one 4K RGBA frame per call, 64 frames = 2.1 GB in rotation)")
print("# NOTE: the profiler's queue interception serialises and slows the lane's dispatches -- the durations below are not the unprofiled ones;")
print("#       what this pass is for is the HBM traffic per dispatch (66.36 MB algorithmic per frame)")
for ln in open(d + "/trace.log"):
    if "fps" in ln: print("# under trace:", ln.strip())
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hsvfilter" in r.get("Kernel_Name", ""):
                acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 2:]
        kb = sum(v) / len(v)
        print(f"{k}: {c} avg {kb:.1f} KB over {len(v)} dispatches -> {kb * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6:.2f} MB per frame" + (" (x 2: the guide's correction for wide coalesced reads)" if c == "FETCH_SIZE" else ""))
PY
    cat $O/lane_summary.txt
fi
cd $REPO
python3 tools/r6_traffic.py $O > $O/summary.txt 2>&1
cp $O/summary.txt $O/traffic.json $O/lane_summary.txt $REPO/gpurun_out/ 2>/dev/null  # the two results, beside the raw files
# (the raw per-dispatch files are large: what comes back is the summary, traffic.json and the small csv; never re-run r6_traffic.py on the
# pruned directory -- its result would miss the pruned workloads)
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
cat $O/summary.txt | head -150
