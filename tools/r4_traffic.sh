#!/bin/bash
# tools/r4_traffic.sh -- on the GPU box: per BASELINE workload a rocprofv3 kernel trace and SEPARATE --pmc FETCH_SIZE and --pmc
# WRITE_SIZE passes (no tracing domain beside --pmc) of `bench.py --workload <w>`, plus the PMC calibration probe
# (tools/probe_pmc_calib.bin: known byte counts per access pattern).  tools/r4_traffic.py turns the result into
# profiles/traffic.json (what bench.py quotes as roofline.traffic) and profiles/r4/traffic_*.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r4traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--full 1 --no-cpu-baseline --pct-steps 0 --settle-seconds 0.05 --stream-threads 0 --content-sweep 0 --other-configs 0 --noise-sweep 0 --gst-pipeline 0 --warmup 2"
run() { # <key> <bench args...>
    local K=$1; shift
    mkdir -p $O/$K
    timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/$K/trace -o trace -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/trace.json 2> $O/$K/trace.err
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $C -f csv -d $O/$K/$C -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/$C.json 2> $O/$K/$C.err
    done
    # round 4: which ceiling binds -- issued VALU instructions (the VALU fraction of bench.py) and the busy percentages, separate passes
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/$K/SQ -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/SQ.json 2> $O/$K/SQ.err
    timeout 400 rocprofv3 --pmc VALUBusy MemUnitBusy -f csv -d $O/$K/BUSY -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/BUSY.json 2> $O/$K/BUSY.err
}
WL=${1:-all}
want() { [ "$WL" = all ] || [ "$WL" = "$1" ]; }
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
cd $REPO
python3 tools/r4_traffic.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/summary.txt | head -150
