#!/bin/bash
# tools/r3_traffic.sh -- on the GPU box: per BASELINE workload a rocprofv3 kernel trace and SEPARATE --pmc FETCH_SIZE and --pmc
# WRITE_SIZE passes (no tracing domain beside --pmc) of `bench.py --workload <w>`, plus the PMC calibration probe
# (tools/probe_pmc_calib.bin: known byte counts per access pattern).  tools/r3_traffic.py turns the result into
# profiles/traffic.json (what bench.py quotes as roofline.traffic) and profiles/r3/traffic_*.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r3traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --pct-steps 0 --settle-seconds 0.05 --stream-threads 0 --content-sweep 0 --other-configs 0 --warmup 2"
run() { # <key> <bench args...>
    local K=$1; shift
    mkdir -p $O/$K
    timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/$K/trace -o trace -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/trace.json 2> $O/$K/trace.err
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $C -f csv -d $O/$K/$C -o pmc -- python3 $REPO/bench.py $COMMON "$@" > $O/$K/$C.json 2> $O/$K/$C.err
    done
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
if want calib; then
    mkdir -p $O/calib
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $C -f csv -d $O/calib/$C -o pmc -- $REPO/tools/probe_pmc_calib.bin > $O/calib/$C.log 2> $O/calib/$C.err
    done
fi
cd $REPO
python3 tools/r3_traffic.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/summary.txt | head -150
