#!/bin/bash
# tools/r6_headline_content.sh -- on the GPU box: the headline leg (16 x 4K RGBA per launch, hsvfilter4_typed_kernel) on the three frame contents
# (videotestsrc bars + snow, natural-like gradients + noise, uniform-random bytes): a rocprofv3 kernel trace and SEPARATE --pmc passes each (no tracing
# domain beside --pmc).  tools/r6_headline_content.py condenses them into gpurun_out/headline_content.txt: per content the timed-launch kernel time,
# the effective clock (GRBM_GUI_ACTIVE / kernel time), VALU instructions, the LDS sextant reads' bank conflicts, L2 / TA busy, the wave-state split,
# HBM bytes.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r6content
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--full 1 --no-cpu-baseline --no-verify --pct-steps 0 --stream-threads 0 --content-sweep 0 --other-configs 0 --gst-pipeline 0 --warmup 5 --steps 30"
for K in videotestsrc natural random; do
    mkdir -p $O/$K
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/$K/trace -o trace -- python3 $REPO/bench.py $COMMON --frame-content $K > $O/$K/trace.json 2> $O/$K/trace.err
    i=0
    for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "TCC_BUSY_avr TA_BUSY_avr GRBM_GUI_ACTIVE" \
             "VALUBusy MemUnitStalled" \
             "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $C -f csv -d $O/$K/p$i -o pmc -- python3 $REPO/bench.py $COMMON --settle-seconds 0.05 --frame-content $K > $O/$K/p$i.json 2> $O/$K/p$i.err
    done
done
cd $REPO
python3 tools/r6_headline_content.py $O | tee $REPO/gpurun_out/headline_content.txt
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
