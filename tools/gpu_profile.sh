#!/bin/bash
# tools/gpu_profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel-trace
# stats + separate PMC passes of bench.py; summaries land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 500 --warmup 100 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -f csv -d "$OUT/pmc_sq" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc VALUBusy MemUnitBusy -f csv -d "$OUT/pmc_busy" -o pmc -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_busy.err"
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files for the merge back
find "$OUT" -name "*.csv" -size +2M -delete
