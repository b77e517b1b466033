#!/bin/bash
# tools/prof_kernels.sh <tag> <bench_kernels selector...> -- on the GPU box: rocprofv3 kernel trace of
# tools/bench_kernels.py for the selected kernels; per-kernel stats land in gpurun_out/prof_<tag>/stats.txt
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python3 "$REPO/tools/bench_kernels.py" "$@" > "$OUT/bench_under_trace.txt" 2> "$OUT/trace.err"
cd "$REPO"
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cut -c1-220 "$STATS" | head -14 > "$OUT/stats.txt"; else echo "no kernel_stats.csv produced" > "$OUT/stats.txt"; tail -5 "$OUT/trace.err" >> "$OUT/stats.txt"; fi
cat "$OUT/stats.txt"
find "$OUT" -name "*.csv" -size +2M -delete
