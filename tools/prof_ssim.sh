#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_ssim; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o t -- python3 $REPO/tools/bench_kernels.py ssim > $OUT/run.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/prof_ssim")
for f in glob.glob(out + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls=%6s avg_us=%10.1f total_ms=%9.1f %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $OUT -name "*.csv" -size +2M -delete
