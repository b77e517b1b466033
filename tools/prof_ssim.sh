#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_ssim; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SSIM_ONLY=8K rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o t -- python3 $REPO/tools/bench_kernels.py ssim > $OUT/run.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/prof_ssim")
for f in glob.glob(out + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls=%6s avg_us=%10.1f total_ms=%9.1f %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $OUT -name "*.csv" -size +2M -delete
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/prof_ssim")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-30s grid=%s,%s calls=%5d avg_us=%9.1f total_ms=%8.1f" % (k[0], k[1], k[2], n, t / n / 1e3, t / 1e6))
PY
