#!/bin/bash
# clock of the VALU microbench vs the real kernel: GRBM_GUI_ACTIVE / (8 XCD * duration)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/exp_clock; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o t -- $REPO/tools/hsv_valu_bench.bin > $OUT/run1.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -f csv -d $OUT/p -o p -- $REPO/tools/hsv_valu_bench.bin > $OUT/run2.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/exp_clock")
dur = collections.defaultdict(list)
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in dur:
    d = max(dur[k])
    line = f"{k:60s} dur_max={d/1e3:9.1f}us"
    for c, v in pmc.get(k, {}).items():
        line += f" {c}={max(v):.4g}"
    if "GRBM_GUI_ACTIVE" in pmc.get(k, {}):
        line += f"  clock={max(pmc[k]['GRBM_GUI_ACTIVE'])/8/d:.3f} GHz"
    print(line)
PY
