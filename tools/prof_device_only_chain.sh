#!/bin/bash
# device-only chain under rocprofv3 --hip-trace --kernel-trace --stats
R=$GRAFT_REPO_ROOT
python3 - <<'PY' > /tmp/pipe_cmd.txt
import os, sys, tempfile
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tests import cubes, gst_env
tmp = tempfile.mkdtemp()
cube = os.path.join(tmp, "look.cube")
open(cube, "w").write(cubes.analytic_3d(33))
e = gst_env.env(tmp)
for k in ("PATH", "GST_PLUGIN_SYSTEM_PATH", "GST_PLUGIN_PATH", "GST_REGISTRY", "GST_REGISTRY_FORK"):
    print(f"export {k}='{e[k]}'")
print(f"CUBE={cube}")
print(f"LAUNCH={gst_env.tool('gst-launch-1.0')}")
PY
source /tmp/pipe_cmd.txt
PIPE="hiptestsrc num-buffers=1500 ! video/x-raw(memory:HIPMemory),format=RGBx,width=3840,height=2160,framerate=30/1 ! hsvfilter hue-shift=45 ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA ! colorlut location=$CUBE ! fakesink sync=false"
time $LAUNCH -q $PIPE
cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats -f csv -d $R/gpurun_out/piped -o p -- $LAUNCH -q $PIPE > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
for name in ("hip_api_stats", "kernel_stats"):
    for f in glob.glob(f"gpurun_out/piped/**/*{name}.csv", recursive=True):
        print("==", name)
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows[:12]:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:9.1f} ms')
PY
find gpurun_out/piped -name "*.csv" -size +1M -delete
