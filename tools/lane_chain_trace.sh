#!/bin/bash
# tools/lane_chain_trace.sh: rocprofv3 kernel trace of a device-only chain (CHAIN=..., CUBE = the 33^3 .cube; default: the three-thread chain
# hsvfilter ! queue ! colorlut ! queue ! colorlut), 4K, pool of POOL (12) blocks, with the lane on and off: per kernel name count / median duration, and
# the span of the whole trace per frame
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d /tmp/lanetrace.XXXX)
python3 -c "
import sys; sys.path.insert(0, '$R')
from tests import cubes
open('$T/look.cube', 'w').write(cubes.analytic_3d(33))"
export PATH=/opt/conda/bin:$PATH GST_PLUGIN_SYSTEM_PATH=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH=$R/gst-plugin-rs_amd/gst-plugins GST_REGISTRY=$T/registry.bin GST_REGISTRY_FORK=no
export MVFX_HIP_POOL_MIN=${POOL:-12} MVFX_LANE_STATS=1
N=${N:-30000}
CHAIN=${CHAIN:-"hsvfilter ! queue max-size-buffers=3 ! colorlut location=CUBE ! queue max-size-buffers=3 ! colorlut location=CUBE"}
PIPE="hiptestsrc num-buffers=$N refresh=false ! video/x-raw(memory:HIPMemory),format=RGBA,width=3840,height=2160 ! ${CHAIN//CUBE/$T/look.cube} ! fakesink sync=false"
/opt/conda/bin/gst-launch-1.0 -q $PIPE > /dev/null 2>&1   # registry
cd /tmp && export TMPDIR=/tmp
for lane in 1 0; do
  export MVFX_DIRECT_DISPATCH=$lane
  rm -rf $T/tr$lane
  s=$(date +%s.%N)
  rocprofv3 --kernel-trace -f csv -d $T/tr$lane -o t -- /opt/conda/bin/gst-launch-1.0 -q $PIPE > $T/log$lane.txt 2>&1
  e=$(date +%s.%N)
  echo "== lane $lane: $(python3 -c "print(f'{$N / ($e - $s):.0f} fps under the tracer (process wall)')")"; grep "mvfx lane" $T/log$lane.txt
  python3 - $T/tr$lane $N <<'PY'
import csv, glob, sys
from collections import defaultdict
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
n = int(sys.argv[2])
rows = list(csv.DictReader(open(fs[0])))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:56]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"   trace span {(t1 - t0) / 1e6:.1f} ms = {(t1 - t0) / 1e3 / n:.1f} us per frame")
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 100: continue
    d = sorted(e - s for s, e, _ in v)
    qs = sorted(set(q for _, _, q in v))
    print(f"   {k:<58} n={len(v):6d} duration p50 {d[len(d)//2]/1e3:6.2f} us p90 {d[len(d)*9//10]/1e3:6.2f} us  sum {sum(d)/1e6:7.1f} ms  queues {len(qs)}")
PY
done
rm -rf $T
