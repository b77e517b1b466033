#!/bin/bash
# tools/lane_chain_stats.sh: thread-separated lane chains (tools/soak_lane_chain.py's) with MVFX_LANE_STATS=1 -- what the lane's acquires did --, pool
# of 12 blocks, lane on and off (MVFX_DIRECT_DISPATCH=0), two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 - <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from tests import cubes, gst_env
launch = gst_env.tool("gst-launch-1.0")
tmp = tempfile.mkdtemp(prefix="lanestats_")
cube = os.path.join(tmp, "look.cube")
open(cube, "w").write(cubes.analytic_3d(33))
hip = "video/x-raw(memory:HIPMemory)"
n = int(os.environ.get("N", "100000"))
chains = {"hsvfilter ! queue ! colorlut ! queue ! colorlut": f"hsvfilter ! queue max-size-buffers=3 ! colorlut location={cube} ! queue max-size-buffers=3 ! colorlut location={cube}"}
for rep in range(2):
    for name, chain in chains.items():
        for pool in os.environ.get("POOLS", "12").split(","):
            for lane in os.environ.get("MODES", "1,0").split(","):
                pipe = f"hiptestsrc num-buffers={n} refresh=false ! {hip},format=RGBA,width=3840,height=2160 ! {chain} ! fakesink sync=false"
                t0 = time.perf_counter()
                r = gst_env.run([launch, "-q"] + pipe.split(), tmp, timeout=300, extra_env={"MVFX_HIP_POOL_MIN": pool, "MVFX_LANE_STATS": "1", "MVFX_DIRECT_DISPATCH": "1" if lane == "nodiscourage" else lane,
                                                                                              "MVFX_DIRECT_DISCOURAGE": "0" if lane == "nodiscourage" else "1",
                                                                                              **{k: v for k, v in os.environ.items() if k.startswith("GPU_") or k.startswith("HIP_") or k.startswith("HSA_")}})
                dt = time.perf_counter() - t0
                print(f"{name}, pool {pool}, lane {lane}: {n / dt:.0f} fps, rc {r.returncode}\n   " + "\n   ".join(l for l in r.stdout.splitlines() if "mvfx lane" in l), flush=True)
PY
