#!/bin/bash
# round-3 evidence run on the GPU box: the driver's bench command (headline + configs 2-5 in one line), then the per-workload
# rocprofv3 kernel traces and PMC traffic passes (tools/r3_traffic.sh)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r3
mkdir -p $O
cd $REPO
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err ) 2> $O/bench_headline.time
tail -3 $O/bench_headline.time
python3 - $O/bench_headline.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", round(d["value"]), d["unit"], "frac_kernel", round(r["frac_kernel"], 4), "frac_wall", round(r["frac_wall"], 4), "ceiling", {k: round(v["GBs"]) for k, v in r["ceilings"].items()}, "launch_us", r["launch_us"])
print(" streams", round(d["config"]["other_launch_model"]["value"]), round(d["config"]["other_launch_model"]["frac_wall"], 4))
print(" cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("all_cores", {}).get("value"))
for k, v in d["config"].get("other_configs", {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    r = v["roofline"]
    c = v.get("cpu_baseline", {})
    print(k, round(v["value"], 1), v["unit"], "frac_kernel", round(r["frac_kernel"], 4), "frac_wall", round(r["frac_wall"], 4), "step_us", r["step_us"] and {q: round(r["step_us"][q], 1) for q in ("p10", "p50", "p90")},
          "cpu1", c.get("value") and round(c["value"], 3), "cpuN", c.get("all_cores", {}).get("value") and round(c["all_cores"]["value"], 2), "traffic", r["traffic"])
PY
bash tools/r3_traffic.sh "${1:-all}"
