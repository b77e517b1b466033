set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3i/bench_driver_command.json 2> gpurun_out/r3i/bench.err ) 2> gpurun_out/r3i/bench.time
tail -3 gpurun_out/r3i/bench.time; tail -3 gpurun_out/r3i/bench.err
python3 - gpurun_out/r3i/bench_driver_command.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", round(d["value"]), "frac_kernel", round(r["frac_kernel"], 4), "frac_wall", round(r["frac_wall"], 4), "ceil", {k: round(v["GBs"]) for k, v in r["ceilings"].items()}, "traffic ratio", r["traffic_over_algorithmic"])
c = d["config"]
print(" streams", round(c["other_launch_model"]["value"]), "combined", round(c["combined_launch_model"]["value"]), c["combined_launch_model"]["frames_per_combined_launch"])
for k, v in c.get("other_configs", {}).items():
    if "error" in v: print(k, "ERROR", v["error"]); continue
    r = v["roofline"]; cb = v.get("cpu_baseline", {})
    print(k, round(v["value"], 1), v["unit"], "frac_kernel", round(r["frac_kernel"], 4), "frac_wall", round(r["frac_wall"], 4), "p50", r["step_us"] and round(r["step_us"]["p50"], 1),
          "cpu1", cb.get("value") and round(cb["value"], 3), "cpuN", cb.get("all_cores", {}).get("value") and round(cb["all_cores"]["value"], 2), "traffic/alg", r["traffic_over_algorithmic"] and round(r["traffic_over_algorithmic"], 3))
PY
bash tools/r3_traffic.sh all > gpurun_out/r3i/traffic.log 2>&1; cp gpurun_out/r3traffic/summary.txt gpurun_out/r3i/traffic_summary.txt; cp gpurun_out/r3traffic/traffic.json gpurun_out/r3i/traffic.json
grep -E "^## |# HBM bytes" gpurun_out/r3i/traffic_summary.txt
bash tools/prof_counters.sh ssim32 ssim32_level --workload videocompare --hash-algo dssim > gpurun_out/r3i/prof_ssim32.log 2>&1; cp gpurun_out/ctr_ssim32/summary.txt gpurun_out/r3i/ssim32_counters.txt
grep -E "VALUBusy|SQ_INSTS_VALU |kernel_stats|level_kernel.*[0-9]+,[0-9]" gpurun_out/r3i/ssim32_counters.txt | cut -c1-200 | head
