#!/bin/bash
# five back-to-back runs of the driver's bench command (CPU baselines skipped after the first): run-to-run spread of the final line and of
# the sub-lines' values.  On the GPU box: bash tools/bench_repeat.sh > gpurun_out/bench_repeat.txt
for i in 1 2 3 4 5; do
  t0=$(date +%s)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) 2>/dev/null > /tmp/rep_$i.txt
  python3 - "$i" "$(( $(date +%s) - t0 ))" <<'PY'
import json, sys
i, wall = sys.argv[1], sys.argv[2]
lines = [json.loads(l) for l in open(f"/tmp/rep_{i}.txt") if l.startswith("{")]
final = lines[-1]
subs = {l["sub"]: l for l in lines[:-1] if "sub" in l}
r = final["roofline"]
print(f"run {i} ({wall} s): value {final['value']:.0f} fps frac {r['frac']:.4f} (kernel {r.get('frac_kernel', 0):.4f}) p50 {final['config'].get('value_p50', 0):.0f} | " +
      " ".join(f"{k} {v.get('value', 0):.0f}" for k, v in subs.items()))
PY
done
