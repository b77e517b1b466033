#!/bin/bash
# power / clock samples while the headline kernel runs (is the kernel power-limited?)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -vE "^=|^$" | head -30
python $REPO/bench.py --no-cpu-baseline --steps 30000 --warmup 10 > /tmp/bench_long.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3; do
  echo "--- sample $i (hsvfilter running)"
  rocm-smi --showpower --showclocks --showuse -t 2>&1 | grep -E "Power|sclk|mclk|fclk|busy|junction" | head -12
  sleep 2
done
wait $BP
cat /tmp/bench_long.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fps=%.0f' % d['value'], 'kernel_ms=%.4f' % d['roofline']['avg_launch_ms'])"
