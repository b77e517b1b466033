for args in "--batch 16 --pool 24" "--batch 1 --pool 1" "--batch 2 --pool 1" "--batch 4 --pool 1" "--batch 1 --pool 8" "--batch 32 --pool 12"; do
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$args', 'fps=%.0f' % d['value'], 'kernel_ms=%.4f' % d['roofline']['avg_launch_ms'], 'frac=%.3f' % d['roofline']['frac'])"
done
