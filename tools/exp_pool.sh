for args in "--batch 8 --pool 48" "--batch 16 --pool 24" "--batch 32 --pool 12" "--batch 4 --pool 96" "--batch 1 --pool 384"; do
  python bench.py --no-cpu-baseline --steps 1500 --warmup 500 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$args', 'fps=%.0f' % d['value'], 'kernel_ms=%.4f' % d['roofline']['avg_launch_ms'], 'frac=%.3f' % d['roofline']['frac'])"
done
