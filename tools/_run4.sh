set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
python -m pytest tests/test_videofx_gpu.py -x -q -k colordetect 2>&1 | tail -3
timeout 900 python tools/exp_ssim32_error.py 2>&1 | tee gpurun_out/r3d/ssim32_error.txt
python bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dssim', d['value'], d['unit'], d['roofline']['frac'], d['roofline']['step_us'], d['config']['last_distance'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 --no-cpu-baseline --pct-steps 0 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3d/trace/*kernel_stats.csv | cut -c1-200 | head -12
