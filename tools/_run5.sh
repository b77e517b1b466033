set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 900 python tools/exp_ssim32_error.py 2>&1 | tee gpurun_out/r3e/ssim32_error.txt | cut -c1-140
python bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dssim', d['value'], d['unit'], d['roofline']['frac'], d['roofline']['step_us'], d['config']['last_distance'])"
bash tools/r3_traffic.sh videocompare_dssim 2>&1 | tail -25
