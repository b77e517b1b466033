set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
python -m pytest tests/test_colorlut_gpu.py -x -q 2>&1 | tail -2
for c in natural smpte random; do python bench.py --workload colorlut --content $c --steps 40 --warmup 10 --no-cpu-baseline --stream-threads 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('colorlut $c', round(d['value']), d['unit'], round(d['roofline']['frac_kernel'], 4), d['roofline']['step_us'])"; done
python tools/bench_kernels.py colorlut 2>/dev/null | cut -c1-200
bash tools/prof_counters.sh ssim32 ssim32_level --workload videocompare --hash-algo dssim > gpurun_out/r3f/prof_ssim32.log 2>&1; cat gpurun_out/ctr_ssim32/summary.txt | cut -c1-200
