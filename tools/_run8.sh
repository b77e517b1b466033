set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests/test_combiner_gpu.py tests/test_convert_gpu.py tests/test_ssim_gpu.py tests/test_gst_pipelines_gpu.py -x -q 2>&1 | tail -6
python bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dssim', d['value'], d['unit'], d['roofline']['frac'], d['roofline']['step_us'], d['config']['last_distance'])"
for win in 30 10 60; do
MVFX_COMBINE_WINDOW_US=$win python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --content-sweep 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('window $win: headline', round(d['value']), round(d['roofline']['frac_kernel'], 4), 'streams', round(c['other_launch_model']['value']), round(c['other_launch_model']['frac_wall'], 4), 'combined', round(c['combined_launch_model']['value']), round(c['combined_launch_model']['frac_wall'], 4), c['combined_launch_model']['frames_per_combined_launch'], c['combined_launch_model']['repetitions_frames_per_sec'])"
done
python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 2>&1 | tail -2
timeout 600 python tools/exp_ssim32_error.py 2>&1 | tee gpurun_out/r3h/ssim32_error.txt | cut -c1-150
