set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 1200 python -m pytest tests/test_combiner_gpu.py tests/test_gst_pipelines_gpu.py tests/test_gst_inprocess_gpu.py tests/test_gst_leaks_gpu.py -x -q 2>&1 | tail -6
python tools/exp_stream_rotation.py 2>/dev/null | tee gpurun_out/r3j/stream_rotation.txt
for win in 30 10; do
MVFX_COMBINE_WINDOW_US=$win python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --content-sweep 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; cm = c['combined_launch_model']; f = cm['fenced_entry']
print('window $win: headline', round(d['value']), round(d['roofline']['frac_kernel'], 4), 'streams', round(c['other_launch_model']['value']), round(c['other_launch_model']['frac_wall'], 4), 'combined', round(cm['value']), round(cm['frac_wall'], 4), round(cm['frames_per_combined_launch'], 1), 'fenced', round(f['value']), round(f['frac_wall'], 4), round(f['frames_per_combined_launch'], 1), f['repetitions_frames_per_sec'])"
done
python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 2>&1 | tail -2 | tee gpurun_out/r3j/gst_branches.json
MVFX_ELEMENT_STREAMS=2 python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 2>&1 | tail -1 | tee gpurun_out/r3j/gst_branches_rot2.json
