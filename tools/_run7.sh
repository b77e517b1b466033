set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
python bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dssim', d['value'], d['unit'], d['roofline']['frac'], d['roofline']['step_us'], d['config']['last_distance'])"
# colorlut LDS cell pitch A/B on this box (MVFX_LIB = the other build)
for rep in 1 2; do
for lib in gst-plugin-rs_amd/libmi355vfx.so build_ab/libmi355vfx_pitch6.so; do
MVFX_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --workload colorlut --content natural --steps 60 --warmup 10 --no-cpu-baseline --stream-threads 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('colorlut natural $lib', round(d['value']), round(d['roofline']['frac_kernel'], 4), d['roofline']['step_us']['p50'])"
done; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --content-sweep 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('headline', round(d['value']), round(d['roofline']['frac_kernel'], 4), 'streams', round(c['other_launch_model']['value']), round(c['other_launch_model']['frac_wall'], 4), 'combined', round(c['combined_launch_model']['value']), round(c['combined_launch_model']['frac_wall'], 4), c['combined_launch_model']['frames_per_combined_launch'], c['combined_launch_model']['repetitions_frames_per_sec'])"
python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 2>&1 | tail -3
