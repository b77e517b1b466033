set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 900 python tools/exp_rot2_crash.py 2 2>&1 | tail -30 | tee gpurun_out/r3m/rot2_crash_after_fix.txt
timeout 900 python -m pytest tests/test_combiner_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -5
for g in 8 12 16 24 32; do
echo "== groups per frame $g"
MVFX_CD_GROUPS=$g python tools/bench_kernels.py colordetect 2>/dev/null | grep "16 frames"
done | tee gpurun_out/r3m/colordetect_groups.txt
for s in 1 2; do
MVFX_ELEMENT_STREAMS=$s python tools/bench_gst_pipeline.py --branches 1 --n1 1000 --n2 6000 > gpurun_out/r3m/gst_branch1_streams$s.txt 2>&1
tail -2 gpurun_out/r3m/gst_branch1_streams$s.txt
done
MVFX_ELEMENT_STREAMS=2 python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 > gpurun_out/r3m/gst_branch16_streams2.txt 2>&1
tail -2 gpurun_out/r3m/gst_branch16_streams2.txt
