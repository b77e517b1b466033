set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
for g in 8 12 16 20 24; do
echo "== groups per frame $g"
MVFX_CD_GROUPS=$g python tools/bench_kernels.py colordetect 2>/dev/null | grep "16 frames"
done | tee gpurun_out/r3k/colordetect_groups.txt
echo "== default"
python tools/bench_kernels.py colordetect 2>/dev/null | tee gpurun_out/r3k/colordetect_default.txt
for s in 1 2; do
MVFX_ELEMENT_STREAMS=$s python tools/bench_gst_pipeline.py --branches 1 --n1 100 --n2 600 > gpurun_out/r3k/gst_branch1_streams$s.txt 2>&1
tail -3 gpurun_out/r3k/gst_branch1_streams$s.txt
MVFX_ELEMENT_STREAMS=$s python tools/bench_gst_pipeline.py --branches 16 --n1 100 --n2 600 > gpurun_out/r3k/gst_branch16_streams$s.txt 2>&1
tail -3 gpurun_out/r3k/gst_branch16_streams$s.txt
done
MVFX_ELEMENT_STREAMS=2 timeout 1200 python -m pytest tests/test_gst_pipelines_gpu.py tests/test_gst_inprocess_gpu.py tests/test_gst_leaks_gpu.py -x -q 2>&1 | tail -6
MVFX_ELEMENT_STREAMS=2 python tools/bench_gst_pipeline.py 2>&1 | tail -3 | tee gpurun_out/r3k/gst_chain_streams2.txt
MVFX_ELEMENT_STREAMS=1 python tools/bench_gst_pipeline.py 2>&1 | tail -3 | tee gpurun_out/r3k/gst_chain_streams1.txt
