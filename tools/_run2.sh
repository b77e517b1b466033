set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python -m pytest tests/test_videofx_gpu.py -x -q -k colordetect 2>&1 | tail -15
python -m pytest tests/test_gst_pipelines_gpu.py -x -q -k "tee or fences or refreshed or colordetect" 2>&1 | tail -8
python tools/bench_kernels.py colordetect 2>&1 | tee gpurun_out/r3b/bench_colordetect.txt
bash tools/trace_kernels.sh colordetect > gpurun_out/r3b/trace_colordetect.txt 2>&1; tail -12 gpurun_out/r3b/trace_colordetect.txt
bash tools/r3_traffic.sh hsvfilter 2>&1 | grep -A12 "## hsvfilter"
