set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 1500 python -m pytest tests/test_colorlut_gpu.py tests/test_combiner_gpu.py tests/test_abi.py -x -q 2>&1 | tail -6
python tools/bench_kernels.py baked 2>/dev/null | tee gpurun_out/r3n/colorlut_baked.txt
python tools/bench_kernels.py colordetect 2>/dev/null | tee gpurun_out/r3n/colordetect.txt
