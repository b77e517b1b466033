set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 1500 python tools/exp_rot2_crash.py 2 2>&1 | tee gpurun_out/r3l/rot2_crash.txt | tail -120
