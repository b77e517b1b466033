set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
python -m pytest tests/test_videofx_gpu.py -x -q -k colordetect 2>&1 | tail -3
O=gpurun_out/r3c/colordetect_sweep.txt
echo "# tools/bench_kernels.py colordetect with MVFX_CD_GROUPS (groups per frame) / MVFX_CD_IF (16-byte loads in flight per lane)" > $O
for G in default; do for I in default; do echo "## groups=$G in_flight=$I" >> $O; python tools/bench_kernels.py colordetect 2>/dev/null >> $O; done; done
for G in 96 128 192 256; do for I in 4 8 12; do echo "## single-frame lines: groups=$G in_flight=$I" >> $O; MVFX_CD_GROUPS=$G MVFX_CD_IF=$I python tools/bench_kernels.py colordetect 2>/dev/null | grep -v "16 frames" >> $O; done; done
for G in 16 24 32 48 64; do for I in 4 8 12; do echo "## batched lines: groups per frame=$G in_flight=$I" >> $O; MVFX_CD_GROUPS=$G MVFX_CD_IF=$I python tools/bench_kernels.py colordetect 2>/dev/null | grep "16 frames" >> $O; done; done
python3 - $O <<'PY'
import json, sys
hdr = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("##"): hdr = line
    elif line.startswith("{"):
        d = json.loads(line); print(hdr, "|", d["kernel"][22:], d["ms_per_call"] * 1e3, "us", d["frac_of_8TBs"])
PY
bash tools/trace_kernels.sh colordetect tools/bench_kernels.py colordetect > gpurun_out/r3c/trace_colordetect.txt 2>&1; cat gpurun_out/r3c/trace_colordetect.txt
