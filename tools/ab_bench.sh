#!/bin/bash
# tools/ab_bench.sh -- build kernel variants (extra -D flags) side by side and bench them on the GPU box.
#   tools/ab_bench.sh build  name1="-DFOO" name2="-DBAR=2" ...   (here, cross-compiles)
#   tools/ab_bench.sh run [bench args]                             (on the GPU box, via gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
AB=$ROOT/gst-plugin-rs_amd/build_ab
if [ "$1" = build ]; then
    shift
    rm -rf "$AB"; mkdir -p "$AB"
    for spec in "$@"; do
        name=${spec%%=*}; flags=${spec#*=}
        make -s -C "$ROOT/gst-plugin-rs_amd" OBJDIR="$AB/$name" OUT="$AB/lib_$name.so" EXTRA_HIPFLAGS="$flags" >/dev/null
        echo "built $name ($flags)"
    done
else
    shift || true
    for so in "$AB"/lib_*.so; do
        name=$(basename "$so" .so); name=${name#lib_}
        for rep in 1 2; do
            MVFX_LIB=$so python "$ROOT/bench.py" --no-cpu-baseline --steps 1500 --warmup 1500 "$@" | \
                python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', 'fps=%.0f' % d['value'], 'ms_per_step=%.4f' % d['ms_per_step'], 'frac=%.3f' % d['roofline']['frac'])"
        done
    done
fi
