#!/bin/bash
# tools/ab_lane_libs.sh: tools/exp_direct_lane.py (one thread, one call per 4K frame: two HIP streams / the lane, cached / MVFX_OPT_NONTEMPORAL) and the
# hsv1080p side leg with the shipped library and every gst-plugin-rs_amd/build_ab/lib_*.so, two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for so in "" $R/gst-plugin-rs_amd/build_ab/lib_*.so; do
    name=$( [ -z "$so" ] && echo shipped || basename $so .so )
    echo "== $name"
    MVFX_LIB=$so python3 $R/tools/exp_direct_lane.py 12 2>&1 | grep "fps"
    MVFX_LIB=$so python3 $R/bench.py --no-cpu-baseline --no-verify --steps 200 --warmup 50 --stream-threads 0 --content-sweep 0 --only-configs hsv1080p 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if d.get('sub') == 'hsv1080p': print('   hsv1080p %.0f (p50 %s)' % (d['value'], d.get('value_p50')))
"
done
done
