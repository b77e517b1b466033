#!/bin/bash
# tools/ab_headline.sh: the headline leg of bench.py (200 timed steps, no side legs) with the shipped library and every gst-plugin-rs_amd/build_ab/lib_*.so,
# non-temporal and cached loads, two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for so in "" $R/gst-plugin-rs_amd/build_ab/lib_*.so; do
  for streaming in 1 0; do
    name=$( [ -z "$so" ] && echo shipped || basename $so .so )
    MVFX_LIB=$so python3 $R/bench.py --no-cpu-baseline --no-verify --steps 200 --warmup 50 --stream-threads 0 --content-sweep 0 --only-configs none --streaming $streaming 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', 'streaming=$streaming', 'fps=%.0f' % d['value'], 'frac=%.3f' % d['roofline']['frac'], 'frac_kernel=%s' % d['roofline'].get('frac_kernel'))"
  done
done
done
# the hsv side legs (sub-lines) with every library: tools/ab_headline.sh prints them when AB_LEGS is set (comma separated --only-configs keys)
if [ -n "${AB_LEGS:-}" ]; then
for rep in 1 2; do
for so in "" $R/gst-plugin-rs_amd/build_ab/lib_*.so; do
    name=$( [ -z "$so" ] && echo shipped || basename $so .so )
    MVFX_LIB=$so python3 $R/bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 5 --stream-threads 0 --content-sweep 0 --only-configs $AB_LEGS 2>/dev/null | \
        python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if 'sub' in d and d.get('value') is not None and d['sub'] in '$AB_LEGS'.split(','): print('$name', d['sub'], 'value=%.0f' % d['value'], 'p50=%s' % d.get('value_p50'), 'frac_kernel=%s' % d.get('frac_kernel'))
"
done
done
fi
