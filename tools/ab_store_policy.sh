#!/bin/bash
# tools/ab_store_policy.sh: round 5's stores (gst-plugin-rs_amd/build_ab/lib_old_stores.so = -DMVFX_STORE_POLICY=0, the hsv side legs without
# MVFX_OPT_NONTEMPORAL as they ran then) against the shipped library, same box, interleaved, three rounds: the headline leg (200 timed steps) and the
# hsv side legs' sub-lines (tools/ab_objs.sh "hsv_typed_kernels hsv_kernels" 'old_stores=-DMVFX_STORE_POLICY=0' builds the library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OLD=$R/gst-plugin-rs_amd/build_ab/lib_old_stores.so
legs=hsv1080p,hsvfilter_rgb,hsvdetector_rgb
one() { # name lib side_nt
  MVFX_LIB=$2 MVFX_BENCH_SIDE_NT=$3 python3 $R/bench.py --no-cpu-baseline --no-verify --steps 200 --warmup 50 --stream-threads 0 --content-sweep 0 --only-configs $legs 2>/dev/null | python3 -c "
import sys, json
out = []
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if d.get('sub') in '$legs'.split(','): out.append('%s %.0f (p50 %s)' % (d['sub'], d['value'], d.get('value_p50')))
    elif 'roofline' in d: out.insert(0, 'headline %.0f = %.3f (kernel %.3f)' % (d['value'], d['roofline']['frac'], d['roofline'].get('frac_kernel') or 0))
print('$1:', ' | '.join(out))
"
}
for rep in 1 2 3; do
  one "round-5 stores" $OLD 0
  one "shipped       " "" 1
done
