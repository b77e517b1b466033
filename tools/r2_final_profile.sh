#!/bin/bash
# round-2 evidence run on the GPU box: headline bench line, the other config lines, rocprofv3 kernel trace + PMC traffic of the
# headline kernel, kernel trace of the per-stream launch model (4 threads) for the overlap picture
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r2final
mkdir -p $O
cd $REPO
python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --warmup 5 --launch-model streams --no-cpu-baseline > $O/bench_headline_streams.json 2>> $O/bench_headline.err
for wl in hsv1080p videofx videocompare; do python bench.py --workload $wl --steps 200 --warmup 20 2>/dev/null >> $O/bench_configs.jsonl; done
python bench.py --workload videofx --element-streams 2 --steps 200 --warmup 20 2>/dev/null >> $O/bench_configs.jsonl
python bench.py --workload videocompare --hash-algo dssim --steps 20 --warmup 3 2>/dev/null >> $O/bench_configs.jsonl
for c in natural random smpte; do python bench.py --workload colorlut --content $c --steps 40 --warmup 10 2>/dev/null >> $O/bench_configs.jsonl; done
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 300 --warmup 50 --no-cpu-baseline --stream-threads 0 --content-sweep 0"
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o trace -- python3 $REPO/bench.py $ARGS > $O/bench_under_trace.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -f csv -d $O/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -f csv -d $O/pmc_sq -o pmc -- python3 $REPO/bench.py $ARGS > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc VALUBusy MemUnitBusy -f csv -d $O/pmc_busy -o pmc -- python3 $REPO/bench.py $ARGS > /dev/null 2> $O/pmc_busy.err
rocprofv3 --kernel-trace -f csv -d $O/trace_streams -o t4 -- python3 $REPO/tools/bench_streams.py --threads 4 --launches 100 > $O/trace_streams.log 2>&1
cd $REPO
python3 tools/summarize_prof.py $O > $O/summary.txt 2>&1
python3 - $O <<'PY' > $O/streams_overlap.txt
import csv, glob, sys, statistics
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace_streams/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hsvfilter" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"][:90]))
rows.sort()
tail = rows[-400:]
t0 = tail[0][0]
dur = [(e - s) / 1e3 for s, e, *_ in tail]
conc = [sum(1 for q in tail if q[0] <= r[0] < q[1]) for r in tail]
span = (tail[-1][1] - t0) / 1e3
print("rocprofv3 --kernel-trace -- python3 tools/bench_streams.py --threads 4 --launches 100  (4 host threads x own HIP stream x single-frame launches, 3840x2160 RGBA)")
print("kernel:", tail[0][4])
print(f"last 400 dispatches (the timed run): duration mean {statistics.mean(dur):.2f} us, median {statistics.median(dur):.2f}, min {min(dur):.2f}, max {max(dur):.2f}")
print(f"span {span:.1f} us = {span / len(tail):.2f} us per frame (profiled); dispatches running when a dispatch starts (itself included): mean {statistics.mean(conc):.2f}")
print("queue stream   start_us     end_us   dur_us")
for s, e, q, st, _ in tail[100:140]:
    print(f"{q:>5} {st:>6} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}")
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/summary.txt | head -60; cat $O/streams_overlap.txt | head -12; cat $O/bench_configs.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['metric'], round(d['value'], 1), d['unit'], round(d['roofline']['frac'], 3), d['config']['workload'][:60])"
