#!/usr/bin/env python3
"""tools/r6_traffic.py <dir> -- condenses tools/r6_traffic.sh's rocprofv3 passes.

Round 5:
  * per kernel, beside rocprofv3's all-launch average: the mean / median / min over the TIMED launches only -- the last `steps` steps of
    the bench run (settle and warm-up launches run while the clocks are still ramping and pull the all-launch average 1-12 % above the
    driver-timed step) -- and the check that the timed kernel time of a step does not exceed the step the same run's bench line reports;
  * `bound` no longer rests on one counter: MemUnitBusy reads 0.0 for every kernel on this stack and is dropped.  Evidence per workload:
    VALUBusy (SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles: passes 100 % on this part, read it as "busy"), the wave-state split
    (SQ_WAIT_ANY / SQ_WAVE_CYCLES: waves parked at s_waitcnt or a barrier), MemUnitStalled, the L2's TCC_BUSY_avr / GRBM_GUI_ACTIVE, the
    texture addresser's TA_BUSY_avr / GRBM_GUI_ACTIVE, the LDS array (SQ_LDS_IDX_ACTIVE / CU-cycles) and the HBM bytes the kernel moved per
    second of its own timed duration.  bound = "valu" when VALUBusy of the step's dominant kernel >= 70 %; otherwise "hbm" (the only other
    roofline this path has: nothing here is a contraction) -- with `bound_note` saying whether the memory system is actually busy
    (traffic rate >= 0.55 of the peak or L2 busy >= 70 %) or the step is short of both ceilings (launch / drain / dependent chains).

Per workload: per-kernel dispatch counts and average duration (kernel trace), per-kernel average FETCH_SIZE / WRITE_SIZE (KB,
separate PMC passes), and the HBM bytes per bench step = sum over the product kernels of (FETCH_SIZE x 1024 x read factor +
WRITE_SIZE x 1024 x write factor) / steps the bench executed in that pass.  The factors come from the calibration probe
(tools/probe_pmc_calib.hip, known byte counts per access pattern) of the same run: gfx950's FETCH_SIZE tallies 128-byte
requests at 64 bytes (MI355X_MICROARCH.md), so wide reads are doubled; which pattern a kernel reads with is named below.
Writes <dir>/traffic.json (copy to profiles/traffic.json) and prints a text summary (copy to profiles/r4/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
PRODUCT = ("hsvfilter", "hsvdetector", "hsv_from_frame", "colorlut", "copy_planes", "colordetect", "blockhash", "ssim", "vsample",
           "hsample", "gray_kernel", "i420", "overlay_blend")
# read pattern of each product kernel (which calibration row applies)
READ_PATTERN = [("hsvfilter4_typed", "read16_nt"), ("hsvfilter", "read16_nt"), ("hsvdetector", "read16"), ("colorlut", "read16"),
                ("copy_planes", "read16"), ("colordetect_hist", "read4_stride40"), ("colordetect", "read16"), ("blockhash", "read16"),
                ("ssim", "read16"), ("", "read16")]
WRITE_PATTERN = [("hsvfilter", "write16_nt"), ("", "write16")]


def short(name):
    """mvfx::(anonymous namespace)::hsvfilter4_typed_kernel<1, 2, true>(args) -> hsvfilter4_typed_kernel<1, 2, true>"""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mvfx::", "")
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:90]


def is_product(name):
    return any(p in name for p in PRODUCT)


def counters(path):
    """{kernel: [values]} of one counter_collection.csv"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row.get("Kernel_Name", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def bench_line(path):
    try:
        for line in open(path):
            line = line.strip()
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


def timed_launches(base, line):
    """{kernel: {"n", "per_step", "mean_ns", "median_ns", "min_ns"}} over the dispatches of the bench run's timed region: the last
    `steps` steps (bench.py runs settle, warm-up, then exactly `steps` timed steps; --pct-steps 0 in the profiling command)"""
    if not line:
        return {}
    steps, executed = line.get("steps"), line.get("config", {}).get("steps_executed")
    if not steps or not executed:
        return {}
    per = defaultdict(list)
    for f in glob.glob(os.path.join(base, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                try:
                    per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
                except (KeyError, ValueError):
                    pass
    res = {}
    for k, v in per.items():
        if not is_product(k):
            continue
        v.sort()
        per_step = len(v) / executed
        n = int(round(per_step * steps))
        if n < 1:
            continue
        d = sorted(x[1] for x in v[-n:])
        res[k] = {"n": n, "per_step": per_step, "mean_ns": sum(d) / len(d), "median_ns": d[len(d) // 2], "min_ns": d[0]}
    return res


# ---- calibration ---------------------------------------------------------------------------------------------------
calib = {}
cal_dir = os.path.join(out, "calib")
if os.path.isdir(cal_dir):
    known = 1 << 30
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, c in counters(os.path.join(cal_dir, ctr)).items():
            if "calib_" in k and ctr in c:
                v = c[ctr]
                calib.setdefault(k.replace("calib_", ""), {})[ctr] = sum(v) / len(v)
    print("# calibration (tools/probe_pmc_calib.bin): every kernel touches 1 GiB once; counters are KB per dispatch")
    for k, c in sorted(calib.items()):
        f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        print(f"{k:18s} FETCH_SIZE {f:12.0f} KB = {f * 1024 / known:6.3f} x touched   WRITE_SIZE {w:12.0f} KB = {w * 1024 / known:6.3f} x touched")


def factor(kind, pattern):
    """known bytes / counted bytes for that pattern; the guide's x2 / x1 when no calibration run is present"""
    row = calib.get(pattern)
    ctr = "FETCH_SIZE" if kind == "read" else "WRITE_SIZE"
    if row and row.get(ctr, 0) > 0:
        return (1 << 30) / (row[ctr] * 1024)
    return 2.0 if kind == "read" else 1.0


def pattern_of(name, table):
    for sub, pat in table:
        if sub in name:
            return pat
    return table[-1][1]


# ---- workloads -------------------------------------------------------------------------------------------------------
traffic = {}
for wl in sorted(d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d)) and d != "calib"):
    base = os.path.join(out, wl)
    print(f"\n## {wl}")
    lines = {k: bench_line(os.path.join(base, k + ".json")) for k in ("trace", "FETCH_SIZE", "WRITE_SIZE")}
    for f in glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for i, row in enumerate(csv.reader(open(f))):
            if i == 0 or is_product(row[0]):
                print(",".join([short(row[0])] + row[1:8]))
    t = lines["trace"]
    timed = timed_launches(base, t)
    timed_step_ns = None
    if t:
        r = t.get("roofline", {})
        print(f"# bench line under the kernel trace: value {t.get('value'):.1f} {t.get('unit')}, ms_per_step {t.get('ms_per_step')}, avg step between HIP events "
              f"{r.get('avg_step_ms', r.get('avg_launch_ms'))} ms, frac_kernel {r.get('frac_kernel', r.get('frac')):.4f}")
    if timed:
        print("# timed launches only (the last `steps` steps of that run): kernel, launches, per step, mean / median / min ns")
        timed_step_ns = 0.0
        for k, d in sorted(timed.items()):
            print(f"timed {k}: n={d['n']} per_step={d['per_step']:.2f} mean={d['mean_ns']:.0f} median={d['median_ns']:.0f} min={d['min_ns']:.0f}")
            timed_step_ns += d["mean_ns"] * d["per_step"]
        step_ns = (t.get("ms_per_step") or 0) * 1e6
        if step_ns:
            print(f"# reconcile: kernel time of a timed step {timed_step_ns:.0f} ns (kernels on two streams overlap: a sum can exceed the step) vs "
                  f"ms_per_step {step_ns:.0f} ns of the same run -> ratio {timed_step_ns / step_ns:.3f}")
    total = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    per_kernel = {}
    ok = True
    for ctr, kind, table in (("FETCH_SIZE", "read", READ_PATTERN), ("WRITE_SIZE", "write", WRITE_PATTERN)):
        line = lines[ctr]
        steps = None
        if line:
            steps = line.get("config", {}).get("steps_executed")
        if not steps:
            ok = False
            print(f"# {ctr}: no bench line / steps_executed in that pass")
            continue
        for k, c in sorted(counters(os.path.join(base, ctr)).items()):
            if not is_product(k) or ctr not in c:
                continue
            v = c[ctr]
            pat = pattern_of(k, table)
            fac = factor(kind, pat)
            nbytes = sum(v) * 1024 * fac
            total[ctr] += nbytes / steps
            per_kernel.setdefault(k, {})[ctr] = {"dispatches": len(v), "avg_KB": sum(v) / len(v), "pattern": pat, "factor": round(fac, 4),
                                                 "bytes_per_step": nbytes / steps}
            print(f"{k}: {ctr} n={len(v)} avg={sum(v) / len(v):.0f} KB  x{fac:.3f} ({pat})  -> {nbytes / steps / 1e6:.2f} MB per step ({steps} steps in the pass)")
    # ---- VALU: issued instructions per step, busy percentages (weighted by SQ_BUSY_CYCLES-free dispatch count: plain mean over dispatches
    # of the product kernels, weighted by each kernel's share of the traced time)
    valu_per_step, valu_busy, mem_busy = None, None, None
    sq_line = bench_line(os.path.join(base, "SQ.json"))
    sq_steps = sq_line.get("config", {}).get("steps_executed") if sq_line else None
    sq = counters(os.path.join(base, "SQ"))
    if sq_steps:
        tot = 0.0
        for k, c in sq.items():
            if is_product(k) and "SQ_INSTS_VALU" in c:
                tot += sum(c["SQ_INSTS_VALU"])
                print(f"{k}: SQ_INSTS_VALU n={len(c['SQ_INSTS_VALU'])} avg={sum(c['SQ_INSTS_VALU']) / len(c['SQ_INSTS_VALU']):.4g}"
                      + (f"  SQ_ACTIVE_INST_VALU avg={sum(c['SQ_ACTIVE_INST_VALU']) / len(c['SQ_ACTIVE_INST_VALU']):.4g} (quad-cycles)" if "SQ_ACTIVE_INST_VALU" in c else "")
                      + (f"  GRBM_GUI_ACTIVE avg={sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE']):.4g}" if "GRBM_GUI_ACTIVE" in c else ""))
        valu_per_step = tot / sq_steps if tot else None
    busy = counters(os.path.join(base, "BUSY"))
    weights = {}
    for f in glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for i, row in enumerate(csv.reader(open(f))):
            if i and is_product(row[0]):
                weights[short(row[0])] = float(row[2])
    dom = max((k for k in weights), key=lambda k: weights[k], default=None) # the step's dominant kernel: largest share of the traced time
    avg = lambda c, name: (sum(c[name]) / len(c[name])) if name in c and c[name] else None
    evidence = {}
    if dom is not None:
        b = busy.get(dom, {})
        evidence["valu_busy_pct"] = avg(b, "VALUBusy")
        evidence["mem_unit_stalled_pct"] = avg(b, "MemUnitStalled")
        wv = counters(os.path.join(base, "WAVE")).get(dom, {})
        wc = avg(wv, "SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("SQ_WAIT_ANY", "wave_wait_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "wave_active_frac")):
                if avg(wv, name) is not None:
                    evidence[key] = avg(wv, name) / wc
        mm = counters(os.path.join(base, "MEM")).get(dom, {})
        gui = avg(mm, "GRBM_GUI_ACTIVE")
        if gui:
            for name, key in (("TCC_BUSY_avr", "l2_busy_frac"), ("TA_BUSY_avr", "ta_busy_frac")):
                if avg(mm, name) is not None:
                    evidence[key] = avg(mm, name) / gui
        ld = counters(os.path.join(base, "LDS")).get(dom, {})
        gui = avg(ld, "GRBM_GUI_ACTIVE")
        if gui and avg(ld, "SQ_LDS_IDX_ACTIVE") is not None:
            # LDS-array cycles summed over the chip per cycle of the launch: "so many of the 256 LDS arrays busy on average"
            evidence["lds_arrays_busy_of_256"] = avg(ld, "SQ_LDS_IDX_ACTIVE") / gui
            if avg(ld, "SQ_LDS_BANK_CONFLICT") is not None and avg(ld, "SQ_LDS_IDX_ACTIVE"):
                evidence["lds_conflict_share"] = avg(ld, "SQ_LDS_BANK_CONFLICT") / avg(ld, "SQ_LDS_IDX_ACTIVE")
        print(f"# evidence for `bound`, dominant kernel {dom}: " + ", ".join(f"{k} {v:.3g}" for k, v in evidence.items() if v is not None))
    valu_busy = evidence.get("valu_busy_pct")
    mem_busy = None
    if ok and lines["FETCH_SIZE"]:
        cfg = lines["FETCH_SIZE"].get("config", {})
        units = cfg.get("units_per_step_per_gpu", cfg.get("frames_per_step_per_gpu"))
        algo = lines["FETCH_SIZE"].get("roofline", {}).get("bytes_per_step", lines["FETCH_SIZE"].get("roofline", {}).get("bytes_per_launch"))
        hbm = total["FETCH_SIZE"] + total["WRITE_SIZE"]
        rate = hbm / timed_step_ns / 8000.0 if timed_step_ns else None  # bytes per ns = GB/s; over 8 TB/s
        busy_mem = (rate is not None and rate >= 0.55) or (evidence.get("l2_busy_frac") or 0) >= 0.70
        bound = ("valu" if valu_busy >= 70 else "hbm") if valu_busy is not None else None
        note = None
        if bound == "hbm":
            note = "memory system busy" if busy_mem else "short of both ceilings (launch / drain / dependent chains)"
        elif bound == "valu":
            note = "VALU busy" + ("; memory system busy as well (the ridge)" if busy_mem else "")
        if rate is not None:
            print(f"# HBM traffic rate over the timed kernel time: {rate:.3f} of 8 TB/s -> bound = {bound} ({note})")
        print(f"# HBM bytes per step: read {total['FETCH_SIZE'] / 1e6:.2f} MB + written {total['WRITE_SIZE'] / 1e6:.2f} MB = {hbm / 1e6:.2f} MB; "
              f"algorithmic {algo / 1e6:.2f} MB; ratio {hbm / algo:.3f}")
        traffic[wl] = {"hbm_bytes_per_step": hbm, "read_bytes_per_step": total["FETCH_SIZE"], "written_bytes_per_step": total["WRITE_SIZE"],
                       "units_per_step": units, "algorithmic_bytes_per_step": algo, "kernels": per_kernel,
                       "valu_insts_per_step": valu_per_step, "valu_busy_pct": valu_busy, "bound": bound, "bound_note": note,
                       "bound_evidence": {k: v for k, v in evidence.items() if v is not None},
                       "timed_kernel_ns_per_step": timed_step_ns,
                       "timed_launches": {k: {kk: round(vv, 1) for kk, vv in d.items()} for k, d in timed.items()},
                       "source": f"committed rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of bench.py --workload {wl}; "
                                 "tools/r6_traffic.sh, profiles/r6/traffic_summary.txt); counters scaled by the known-byte calibration of the same "
                                 "run (tools/probe_pmc_calib.hip)"}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
