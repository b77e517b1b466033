#!/usr/bin/env python3
"""tools/r4_traffic.py <dir> -- condenses tools/r4_traffic.sh's rocprofv3 passes (round 3's tool + the VALU counters: issued VALU
instructions per step and the VALUBusy / MemUnitBusy averages, from which `bound` is set: "valu" when the duration-weighted VALUBusy of
the step's dominant kernel (largest share of the traced time) is above 70 %, else "hbm").

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
    if t:
        r = t.get("roofline", {})
        print(f"# bench line under the kernel trace: value {t.get('value'):.1f} {t.get('unit')}, avg step {r.get('avg_step_ms', r.get('avg_launch_ms'))} ms, "
              f"frac_kernel {r.get('frac_kernel', r.get('frac')):.4f}")
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
    wsum = sum(weights.get(k, 0.0) for k in busy if is_product(k)) or 1.0
    vb = sum(weights.get(k, 0.0) * sum(c["VALUBusy"]) / len(c["VALUBusy"]) for k, c in busy.items() if is_product(k) and "VALUBusy" in c) / wsum
    mb = sum(weights.get(k, 0.0) * sum(c["MemUnitBusy"]) / len(c["MemUnitBusy"]) for k, c in busy.items() if is_product(k) and "MemUnitBusy" in c) / wsum
    if busy:
        valu_busy, mem_busy = vb, mb
        for k, c in sorted(busy.items()):
            if is_product(k) and "VALUBusy" in c:
                print(f"{k}: VALUBusy {sum(c['VALUBusy']) / len(c['VALUBusy']):.1f} %  MemUnitBusy {sum(c.get('MemUnitBusy', [0])) / max(len(c.get('MemUnitBusy', [0])), 1):.1f} %")
        # the ceiling of the step is the ceiling of its DOMINANT kernel (largest share of the traced time)
        dom = max((k for k in busy if is_product(k) and "VALUBusy" in busy[k]), key=lambda k: weights.get(k, 0.0), default=None)
        if dom is not None:
            valu_busy = sum(busy[dom]["VALUBusy"]) / len(busy[dom]["VALUBusy"])
        print(f"# time-weighted over the product kernels: VALUBusy {vb:.1f} %, MemUnitBusy {mb:.1f} %; dominant kernel {dom}: VALUBusy {valu_busy:.1f} % "
              f"-> bound = {'valu' if valu_busy > 70 else 'hbm'}")
    if ok and lines["FETCH_SIZE"]:
        cfg = lines["FETCH_SIZE"].get("config", {})
        units = cfg.get("units_per_step_per_gpu", cfg.get("frames_per_step_per_gpu"))
        algo = lines["FETCH_SIZE"].get("roofline", {}).get("bytes_per_step", lines["FETCH_SIZE"].get("roofline", {}).get("bytes_per_launch"))
        hbm = total["FETCH_SIZE"] + total["WRITE_SIZE"]
        print(f"# HBM bytes per step: read {total['FETCH_SIZE'] / 1e6:.2f} MB + written {total['WRITE_SIZE'] / 1e6:.2f} MB = {hbm / 1e6:.2f} MB; "
              f"algorithmic {algo / 1e6:.2f} MB; ratio {hbm / algo:.3f}")
        traffic[wl] = {"hbm_bytes_per_step": hbm, "read_bytes_per_step": total["FETCH_SIZE"], "written_bytes_per_step": total["WRITE_SIZE"],
                       "units_per_step": units, "algorithmic_bytes_per_step": algo, "kernels": per_kernel,
                       "valu_insts_per_step": valu_per_step, "valu_busy_pct": valu_busy, "mem_unit_busy_pct": mem_busy,
                       "bound": ("valu" if valu_busy is not None and valu_busy > 70 else "hbm") if valu_busy is not None else None,
                       "source": f"committed rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of bench.py --workload {wl}; "
                                 "tools/r4_traffic.sh, profiles/r4/traffic_summary.txt); counters scaled by the known-byte calibration of the same "
                                 "run (tools/probe_pmc_calib.hip)"}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
