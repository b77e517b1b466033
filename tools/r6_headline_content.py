#!/usr/bin/env python3
"""tools/r6_headline_content.py <dir> -- condenses tools/r6_headline_content.sh's rocprofv3 passes: one block per frame content."""
import csv
import glob
import json
import sys
from collections import defaultdict

KERNEL = "hsvfilter4_typed_kernel"
BYTES = 16 * 2 * 3840 * 2160 * 4


def main():
    root = sys.argv[1]
    print("# headline kernel (16 x 3840x2160 RGBA per launch, in place, 1 061.68 MB algorithmic) by frame content; rocprofv3, separate --pmc passes")
    rows = {}
    for kind in ("videotestsrc", "natural", "random"):
        d = f"{root}/{kind}"
        r = {}
        try:
            doc = json.loads([ln for ln in open(f"{d}/trace.json") if ln.startswith("{")][-1])
            r["bench_value_under_trace"] = doc["value"]
            r["bench_step_us"] = doc["ms_per_step"] * 1e3
            steps = doc["steps"]
        except Exception as e:  # noqa: BLE001
            r["error"] = f"no bench line: {e}"
            steps = 30
        tr = glob.glob(f"{d}/trace/**/*kernel_trace.csv", recursive=True)
        if tr:
            k = [x for x in csv.DictReader(open(tr[0])) if KERNEL in x["Kernel_Name"]]
            k.sort(key=lambda x: int(x["Start_Timestamp"]))
            dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in k]
            r["launches"] = len(dur)
            r["kernel_us_all"] = sum(dur) / len(dur)
            r["kernel_us_timed"] = sum(dur[-steps:]) / steps  # the timed steps are the last ones of the run
        ctr = defaultdict(list)
        for f in glob.glob(f"{d}/p*/**/*counter_collection.csv", recursive=True):
            per = defaultdict(list)
            for x in csv.DictReader(open(f)):
                if KERNEL in x.get("Kernel_Name", ""):
                    per[x["Counter_Name"]].append(float(x["Counter_Value"]))
            for c, v in per.items():
                ctr[c] = v[len(v) // 2:]  # the second half of the run: clocks settled
        r["ctr"] = {c: sum(v) / len(v) for c, v in ctr.items() if v}
        rows[kind] = r
    for kind, r in rows.items():
        c = r.get("ctr", {})
        t = r.get("kernel_us_timed")
        print(f"\n## {kind}")
        if "error" in r:
            print("  ", r["error"])
        if t:
            print(f"  kernel time: {t:.2f} us over the timed launches ({r['kernel_us_all']:.2f} all {r['launches']}), bench step {r.get('bench_step_us', 0):.2f} us, "
                  f"{r.get('bench_value_under_trace', 0):.0f} fps under trace; frac of 8 TB/s (kernel) {BYTES / (t * 1e-6) / 8e12:.4f}")
        g = c.get("GRBM_GUI_ACTIVE")
        if g and t:
            print(f"  GRBM_GUI_ACTIVE {g:.4g} per launch -> {g / t / 1e3:.3f} GHz if it counts one clock domain for the kernel's duration ({g / 8 / t / 1e3:.3f} if summed over 8 XCDs)")
        if c.get("SQ_INSTS_VALU"):
            print(f"  SQ_INSTS_VALU {c['SQ_INSTS_VALU']:.5g} per launch = {c['SQ_INSTS_VALU'] * 64 / (16 * 3840 * 2160):.2f} per pixel; SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES', 0):.4g}; "
                  f"SQ_ACTIVE_INST_VALU {c.get('SQ_ACTIVE_INST_VALU', 0):.4g}")
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            print(f"  wave states: waiting {c.get('SQ_WAIT_ANY', 0) / w:.3f}, issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / w:.3f}, active {c.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} "
                  f"(LDS {c.get('SQ_ACTIVE_INST_LDS', 0) / w:.4f}, VMEM {c.get('SQ_ACTIVE_INST_VMEM', 0) / w:.4f}) of SQ_WAVE_CYCLES {w:.4g}")
        if c.get("SQ_INSTS_LDS"):
            print(f"  LDS: SQ_INSTS_LDS {c['SQ_INSTS_LDS']:.5g}, SQ_LDS_IDX_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE', 0):.5g}, SQ_LDS_BANK_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT', 0):.5g} "
                  f"({c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} of the active cycles)")
        if c.get("TCC_BUSY_avr") is not None and c.get("GRBM_GUI_ACTIVE"):
            print(f"  TCC_BUSY_avr {c['TCC_BUSY_avr']:.4g}, TA_BUSY_avr {c.get('TA_BUSY_avr', 0):.4g} per launch")
        if c.get("VALUBusy") is not None:
            print(f"  VALUBusy {c['VALUBusy']:.1f} %, MemUnitStalled {c.get('MemUnitStalled', 0):.2f} %")
        if c.get("FETCH_SIZE") is not None:
            # FETCH_SIZE / WRITE_SIZE are in KB; a wide coalesced 16 B / lane read stream reports half its bytes (tools/probe_pmc_calib.hip, the guide's HBM section)
            rd, wr = c["FETCH_SIZE"] * 1024 * 2, c.get("WRITE_SIZE", 0) * 1024
            print(f"  HBM: FETCH_SIZE x 2 = {rd / 1e6:.1f} MB, WRITE_SIZE = {wr / 1e6:.1f} MB per launch -> {(rd + wr) / BYTES:.3f} x algorithmic")
    print("\n# reading: see profiles/r6/README.md")


if __name__ == "__main__":
    main()
