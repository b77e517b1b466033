#!/usr/bin/env python3
"""tools/valu_cost.py <file.s> [kernel substring] [--blocks] -- the VALU instruction mix of each kernel in a device assembly file
(hipcc -S --cuda-device-only), priced with the issue costs tools/probes/valu_issue.hip measured on an MI355X
(profiles/r4/valu_issue_probe.txt):

  fast   1.07 ns per wave64 instruction per SIMD (2 cycles): v_fma/fmac/add/sub/mul f32 with VGPR, inline or literal operands and
         any modifier, v_add/sub_u32, v_and/or/xor, v_lshrrev_b32, v_mov_b32, and a lone v_cndmask_b32_e32
  slow   1.75 ns on a second pipe that overlaps with `fast` issue: everything else -- min/max/med3, cvt, floor/fract, perm, bfe,
         lshlrev, lshl_or, and_or, add3, mad/mul_u24, mul_lo, sad, cmp, cndmask_e64, SDWA, DPP, packed 16-bit, f64, and a fast
         instruction with an SGPR operand
  pk32   1.88 ns on the FIRST pipe (no overlap with `fast`): v_pk_{mul,add,fma}_f32
  trans  3.46 ns, overlaps with nothing: v_rcp/rsq/sqrt/exp/log/sin/cos
  back-to-back v_cndmask_b32_e32: 9.5 ns each (listed separately)

Static counts, not execution counts: a kernel's loops are weighted only with --blocks (per basic block listing)."""
import collections
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_cndmask_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_rcp_f64",
         "v_rsq_f64", "v_sqrt_f64"}
NS = {"fast": 1.07, "slow": 1.75, "pk32": 1.88, "trans": 3.46, "cnd_chain": 9.5}
PK32 = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"}


def classify(line, prev_cls_op):
    m = re.match(r"(v_\w+)", line)
    if not m:
        return None, None
    op = m.group(1)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    suffix = op[len(base):]
    operands = line[len(op):].split(";")[0]
    if base in TRANS:
        return "trans", op
    if base in PK32:
        return "pk32", op
    if base == "v_cndmask_b32":
        if suffix == "_e32":
            return ("cnd_chain" if prev_cls_op == "v_cndmask_b32_e32" else "fast"), op
        return "slow", op
    if base in FAST and suffix not in ("_sdwa", "_dpp"):
        if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc\b|\bexec\b|\bttmp", operands):
            return "slow", op + " (sgpr)"
        return "fast", op
    return "slow", op


def kernels(text):
    for m in re.finditer(r"^(_Z\w+|[A-Za-z_]\w*):\s*(?:;.*)?\n((?:.*\n)*?)\s+s_endpgm", text, re.M):
        yield m.group(1), m.group(2)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    per_block = "--blocks" in sys.argv
    text = open(args[0]).read()
    want = args[1] if len(args) > 1 else ""
    for name, body in kernels(text):
        if want not in name or ".LBB" in name:
            continue
        cls = collections.Counter()
        ops = collections.defaultdict(collections.Counter)
        blocks = []
        cur = ["entry", collections.Counter()]
        prev = None
        other = collections.Counter()
        for raw in body.split("\n"):
            line = raw.strip()
            lab = re.match(r"(\.LBB\w+):", line)
            if lab:
                blocks.append(cur)
                cur = [lab.group(1), collections.Counter()]
                continue
            c, op = classify(line, prev)
            if c is None:
                m = re.match(r"(s_\w+|ds_\w+|buffer_\w+|global_\w+|flat_\w+)", line)
                if m:
                    other[m.group(1).split("_")[0]] += 1
                    if not m.group(1).startswith("s_"):
                        prev = None
                continue
            prev = op
            cls[c] += 1
            ops[c][op] += 1
            cur[1][c] += 1
        blocks.append(cur)
        n = sum(cls.values())
        if not n:
            continue
        issue = NS["fast"] * (cls["fast"] + cls["slow"]) + NS["pk32"] * cls["pk32"]
        slow = NS["slow"] * cls["slow"]
        extra = NS["trans"] * cls["trans"] + NS["cnd_chain"] * cls["cnd_chain"]
        print(f"{name[:110]}")
        print(f"  VALU {n}: fast {cls['fast']}  slow {cls['slow']}  pk32 {cls['pk32']}  trans {cls['trans']}  back-to-back cndmask_e32 {cls['cnd_chain']}"
              f"   | other: {dict(other)}")
        print(f"  static price: issue {issue:.0f} ns, slow pipe {slow:.0f} ns -> max {max(issue, slow):.0f} + serial {extra:.0f} ns"
              f"  ({'slow-pipe' if slow > issue else 'issue'} limited)")
        for c in ("slow", "pk32", "trans", "fast"):
            top = ", ".join(f"{o} {k}" for o, k in ops[c].most_common(14))
            print(f"    {c}: {top}")
        if per_block:
            for lab, cc in blocks:
                t = sum(cc.values())
                if t >= 8:
                    print(f"    block {lab}: {t} VALU  fast {cc['fast']} slow {cc['slow']} pk32 {cc['pk32']} trans {cc['trans']} chain {cc['cnd_chain']}")


if __name__ == "__main__":
    main()
