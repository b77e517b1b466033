#!/usr/bin/env python3
"""tools/kernel_regs.py <file.s> [substring] -- VGPR / SGPR / LDS / spill counts per kernel from the amdhsa.kernels metadata of a
device assembly file (hipcc -S --cuda-device-only)."""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if want not in name:
        continue
    get = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(f"{name[:100]}: vgpr {get('vgpr_count')} sgpr {get('sgpr_count')} lds {get('group_segment_fixed_size')} "
          f"spill v{get('vgpr_spill_count')} s{get('sgpr_spill_count')} scratch {get('private_segment_fixed_size')}")
