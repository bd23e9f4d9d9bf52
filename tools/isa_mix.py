#!/usr/bin/env python3
"""Instruction mix of every kernel in a device-only assembly file (hipcc --cuda-device-only -S): total, VALU, accumulation-
register moves, scalar / vector memory, waits - to see what a register plan costs before going to the GPU."""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
name, counts = None, None
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, counts = m.group(1), collections.Counter()
        continue
    if name is None:
        continue
    s = line.strip()
    if s.startswith("s_endpgm"):
        tot = sum(counts.values())
        grp = collections.Counter()
        for op, v in counts.items():
            g = ("accvgpr" if "accvgpr" in op else "valu" if op.startswith("v_") else "smem" if op.startswith("s_load") else
                 "waitcnt" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else
                 "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "scratch" if op.startswith("scratch_") else
                 "lds" if op.startswith("ds_") else "other")
            grp[g] += v
        print(name[:110])
        print("   total", tot, dict(grp))
        print("   top:", ", ".join(f"{k} {v}" for k, v in counts.most_common(12)))
        name = None
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    counts[s.split()[0]] += 1
