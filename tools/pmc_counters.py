#!/usr/bin/env python3
"""Per-kernel averages of the counters found in rocprofv3 counter-collection CSVs (one --pmc pass each).

    python tools/pmc_counters.py OUTDIR pass1.csv pass2.csv ...   ->  OUTDIR/summary.json + a table on stdout

Counter values are per dispatch summed over all XCDs / SEs as rocprofv3 reports them; durations come from the same rows.
Derived figures (when their inputs are present): valu_busy = SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CU_CYCLES-like denominators
are NOT formed here - only raw means and two robust ratios: VALU instructions per wave and the share of wave cycles spent
waiting on any instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for path in sys.argv[2:]:
        seen = set()
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if "qp::" not in k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                key = (path, row.get("Dispatch_Id"))
                if key not in seen and row.get("End_Timestamp"):
                    seen.add(key)
                    d = dur[k]
                    d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    d[1] += 1
    summary = {}
    for k, counters in acc.items():
        m = {c: v[0] / v[1] for c, v in counters.items()}
        m["avg_us"] = dur[k][0] / max(dur[k][1], 1) / 1e3
        if "SQ_WAVES" in m and "SQ_INSTS_VALU" in m:
            m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
            m["wait_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
            m["valu_active_share_of_wave_cycles"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_LDS" in m and "SQ_WAVE_CYCLES" in m:
            m["lds_active_share_of_wave_cycles"] = m["SQ_ACTIVE_INST_LDS"] / m["SQ_WAVE_CYCLES"]
        summary[k] = m
    json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
    for k, m in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"]):
        print(f"{k[:100]}\n    " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))


if __name__ == "__main__":
    main()
