#!/usr/bin/env python3
"""Reduce a rocprofv3 `--kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES` pass to per-kernel ratios
that need no elapsed time (the counter window of a 5-35 us kernel is longer than its timestamps, tools/sq_summary.py): VALU-active and
LDS-active quad-cycles (x 4 = SIMD cycles, MI355X_MICROARCH.md) per busy-CU SIMD cycle (4 x SQ_BUSY_CU_CYCLES), and waves resident per
busy SIMD cycle.   usage: valu_summary.py <counter_collection.csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float); seen = set()
for r in csv.DictReader(open(sys.argv[1], newline="")):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for k in sorted(acc, key=lambda k: -dur[k])[:10]:
    c = {name: v / n[k] for name, v in acc[k].items()}
    busy = 4.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0)
    out[k] = {"launches": n[k], "mean_us_under_pmc": round(dur[k] / n[k], 1),
              "valu_active_per_busy_simd_cycle": round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / busy, 3) if busy else None,
              "lds_active_per_busy_simd_cycle": round(4.0 * c.get("SQ_ACTIVE_INST_LDS", 0.0) / busy, 3) if busy else None,
              "waves_per_busy_simd_cycle": round(4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / busy, 2) if busy else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
