#!/usr/bin/env python3
"""Reduce a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass over bench.py to
per-kernel means: duration, effective clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and matrix-pipe utilisation
(MFMA-busy cycles summed over the 1024 SIMDs / elapsed cycles).   usage: sq_summary.py <counter_collection.csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float); seen = set()
for r in csv.DictReader(open(sys.argv[1], newline="")):
    m = re.search(r"_GLOBAL__N_1(\d\d)(\w+)", r["Kernel_Name"])
    k = m.group(2)[: int(m.group(1))] if m else re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for k in sorted(acc, key=lambda k: -dur[k])[:8]:
    c = {name: v / n[k] for name, v in acc[k].items()}
    us = dur[k] / n[k]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out[k] = {"launches": n[k], "mean_us_under_pmc": round(us, 1), "total_ms": round(dur[k] / 1e3, 1),
              "cycles_per_launch": round(cyc), "effective_clock_ghz": round(cyc / us / 1e3, 3) if us else None,
              "mfma_busy_frac_of_simd_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc, 3) if cyc else None,
              "cu_busy_frac": round(c.get("SQ_BUSY_CU_CYCLES", 0.0) / 256.0 / cyc, 3) if cyc else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
