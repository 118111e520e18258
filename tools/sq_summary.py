#!/usr/bin/env python3
"""Reduce a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass to per-kernel means:
duration, effective clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and matrix-pipe utilisation (MFMA-busy cycles summed over
the 1024 SIMDs / elapsed cycles).   usage: sq_summary.py <counter_collection.csv> <out.json>

The elapsed-cycle figures are only meaningful when the counter window and the kernel's timestamps describe the same interval.
For kernels of a few tens of microseconds they do not (the counters run from before the dispatch to after its drain): the implied
clock then comes out ABOVE what the part can run at (round 4: 2.7 - 4.9 GHz for every config-2 kernel on a 2.4 GHz chip).  Such a
row gets `clock_valid: false`, no clock and no elapsed-normalised busy fraction; what it keeps is the ratio that needs no
elapsed time: MFMA-busy SIMD cycles per busy-CU SIMD cycle (SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES))."""
import csv, json, re, sys
from collections import defaultdict
MAX_CLOCK_GHZ = 2.45          # MI355X peak engine clock 2.4 GHz (MI355X_MICROARCH.md) + measurement slack
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
    clock = cyc / us / 1e3 if us else None
    valid = clock is not None and clock <= MAX_CLOCK_GHZ
    busy_cu = c.get("SQ_BUSY_CU_CYCLES", 0.0)
    row = {"launches": n[k], "mean_us_under_pmc": round(us, 1), "total_ms": round(dur[k] / 1e3, 1), "clock_valid": valid,
           "mfma_busy_per_busy_cu_simd_cycle": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy_cu), 3) if busy_cu else None}
    if valid:
        row.update({"cycles_per_launch": round(cyc), "effective_clock_ghz": round(clock, 3),
                    "mfma_busy_frac_of_simd_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc, 3) if cyc else None,
                    "cu_busy_frac": round(busy_cu / 256.0 / cyc, 3) if cyc else None})
    else:
        row["note"] = (f"implied clock {clock:.2f} GHz > {MAX_CLOCK_GHZ}: the counter window is longer than the kernel's timestamps - "
                       "no clock / elapsed-normalised figure for this kernel") if clock else "no duration"
    out[k] = row
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
