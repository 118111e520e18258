#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv with demangled short kernel names.  usage: stats_summary.py <csv> [n_steps]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:16]:
    n = r["Name"]
    m = re.search(r"_GLOBAL__N_1(\d\d)(\w+)", n)
    n = m.group(2)[: int(m.group(1))] if m else re.sub(r"\(anonymous namespace\)::|void ", "", n)[:40]
    print(f"{n:36s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/steps/1e6:7.1f} ms  {float(r['Percentage']):5.1f}%")
