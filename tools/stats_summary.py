#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv with demangled short kernel names and a per-pass column.
usage: stats_summary.py <csv> [passes]
`passes` = how many passes of the workload the capture holds - warm-up passes INCLUDED (`bench.py --steps 2 --warmup 1` is 3
passes; round 3 divided by the timed steps only and printed a column that summed to 1.5 steps).  Without the argument it is
taken from the capture itself: the int16 cast runs exactly once per bench step, a whole ODE solve calls cfg_axpy NFE times."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r"_GLOBAL__N_1(\d\d)(\w+)", n)
    return m.group(2)[: int(m.group(1))] if m else re.sub(r"\(anonymous namespace\)::|void ", "", n)[:40]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else None
how = "given"
if passes is None:
    for r in rows:
        if "wav_to_int16" in r["Name"]:
            passes, how = float(r["Calls"]), "calls of wav_to_int16_kernel (once per bench step)"
            break
if passes is None:
    for r in rows:
        if "cfg_axpy" in r["Name"]:
            passes, how = float(r["Calls"]) / 32.0, "calls of cfg_axpy_kernel / 32 NFE"
            break
if passes is None:
    passes, how = 1.0, "unknown: totals of the whole capture"
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# passes in the capture: {passes:g} ({how}); sum of kernel time per pass {tot / passes / 1e6:.1f} ms")
for r in rows[:18]:
    print(f"{short(r['Name']):36s} calls/pass {float(r['Calls'])/passes:8.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  per-pass {float(r['TotalDurationNs'])/passes/1e6:7.2f} ms  {float(r['Percentage']):5.1f}%")
