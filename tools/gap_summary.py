#!/usr/bin/env python3
"""Where the device idles: from a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv), the gaps between the end of one kernel and the
start of the next (all queues merged), largest first, with the kernels on either side - and the idle total by gap size.
usage: gap_summary.py <kernel_trace.csv> [top=25] [skip_ms=0: ignore everything before this many ms after the first kernel; negative: keep only the last |skip_ms| ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
def short(n):
    m = re.search(r"_GLOBAL__N_1(\d\d)(\w+)", n)
    return m.group(2)[: int(m.group(1))] if m else re.sub(r"\(anonymous namespace\)::|void ", "", n)[:44]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_first = ev[0][0] + skip * 1e6 if skip >= 0 else max(e[1] for e in ev) + skip * 1e6        # (negative: only the LAST |skip_ms| ms)
ev = [e for e in ev if e[0] >= t_first]
gaps, end, prev = [], ev[0][1], ev[0][2]
busy = 0
for s, e, n in ev:
    if s > end:
        gaps.append((s - end, prev, n, (end - ev[0][0]) / 1e6))
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, n
span = end - ev[0][0]
idle = sum(g[0] for g in gaps)
print(f"# {len(ev)} kernels over {span / 1e6:.1f} ms: idle {idle / 1e6:.1f} ms ({100.0 * idle / span:.1f} %) in {len(gaps)} gaps")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print(f"#   gaps of {lo / 1e3:g}-{hi / 1e3:g} us: {len(sel):6d}, {sum(sel) / 1e6:8.2f} ms")
for g in sorted(gaps, reverse=True)[:top]:
    print(f"{g[0] / 1e3:9.1f} us at {g[3]:9.2f} ms  after {g[1]:40s} before {g[2]}")
