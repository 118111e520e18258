#!/usr/bin/env python3
"""Print per-kernel means of every counter in a rocprofv3 *_counter_collection.csv (+ mean duration in us)."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for row in csv.DictReader(open(sys.argv[1], newline="")):
    m = re.search(r"(\w+_kernel)(<\d+>)?", row["Kernel_Name"])
    k = (m.group(1) + (m.group(2) or "")) if m else row["Kernel_Name"][:40]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    k = f'{k}[grid={row["Grid_Size"]}]'
    a = acc[k][row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
    d = (row["Dispatch_Id"])
    if d not in seen:
        seen.add(d)
        dur[k][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3; dur[k][1] += 1
for k in acc:
    print(k, "calls", dur[k][1], "mean_us", round(dur[k][0] / max(dur[k][1], 1), 1))
    for c, (s, n) in sorted(acc[k].items()):
        print("    %-28s %.4g" % (c, s / n))
