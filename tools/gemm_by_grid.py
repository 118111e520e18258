#!/usr/bin/env python3
"""Mean duration of the large-problem GEMM per grid size from a rocprofv3 *_kernel_trace.csv (qkv / ff1 / N=1024 shapes)."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "p8s_kernel" not in n:
        continue
    tag = "p8s" + n.split("p8s_kernel")[1][:14]
    k = (tag, r.get("Grid_Size") or r.get("Grid_Size_X"))
    acc[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; acc[k][1] += 1
for k, (s, n) in sorted(acc.items()):
    print(k, "calls", n, "mean_us %.1f" % (s / n), "total_ms %.1f" % (s / 1e3))
