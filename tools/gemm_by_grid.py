#!/usr/bin/env python3
"""Mean duration of the large-problem GEMM per grid size from a rocprofv3 *_kernel_trace.csv (qkv / ff1 / N=1024 shapes)."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "dma256" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"].split("dma256_kernel")[1][:24], r.get("Grid_Size") or r.get("Grid_Size_X"))
    acc[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; acc[k][1] += 1
for k, (s, n) in sorted(acc.items()):
    print(k, "calls", n, "mean_us %.1f" % (s / n), "total_ms %.1f" % (s / 1e3))
