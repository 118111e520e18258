#!/usr/bin/env python3
"""Mean duration of the vocoder kernels per (kernel, grid) from a rocprofv3 *_kernel_trace.csv; with CALLS=n also ms per generator call."""
import csv, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not any(t in n for t in ("conv_f16x3", "resblock_pair", "cl_split", "amax", "post_cl", "pow2", "conv1d_mfma", "cm_to_cl", "cl_to_cm")):
        n = "(other) " + n[:40]
    k = (n.split("(")[0][-48:] if not n.startswith("(other)") else n, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
    acc[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; acc[k][1] += 1
calls = int(os.environ.get("CALLS", "0"))
tot = 0.0
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    tot += s
    print("%-52s grid %-9s wg %-5s calls %5d mean_us %8.1f total_ms %8.2f" % (k[0], k[1], k[2], n, s / n, s / 1e3) + ("  per_call_ms %.3f" % (s / 1e3 / calls) if calls else ""))
print("sum of kernel time: %.2f ms" % (tot / 1e3) + ("  per call %.3f ms" % (tot / 1e3 / calls) if calls else ""))
