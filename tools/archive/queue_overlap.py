#!/usr/bin/env python3
"""From a rocprofv3 *_kernel_trace.csv of tools/bench_pipeline.py: are the text2semantic decode kernels and the acoustic-solve
kernels on the GPU AT THE SAME TIME?  Kernels are classed by name (decode: gemv / attn / sample kernels of t2s_decode.hip; solve:
the split-precision GEMM / attention / HiFi-GAN kernels); the trace is cut into the schedules of the tool by the long idle gaps
between them, and for each the busy time of either class (union of its kernel intervals) and of BOTH at once is printed - plus
the queues (Queue_Id) each class ran on.   usage: queue_overlap.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
DECODE = ("gemv_kernel", "attn_kernel", "sample_kernel")
SOLVE = ("gemm_f16x3", "attention_f16x3", "conv_f16x3", "resblock_pair", "splitk_reduce", "rownorm_scale", "adarmsnorm", "cfg_axpy", "dwconv31")
ev = []
for r in csv.DictReader(open(sys.argv[1], newline="")):
    n = r["Kernel_Name"]
    cls = "decode" if any(k in n for k in DECODE) else "solve" if any(k in n for k in SOLVE) else None
    if cls:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls, r.get("Queue_Id", "?")))
ev.sort()
if not ev:
    sys.exit("no decode / solve kernels in the trace")
# segments: a gap of > 150 ms without any kernel separates the schedules (model upload, warm-up, printing)
segs, cur = [], [ev[0]]
last_end = ev[0][1]
for e in ev[1:]:
    if e[0] - last_end > 150e6:
        segs.append(cur); cur = []
    cur.append(e); last_end = max(last_end, e[1])
segs.append(cur)


def union(iv):
    tot, end = 0, None
    out = []
    for s, e in sorted(iv):
        if end is None or s > end:
            out.append([s, e]); end = e
        elif e > end:
            out[-1][1] = e; end = e
    return out


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


print(f"{'segment':>7} {'wall ms':>9} {'decode busy':>12} {'solve busy':>11} {'both busy':>10} {'decode hidden':>14}  queues (decode | solve)")
for k, seg in enumerate(segs):
    if len(seg) < 2000:
        continue
    d = union([(s, e) for s, e, c, _ in seg if c == "decode"])
    v = union([(s, e) for s, e, c, _ in seg if c == "solve"])
    if not d or not v:
        continue
    wall = (max(e for _, e, _, _ in seg) - min(s for s, _, _, _ in seg)) / 1e6
    db, vb, both = sum(e - s for s, e in d) / 1e6, sum(e - s for s, e in v) / 1e6, inter(d, v) / 1e6
    qd = sorted({q for _, _, c, q in seg if c == "decode"}); qv = sorted({q for _, _, c, q in seg if c == "solve"})
    print(f"{k:7d} {wall:9.1f} {db:12.1f} {vb:11.1f} {both:10.1f} {100 * both / db:13.1f}%  {','.join(qd)} | {','.join(qv)}")
