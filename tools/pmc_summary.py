#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/pmc_summary.json.

Units and corrections follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is used as reported
(uncalibrated).  Values are averaged per launch per kernel.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import json
import re
import sys
from collections import defaultdict


EPI = {0: "generic", 1: "qkv", 2: "res", 3: "gelu_split", 4: "bias", 5: "res_tw", 6: "bias_tw", 7: "gelu_rs", 8: "qkv_rs"}    # gemm_common.h


def short(name: str) -> str:
    m = re.search(r"_GLOBAL__N_1(\d\d)(\w+)", name)          # mangled: _ZN12_GLOBAL__N_1<len><name>...
    if m:
        return m.group(2)[: int(m.group(1))]
    m = re.search(r"(\w+_kernel)(<\d+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def instance(name: str) -> str:
    """The template instance of the GEMM kernels, e.g. gemm_f16x3_p8s_kernel<a2=0,epi=res_tw>: one row per epilogue, so that the
    traffic ratio of the dominant kernel can be attributed to a shape (round-4 review: all eight instances were one row)."""
    k = short(name)
    if not k.startswith("gemm_f16x3_p8"):
        return k
    m = re.search(re.escape(k) + r"ILb([01])ELi(\d+)E", name) or re.search(re.escape(k) + r"<(true|false|[01]), *(\d+)", name)
    if not m:
        return k
    a2 = {"true": 1, "false": 0}.get(m.group(1), m.group(1))
    return f"{k}<a2={a2},epi={EPI.get(int(m.group(2)), m.group(2))}>"


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for k in {short(row["Kernel_Name"]), instance(row["Kernel_Name"])}:      # the kernel as a whole and its template instance
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if "kernel" not in k or k.startswith("void at") or "at::native" in k:
            continue
        fk = f.get(k, [0.0, 0]); wk = w.get(k, [0.0, 0])
        n = max(fk[1], wk[1], 1)
        fetch_b = 2.0 * fk[0] * 1024 / max(fk[1], 1)
        write_b = wk[0] * 1024 / max(wk[1], 1)
        res[k] = {"launches": n, "fetch_bytes_per_launch_x2_corrected": round(fetch_b),
                  "write_bytes_per_launch": round(write_b), "hbm_bytes_per_launch": round(fetch_b + write_b)}
    # aggregate over the template instances of the dominant kernel
    g = [v for k, v in res.items() if k.startswith("gemm_f32_kernel")]
    if g:
        tot_l = sum(v["launches"] for v in g)
        res["gemm_f32_kernel"] = {
            "launches": tot_l,
            "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in g) / tot_l)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
