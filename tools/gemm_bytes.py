#!/usr/bin/env python3
"""Algorithmic bytes and FLOPs of the 36 transformer GEMMs of one evaluation at BASELINE config 3 (2B x T = 16,000 rows),
per shape and launch-weighted - the figures DESIGN.md section 7 compares the PMC traffic with.

Operands are split pairs (fp16 hi + fp16 lo = 4 bytes per element, the same as fp32).  "algorithmic" = every operand and
every output touched once:  A [M, K] + W [N, K] (+ residual [M, N] fp32) read;  C written as fp32 and / or as a split pair.
    python tools/gemm_bytes.py [M]"""
import sys
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
E = 4                     # bytes per element of a split pair or an fp32 value
shapes = [  # name, N, K, launches per evaluation, residual read, fp32 written, split written
    ("to_qkv  (RoPE, split q|k, split v^T)", 3072, 1024, 8, 0, 0, 1),
    ("to_out  (+ residual)",                 1024, 1024, 8, 1, 1, 0),
    ("ff1     (bias, GELU, split)",          4096, 1024, 8, 0, 0, 1),
    ("ff2     (bias, + residual, + twin)",   1024, 4096, 8, 1, 1, 7 / 8),      # no twin after the last layer
    ("skip    (K-split A | A2, bias)",       1024, 2048, 4, 0, 1, 0),
]
tot_b = tot_f = tot_n = 0
print(f"M = {M} rows")
print(f"{'GEMM':40s} {'N':>5s} {'K':>5s} {'n':>2s} {'read MB':>8s} {'write MB':>9s} {'total MB':>9s} {'GFLOP':>8s} {'FLOP/B':>7s}")
for name, N, K, n, res, f32, spl in shapes:
    rd = M * K * E + N * K * E + res * M * N * 4
    wr = f32 * M * N * 4 + spl * M * N * E
    fl = 2.0 * M * N * K
    print(f"{name:40s} {N:5d} {K:5d} {n:2d} {rd / 1e6:8.1f} {wr / 1e6:9.1f} {(rd + wr) / 1e6:9.1f} {fl / 1e9:8.1f} {fl / (rd + wr):7.0f}")
    tot_b += n * (rd + wr); tot_f += n * fl; tot_n += n
print(f"launch-weighted mean over {tot_n} GEMMs: {tot_b / tot_n / 1e6:.1f} MB and {tot_f / tot_n / 1e9:.1f} algorithmic GFLOP per launch "
      f"({3 * tot_f / tot_n / 1e9:.1f} executed); per evaluation {tot_b / 1e9:.2f} GB, {tot_f / 1e12:.2f} TFLOP")
