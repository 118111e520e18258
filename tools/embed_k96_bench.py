#!/usr/bin/env python3
"""Dev: to_embed's state columns (K = 80) as a K = 96 split-precision product on the large / medium kernel vs the exact-fp32
generic kernel it runs on today.  python tools/embed_k96_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa
import torch
import covomix_amd.ops as ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (16000, 8000, 4000, 1000, 500):
    x = torch.randn(M, 80, generator=g).to(dev)
    w = (torch.randn(1024, 80, generator=g) / 9).to(dev)
    base = torch.randn(M, 1024, generator=g).to(dev)
    out0, out1 = torch.empty(M, 1024, device=dev), torch.empty(M, 1024, device=dev)
    t0 = timeit(lambda: ops.gemm(x, w, out0, residual=base))
    xp = torch.zeros(M, 96, device=dev); xp[:, :80] = x
    wp = torch.zeros(1024, 96, device=dev); wp[:, :80] = w
    ws = ops.split_f16(wp); wil = ops.split_f16_interleaved(ws)
    one = torch.tensor([64.0], device=dev)
    il = ops.SplitIL(M, 96, dev); ops.split_act_f16(xp, il, scale=one)
    f = lambda: ops.gemm(xp, wp, out1, residual=base, a_split=il, w_split=ws, w_il=wil, a_scale=one)
    t1 = timeit(f)
    ref = (x.double() @ w.double().T + base.double())
    e0 = float((out0.double() - ref).norm() / ref.norm()); e1 = float((out1.double() - ref).norm() / ref.norm())
    print(f"M={M:6d}: generic fp32 {t0:7.1f} us (err {e0:.1e})   K=96 split {t1:7.1f} us (err {e1:.1e})")
