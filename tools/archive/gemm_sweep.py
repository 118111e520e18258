#!/usr/bin/env python3
"""Dev: GEMM time vs K / M / N to separate per-launch fixed cost from per-K cost."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N) in [(16000, 1024), (16000, 4096), (4096, 1024), (32768, 1024), (16384, 1024)]:
    for K in (128, 256, 512, 1024, 2048, 4096):
        a, w, c = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm(a, w, c))
        print(f"M={M:6d} N={N:5d} K={K:5d}: {t:9.1f} us  {2*M*N*K/t/1e6:7.2f} TF  ideal {2*M*N*K/157.3e6:8.1f} us")
