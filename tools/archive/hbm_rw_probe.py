#!/usr/bin/env python3
"""Dev: what HBM sustains for pure writes, pure reads and copies at the sizes the GEMM epilogues move (torch kernels)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for mb in (65, 262, 1048):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tf = timeit(lambda: a.fill_(1.0))
    tz = timeit(lambda: a.zero_())
    tc = timeit(lambda: b.copy_(a))
    tr = timeit(lambda: a.sum())
    print(f"{mb:5d} MB: fill {tf:7.1f} us = {mb/tf*1e-3*1e3:6.2f} TB/s   zero {tz:7.1f} us = {mb/tz:6.2f} TB/s   copy {tc:7.1f} us = {2*mb/tc:6.2f} TB/s (r+w)   sum {tr:7.1f} us = {mb/tr:6.2f} TB/s")
