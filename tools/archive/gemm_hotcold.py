#!/usr/bin/env python3
"""Dev: what a COLD weight matrix costs a medium-kernel launch at config 2 (M = 1000): the same GEMM with one weight matrix
(hot in L2 / MALL after the first launch) vs cycling through 24 copies (> 256 MB: every launch streams its W from HBM, as in the model
where 400 MB of weights pass between two uses).  The difference bounds what an L2 prefetch of the next W could give."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "1000"))
def run(fns, iters=120):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (N, K, name) in [(3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "ff1"), (1024, 4096, "ff2")]:
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il)
    out = torch.empty(M, N, device=dev)
    o16 = ops.SplitIL(M, N, dev)
    ncopy = max(2, int(300e6 / (N * K * 4)))
    fns = []
    for c in range(ncopy):
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
        ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
        fns.append(lambda w=w, ws=ws, wil=wil: ops.gemm(a, w, out, a_split=il, w_split=ws, w_il=wil, out_split=o16, write_f32=(name in ("out", "ff2"))))
    hot = run(fns[:1]); cold = run(fns)
    print(f"{name:4s} N={N} K={K}: hot {hot:6.1f} us   cold ({ncopy} weight copies) {cold:6.1f} us   diff {cold - hot:5.1f}")
