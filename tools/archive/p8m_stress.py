#!/usr/bin/env python3
"""Dev: race screen for the medium-problem GEMM, the fused reduction + norm and the key-split attention: every case is run ITERS times
and compared BITWISE with its first result (the kernels are deterministic), while other processes load the GPU (run two or three
copies side by side: `for i in 1 2 3; do python tools/p8m_stress.py & done; wait`).  A DMA-ring hazard shows up as a rare mismatch."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "300"))
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "0")))
bad = 0
def case(name, fn, outs):
    global bad
    fn(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs()]
    n_bad = 0
    for it in range(ITERS):
        fn()
        if it % 10 == 9 or it == ITERS - 1:
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(ref, outs())):
                n_bad += 1
    print(f"{name:40s} mismatching checks: {n_bad} of {ITERS // 10}", flush=True)
    bad += n_bad
for (M, N, K, K1) in [(1000, 3072, 1024, 0), (1000, 1024, 4096, 0), (1000, 1024, 2048, 1024), (333, 4096, 1024, 0), (4000, 1024, 1024, 0), (1000, 80, 1024, 0)]:
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev); r = torch.randn(M, N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    kw = dict(w_split=ws, w_il=wil)
    if K1:
        a1, a2 = a[:, :K1].contiguous(), a[:, K1:].contiguous()
        i1, i2 = ops.SplitIL(M, K1, dev), ops.SplitIL(M, K - K1, dev); ops.split_act_f16(a1, i1); ops.split_act_f16(a2, i2)
        kw.update(a2=a2, a_split=i1, a2_split=i2); a_in = a1
    else:
        il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il); kw.update(a_split=il); a_in = a
    c = torch.empty(M, N, device=dev)
    if N % 32 == 0:
        tw = ops.SplitIL(M, N, dev)
        case(f"gemm M={M} N={N} K={K} res+twin", lambda: ops.gemm(a_in, w, c, bias=b, residual=r, out_split=tw, **kw), lambda: (c, tw.buf))
        if N <= 1024:
            y = ops.SplitIL(M, N, dev)
            case(f"gemm+norm M={M} N={N} K={K}", lambda: ops.gemm(a_in, w, c, bias=b, residual=r, norm=dict(gamma=b, beta=b, out_split=y, scale=None), **kw), lambda: (c, y.buf))
        o = ops.SplitIL(M, N, dev)
        case(f"gemm M={M} N={N} K={K} gelu split", lambda: ops.gemm(a_in, w, c, bias=b, act=1, out_split=o, write_f32=False, **kw), lambda: (o.buf,))
    else:
        case(f"gemm M={M} N={N} K={K} plain", lambda: ops.gemm(a_in, w, c, **kw), lambda: (c,))
for (Bt, T) in [(2, 500), (2, 333), (4, 250)]:
    H = 16; Tp = (T + 31) // 32 * 32
    q = torch.randn(Bt * T, 2 * H * 64, generator=g).to(dev); v = torch.zeros(Bt * H * 64, Tp, device=dev); v[:, :T] = torch.randn(Bt * H * 64, T, generator=g).to(dev)
    qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
    oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
    case(f"attention Bt={Bt} T={T}", lambda: ops.attention_f16x3((qh, ql), (vh, vl), None, Bt, T, H, 0.125, out_split=(oh, ol)), lambda: (oh, ol))
print("TOTAL mismatching checks:", bad)
sys.exit(1 if bad else 0)
