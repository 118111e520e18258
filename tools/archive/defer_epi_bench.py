#!/usr/bin/env python3
"""Dev: what the producer forms of the deferred norm cost per launch at the bench shape (16,000 rows): to_out (K = 1024) and ff2
(K = 4096), N = 1024, interleaved operands, isolated launches back to back.
  fp32            residual fp32 -> fp32 store                        (round-3 to_out)
  fp32+twin       residual fp32 -> fp32 store + split twin           (round-3 ff2)
  pair            residual PAIR -> pair store                        (no fp32, no row sums)
  pair+rowsq      ... + row sums of squares                          (the pair-only residual stream)
  f32res->pair    residual fp32 -> pair store only
Env: M, REPS."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
M, N = int(os.environ.get("M", "16000")), 1024
def timeit(fn, iters=int(os.environ.get("REPS", "30")), warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for K, name in ((1024, "to_out"), (4096, "ff2")):
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il)
    c = torch.zeros(M, N, device=dev)
    tw = ops.SplitIL(M, N, dev)
    rp = ops.SplitIL(M, N, dev); ops.split_act_f16(res, rp)
    rowsq = torch.zeros(M, N // 64, device=dev)
    kw = dict(w_split=ws, w_il=wil, a_split=il, bias=b)
    forms = {
        "fp32": lambda: ops.gemm(a, w, c, residual=res, **kw),
        "fp32+twin": lambda: ops.gemm(a, w, c, residual=res, out_split=tw, **kw),
        "pair": lambda: ops.gemm(a, w, c, res_split=rp, out_split=tw, write_f32=False, **kw),
        "pair+rowsq": lambda: ops.gemm(a, w, c, res_split=rp, out_split=tw, c_rowsq=rowsq, write_f32=False, **kw),
        "f32res->pair": lambda: ops.gemm(a, w, c, residual=res, out_split=tw, c_rowsq=rowsq, write_f32=False, **kw),
    }
    for rnd in range(2):
        print(name, " ".join(f"{k} {timeit(f):.1f}" for k, f in forms.items()), flush=True)
