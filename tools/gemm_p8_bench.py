#!/usr/bin/env python3
"""Dev: eight-phase ping-pong GEMM (csrc/gemm_f16x3_p8.hip) against the two-stage kernel on the transformer shapes of
BASELINE config 3 with interleaved A and W: bit-identity of the results (same MFMA sequence per accumulator) and time.
Env: M=16000, SHAPES=qkv,ff2, ZERO=1 (power probe), REPS=20."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=int(os.environ.get("REPS", "20")), warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = int(os.environ.get("M", "16000"))
only = os.environ.get("SHAPES")
tot = {0: 0.0, 1: 0.0}
for (N, K, K1, name, cnt, act) in [(3072, 1024, 0, "qkv", 8, 0), (1024, 1024, 0, "out", 8, 0), (4096, 1024, 0, "ff1", 8, 1),
                                   (1024, 4096, 0, "ff2", 8, 0), (1024, 2048, 1024, "skip", 4, 0)]:
    if only and name not in only.split(","):
        continue
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    if os.environ.get("ZERO") == "1":
        a.zero_(); w.zero_(); w[0, 0] = 1.0
    ws = ops.split_f16(w)
    wil = ops.split_f16_interleaved(ws)
    if K1:
        a1, a2 = a[:, :K1].contiguous(), a[:, K1:].contiguous()
        il1, il2 = ops.SplitIL(M, K1, dev), ops.SplitIL(M, K - K1, dev)
        ops.split_act_f16(a1, il1); ops.split_act_f16(a2, il2)
        kw = dict(a2=a2, a_split=il1, a2_split=il2)
        a_in = a1
    else:
        il = ops.SplitIL(M, K, dev)
        ops.split_act_f16(a, il)
        kw = dict(a_split=il)
        a_in = a
    outs, times = [], []
    for flags in (1, 0):                 # 1 = two-stage kernel, 0 = eight-phase
        ops._GEMM_FLAGS = flags
        c = torch.full((M, N), float("nan"), device=dev)
        fn = lambda: ops.gemm(a_in, w, c, bias=b, act=act, w_split=ws, w_il=wil, **kw)
        fn(); torch.cuda.synchronize()
        outs.append(c.clone())
        times.append(timeit(fn))
        tot[flags] += times[-1] * cnt
    same = torch.equal(outs[0], outs[1])
    maxd = float((outs[0] - outs[1]).abs().max())
    ref = None
    if N * K <= 4 * 1024 * 1024 and M <= 16000:
        hi, lo = (il1.dense() if K1 else il.dense())
        aa = hi.double() + lo.double()
        if K1:
            h2, l2 = il2.dense()
            aa = torch.cat([aa, h2.double() + l2.double()], 1)
        ref = aa[:2048] @ w.double().T + b.double()
        if act == 1: ref = torch.nn.functional.gelu(ref)
    err = float((outs[1][:2048].double() - ref).norm() / ref.norm()) if ref is not None else float("nan")
    fl = 2.0 * M * N * K
    print(f"{name:5s} N={N:5d} K={K:5d}: two-stage {times[0]:7.1f} us ({3*fl/times[0]/1e6:6.0f} TF exec)   p8 {times[1]:7.1f} us ({3*fl/times[1]/1e6:6.0f} TF exec)"
          f"   x{times[0]/times[1]:.3f}   bit-identical={same} maxdiff={maxd:.2e} err_vs_f64={err:.2e}", flush=True)
print(f"per-eval GEMM total: two-stage {tot[1]/1e3:.2f} ms, p8 {tot[0]/1e3:.2f} ms")
