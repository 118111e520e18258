#!/usr/bin/env python3
"""Dev: the large-problem GEMM on the transformer shapes of BASELINE config 3 WITH THE MODEL'S EPILOGUES (qkv: RoPE +
split q|k + transposed split v; out: residual; ff1: bias + GELU + split, no fp32 store; ff2: bias + residual + split twin;
skip: K-split A|A2 + bias), interleaved A and W.  Columns = cvx_gemm_split_io.flags values (VARIANTS=, default "144,80,16,8"):
16 + 128 large-problem kernel with 256-row tiles, 16 + 64 with 192-row tiles, 16 its own choice of height, 8 medium-problem kernel,
0 the library's choice of kernel (+ 4 one tile per block; dev builds: + 256 main loop only).
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
T = int(os.environ.get("T", "1000"))
if M % T:
    T = M                                   # (one sequence: the RoPE table covers every row)
only = os.environ.get("SHAPES")
variants = [int(v) for v in os.environ.get("VARIANTS", "144,80,16,8").split(",")]
tot = {v: 0.0 for v in variants}
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
ang = torch.arange(T).float()[:, None] * inv[None, :]
rope = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
for (N, K, K1, name, cnt) in [(3072, 1024, 0, "qkv", 8), (1024, 1024, 0, "out", 8), (4096, 1024, 0, "ff1", 8),
                              (1024, 4096, 0, "ff2", 8), (1024, 2048, 1024, "skip", 4)]:
    if only and name not in only.split(","):
        continue
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if name in ("out", "ff2") else None
    if os.environ.get("ZERO") == "1":
        a.zero_(); w.zero_(); w[0, 0] = 1.0
    ws = ops.split_f16(w)
    wil = ops.split_f16_interleaved(ws)
    if K1:
        a1, a2 = a[:, :K1].contiguous(), a[:, K1:].contiguous()
        il1, il2 = ops.SplitIL(M, K1, dev), ops.SplitIL(M, K - K1, dev)
        ops.split_act_f16(a1, il1); ops.split_act_f16(a2, il2)
        kw = dict(a2=a2, a_split=il1, a2_split=il2, bias=b)
        a_in = a1
    else:
        il = ops.SplitIL(M, K, dev)
        ops.split_act_f16(a, il)
        kw = dict(a_split=il)
        a_in = a
    c = torch.zeros(M, N, device=dev)
    if name == "qkv":
        qk = (torch.empty(M, 2048, dtype=torch.float16, device=dev), torch.empty(M, 2048, dtype=torch.float16, device=dev))
        Tp = (T + 31) // 32 * 32
        vt = (torch.zeros((M // T) * 16 * 64, Tp, dtype=torch.float16, device=dev), torch.zeros((M // T) * 16 * 64, Tp, dtype=torch.float16, device=dev))
        kw.update(rope=rope, rope_cols=2048, out_split=qk, vt_split=vt, write_f32=False)
    elif name == "out":
        kw.update(residual=res)
    elif name == "ff1":
        kw.update(bias=b, act=1, out_split=ops.SplitIL(M, N, dev), write_f32=False)
    elif name == "ff2":
        kw.update(bias=b, residual=res, out_split=ops.SplitIL(M, N, dev))
    times = {v: [] for v in variants}
    fn = lambda: ops.gemm(a_in, w, c, w_split=ws, w_il=wil, **kw)
    for rep in range(int(os.environ.get("ROUNDS", "3")) + 1):          # interleaved A/B/C rounds; the first one is a warm-up
        for flags in variants:
            ops._GEMM_FLAGS = flags
            t = timeit(fn, iters=10)
            if rep:
                times[flags].append(t)
    times = {v: sorted(ts)[len(ts) // 2] for v, ts in times.items()}
    for flags in variants:
        tot[flags] += times[flags] * cnt
    fl = 2.0 * M * N * K
    print(f"{name:5s} N={N:5d} K={K:5d}: " + "   ".join(f"[{v}] {times[v]:7.1f} us ({3*fl/times[v]/1e6:5.0f} TF)" for v in variants), flush=True)
print("per-eval GEMM total (ms): " + "  ".join(f"[{v}] {tot[v]/1e3:.2f}" for v in variants))
