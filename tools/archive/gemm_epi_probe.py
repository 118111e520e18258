#!/usr/bin/env python3
"""Dev: what the epilogue of the large-problem GEMM costs, by switching its pieces off from the caller (M=16000)."""
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
M, T = 16000, 1000
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
ang = torch.arange(T).float()[:, None] * inv[None, :]
rope = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
def setup(N, K):
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il)
    return a, w, b, ws, wil, il, torch.zeros(M, N, device=dev)
def run(tag, a, w, c, ws, wil, il, **kw):
    ops._GEMM_FLAGS = 0
    t = timeit(lambda: ops.gemm(a, w, c, w_split=ws, w_il=wil, a_split=il, **kw))
    ops._GEMM_FLAGS = 256
    t0 = timeit(lambda: ops.gemm(a, w, c, w_split=ws, w_il=wil, a_split=il, **kw))
    print(f"{tag:52s} {t:7.1f} us   (main loop {t0:6.1f}, epilogue {t - t0:6.1f})", flush=True)
# ff1
a, w, b, ws, wil, il, c = setup(4096, 1024)
o_il = ops.SplitIL(M, 4096, dev)
o_sep = (torch.empty(M, 4096, dtype=torch.float16, device=dev), torch.empty(M, 4096, dtype=torch.float16, device=dev))
run("ff1 fp32 store only", a, w, c, ws, wil, il)
run("ff1 bias + fp32 store", a, w, c, ws, wil, il, bias=b)
run("ff1 bias + GELU + fp32 store", a, w, c, ws, wil, il, bias=b, act=1)
run("ff1 bias + split IL store (no fp32)", a, w, c, ws, wil, il, bias=b, out_split=o_il, write_f32=False)
run("ff1 bias + split separate store (no fp32)", a, w, c, ws, wil, il, bias=b, out_split=o_sep, write_f32=False)
run("ff1 bias + GELU + split IL (model)", a, w, c, ws, wil, il, bias=b, act=1, out_split=o_il, write_f32=False)
run("ff1 bias + SiLU + split IL", a, w, c, ws, wil, il, bias=b, act=2, out_split=o_il, write_f32=False)
# qkv
a, w, b, ws, wil, il, c = setup(3072, 1024)
qk = (torch.empty(M, 2048, dtype=torch.float16, device=dev), torch.empty(M, 2048, dtype=torch.float16, device=dev))
qk3 = (torch.empty(M, 3072, dtype=torch.float16, device=dev), torch.empty(M, 3072, dtype=torch.float16, device=dev))
vt = (torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev), torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev))
run("qkv fp32 store only", a, w, c, ws, wil, il)
run("qkv rope + fp32 store", a, w, c, ws, wil, il, rope=rope, rope_cols=2048)
run("qkv split separate q|k|v row-major (no rope, no vt)", a, w, c, ws, wil, il, out_split=qk3, write_f32=False)
run("qkv rope + split row-major q|k|v (no vt)", a, w, c, ws, wil, il, rope=rope, rope_cols=2048, out_split=qk3, write_f32=False)
run("qkv rope + split q|k + transposed v (model)", a, w, c, ws, wil, il, rope=rope, rope_cols=2048, out_split=qk, vt_split=vt, write_f32=False)
# out / ff2
a, w, b, ws, wil, il, c = setup(1024, 1024)
res = torch.randn(M, 1024, device=dev)
run("out fp32 store only", a, w, c, ws, wil, il)
run("out residual + fp32 store (model)", a, w, c, ws, wil, il, residual=res)
a, w, b, ws, wil, il, c = setup(1024, 4096)
tw = ops.SplitIL(M, 1024, dev)
run("ff2 bias + residual + fp32", a, w, c, ws, wil, il, bias=b, residual=res)
run("ff2 bias + residual + fp32 + split twin (model)", a, w, c, ws, wil, il, bias=b, residual=res, out_split=tw)
