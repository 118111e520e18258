#!/usr/bin/env python3
"""Dev: per-block cycle stamps of the medium-problem GEMM (gemm_f16x3_p8m.hip) on the shapes of one utterance.
Needs a -DCVX_DEV_FLAGS build:  python tools/devbuild.py dev -DCVX_DEV_FLAGS --files=gemm_f16x3.hip,gemm_f16x3_p8m.hip
                               CVX_LIB_PATH=tools/dev_dev.so python tools/gemm_small_trace.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "1000")); T = M // 2
ws_buf = ops._splitk_workspace(dev)
trace = ws_buf.view(torch.int64)[(100 << 20) // 8: (100 << 20) // 8 + 4096 * 16]
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
ang = torch.arange(T).float()[:, None] * inv[None, :]
rope = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
for (N, K, K1, name) in [(3072, 1024, 0, "qkv"), (1024, 1024, 0, "out"), (4096, 1024, 0, "ff1"), (1024, 4096, 0, "ff2"), (1024, 2048, 1024, "skip")]:
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev); b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    kw = dict(w_il=wil)
    if K1:
        a1, a2 = a[:, :K1].contiguous(), a[:, K1:].contiguous()
        i1, i2 = ops.SplitIL(M, K1, dev), ops.SplitIL(M, K - K1, dev); ops.split_act_f16(a1, i1); ops.split_act_f16(a2, i2)
        kw.update(a2=a2, a_split=i1, a2_split=i2, bias=b); a_in = a1
    else:
        il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il); kw.update(a_split=il); a_in = a
    c = torch.zeros(M, N, device=dev)
    if name == "qkv":
        qk = (torch.empty(M, 2048, dtype=torch.float16, device=dev), torch.empty(M, 2048, dtype=torch.float16, device=dev))
        Tp = ((T + 31) // 32) * 32
        vt = (torch.zeros(2 * 16 * 64, Tp, dtype=torch.float16, device=dev), torch.zeros(2 * 16 * 64, Tp, dtype=torch.float16, device=dev))
        kw.update(rope=rope, rope_cols=2048, out_split=qk, vt_split=vt, write_f32=False)
    elif name == "out": kw.update(residual=res)
    elif name == "ff1": kw.update(bias=b, act=1, out_split=ops.SplitIL(M, N, dev), write_f32=False)
    elif name == "ff2": kw.update(bias=b, residual=res, out_split=ops.SplitIL(M, N, dev))
    ops._GEMM_FLAGS = 4 << 8
    orig = ops.GemmSplitIO
    class IO(orig):                     # (the RoPE call gets no split-K scratch from ops.gemm: hand it over for the stamps)
        def __init__(self):
            super().__init__()
            self.workspace = ws_buf.data_ptr(); self.workspace_floats = (100 << 20) // 4
    ops.GemmSplitIO = IO
    for _ in range(3):
        trace.zero_()
        ops.gemm(a_in, w, c, w_split=ws, **kw)
    torch.cuda.synchronize()
    ops._GEMM_FLAGS = 0
    ops.GemmSplitIO = orig
    t = trace.view(-1, 2, 8).cpu().double()
    t = t[t[:, 0, 0] > 0]
    for gi in (0, 1):
        g_ = t[:, gi]
        pro, loop, xch, epi = g_[:, 1] - g_[:, 0], g_[:, 2] - g_[:, 1], g_[:, 3] - g_[:, 2], g_[:, 4] - g_[:, 3]
        wall = (g_[:, 6] - g_[:, 5]) / 100
        nk = (K // 32) // (4 if name in ("ff2", "skip", "out") else 1)
        print(f"{name:5s} group{gi} blocks {t.shape[0]:4d}: prologue {pro.mean():7.0f}  main loop {loop.mean():8.0f} ({loop.mean()/max(nk,1):6.0f} per K-tile, max {loop.max():8.0f})  exchange {xch.mean():6.0f}  "
              f"epilogue {epi.mean():7.0f}  total {(g_[:,4]-g_[:,0]).mean():8.0f} cyc = {wall.mean():5.1f} us (clk {(g_[:,4]-g_[:,0]).mean()/wall.mean()/1e3:.2f} GHz)")
    span = (t[:, :, 6].max() - t[:, :, 5].min()) / 100
    print(f"      kernel span (first block start -> last block end) {span:.1f} us; block starts spread {(t[:,0,5].max()-t[:,0,5].min())/100:.1f} us")
