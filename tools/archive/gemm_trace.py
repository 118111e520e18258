#!/usr/bin/env python3
"""Dev: per-block cycle stamps of the eight-phase GEMM (dbg bit 2): prologue, main loop, epilogue issue, store drain."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
import importlib
gb = importlib.import_module("tools.gemm_epi_probe") if False else None
dev = torch.device("cuda:0")
M, T = 16000, 1000
def setup(N, K):
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev); b = torch.randn(N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, K, dev); ops.split_act_f16(a, il)
    return a, w, b, ws, wil, il, torch.zeros(M, N, device=dev)
trace = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
real = ops._splitk_workspace
def run(tag, N, K, **kwf):
    a, w, b, ws, wil, il, c = setup(N, K)
    kw = kwf(b, N) if callable(kwf) else {}
    from covomix_amd._lib import GemmSplitIO
    ops._GEMM_FLAGS = 4 << 8
    # smuggle the trace buffer through io.workspace: patch gemm to set it
    orig = ops.GemmSplitIO
    class IO(orig):
        def __init__(self):
            super().__init__()
            self.workspace = trace.data_ptr(); self.workspace_floats = trace.numel() * 2
    ops.GemmSplitIO = IO
    for _ in range(3):
        trace.zero_()
        ops.gemm(a, w, c, w_split=ws, w_il=wil, a_split=il, **kw)
    torch.cuda.synchronize()
    ops.GemmSplitIO = orig
    t = trace.view(-1, 8, 8).cpu().double()
    t = t[t[:, 0, 0] > 0]
    nb = t.shape[0]
    g0, g1 = t[:, 0], t[:, 4]                      # wave 0 (group 0) and wave 4 (group 1)
    for nm, g in (("group0", g0), ("group1", g1)):
        pro, loop, epi, drain = (g[:, 1] - g[:, 0]), (g[:, 2] - g[:, 1]), (g[:, 3] - g[:, 2]), (g[:, 4] - g[:, 3])
        print(f"{tag:28s} {nm} blocks {nb:5d}: prologue {pro.mean():8.0f}  main loop {loop.mean():9.0f}  epilogue issue {epi.mean():8.0f}  store drain {drain.mean():8.0f}  total {(g[:,4]-g[:,0]).mean():9.0f} cycles"
              f"   (max total {(g[:,4]-g[:,0]).max():9.0f})")
    # wall-clock view (100 MHz constant counter): per CU, gaps between consecutive blocks
    hw = t[:, 0, 7].long()
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; xcc = (hw >> 16) & 0xf       # HW_ID: cu_id [11:8], sh [12], se [15:13]; XCC_ID is a separate reg on gfx950
    key = ((hw >> 8) & 0xff) | (((hw >> 24) & 0xf) << 8)          # (se, sh, cu) | xcc
    start, end = t[:, 0, 6], torch.maximum(t[:, 0, 5], t[:, 4, 5])
    k0 = float(start.min())
    print(f"    wall: kernel span {float(end.max() - start.min()) / 100:.1f} us; block duration mean {float((end - start).mean()) / 100:.1f} us; "
          f"distinct hw keys {len(set(key.tolist()))}")
    import collections
    by = collections.defaultdict(list)
    for i in range(nb):
        by[int(key[i]) * 16 + 0].append((float(start[i]), float(end[i])))
    gaps = []
    for k, v in by.items():
        v.sort()
        for (s0, e0), (s1, e1) in zip(v[:-1], v[1:]):
            gaps.append((s1 - e0) / 100)
    if gaps:
        gt = torch.tensor(gaps)
        print(f"    wall: {len(gaps)} same-key successions, gap mean {float(gt.mean()):.2f} us  median {float(gt.median()):.2f}  max {float(gt.max()):.2f}")
    ss = torch.sort(start)[0]; ee = torch.sort(end)[0]
    print(f"    wall: start of 1st/128th/256th block: 0 / {float(ss[min(127, nb-1)] - ss[0]) / 100:.2f} / {float(ss[min(255, nb - 1)] - ss[0]) / 100:.2f} us;  ends of the last 256 blocks span {float(ee[-1] - ee[max(0, nb - 256)]) / 100:.2f} us")
    # per round: sort by start time
    st = g0[:, 5] if False else g0[:, 0]
    order = torch.argsort(st)
    n_round = 256
    for r in range(0, nb, n_round):
        idx = order[r:r + n_round]
        print(f"    round {r // n_round}: start spread {float(st[idx].max() - st[idx].min()):9.0f}  mean prologue {float((g0[idx,1]-g0[idx,0]).mean()):7.0f} loop {float((g0[idx,2]-g0[idx,1]).mean()):8.0f} epi {float((g0[idx,3]-g0[idx,2]).mean()):7.0f} drain {float((g0[idx,4]-g0[idx,3]).mean()):7.0f}")
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
ang = torch.arange(T).float()[:, None] * inv[None, :]
rope = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
res = torch.randn(M, 1024, device=dev)
run("out (residual + fp32)", 1024, 1024, kwf=lambda b, N: dict(residual=res))
run("ff1 (bias+gelu+split IL)", 4096, 1024, kwf=lambda b, N: dict(bias=b, act=1, out_split=ops.SplitIL(M, N, dev), write_f32=False))
qk = (torch.empty(M, 2048, dtype=torch.float16, device=dev), torch.empty(M, 2048, dtype=torch.float16, device=dev))
vt = (torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev), torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev))
run("qkv (rope+split+vt)", 3072, 1024, kwf=lambda b, N: dict(rope=rope, rope_cols=2048, out_split=qk, vt_split=vt, write_f32=False))
run("ff2 (bias+res+fp32+twin)", 1024, 4096, kwf=lambda b, N: dict(bias=b, residual=res, out_split=ops.SplitIL(M, N, dev)))
