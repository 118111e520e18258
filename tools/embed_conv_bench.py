#!/usr/bin/env python3
"""Dev: to_embed (state columns) + ConvPositionEmbed at the bench shape (16 x 1000 rows, K = 80, C = 1024): the fused launch
(cvx_embed_conv31_f32) against the GEMM + depthwise-convolution pair it replaces, us per call."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
Bt, T, C, K = int(os.environ.get("BT", "16")), int(os.environ.get("T", "1000")), 1024, 80
g = torch.Generator().manual_seed(0)
x, base = torch.randn(Bt * T, K, generator=g).to(dev), torch.randn(Bt * T, C, generator=g).to(dev)
w = (torch.randn(C, 2288, generator=g) / math.sqrt(K)).to(dev)
dw, db = (torch.randn(C, 31, generator=g) / 5).to(dev), torch.randn(C, generator=g).to(dev)
y, h0, y2 = (torch.empty(Bt * T, C, device=dev) for _ in range(3))
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
def pair():
    ops.gemm(x, w[:, :K], h0, residual=base)
    ops.dwconv31_gelu_res(h0, dw, db, y2, Bt, T)
print(f"fused {timeit(lambda: ops.embed_conv31(x, w, base, dw, db, y, Bt, T)):.1f} us   gemm + dwconv {timeit(pair):.1f} us   "
      f"gemm alone {timeit(lambda: ops.gemm(x, w[:, :K], h0, residual=base)):.1f} us   rel diff {float((y - y2).norm() / y2.norm()):.2e}")
if "ectrace" in os.environ.get("CVX_LIB_PATH", ""):
    ops.embed_conv31(x, w, base, dw, db, y, Bt, T); torch.cuda.synchronize()
    st = y.view(Bt, T, C)[:, ::98, :].reshape(Bt, -1, 16, 64)[..., :12].contiguous().view(torch.int64).view(-1, 6).double()
    t0 = st[:, 0].min()
    d = (st[:, 1:] - st[:, :-1]) / 100
    print("blocks", st.shape[0], "start max %.1f us" % float((st[:, 0] - t0).max() / 100), "end max %.1f us" % float((st[:, 5] - t0).max() / 100))
    print("per block us: load+stage %.2f | barrier %.2f | mfma+rmw %.2f | barrier %.2f | conv %.2f" % tuple(d.mean(0).tolist()))
