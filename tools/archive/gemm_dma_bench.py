#!/usr/bin/env python3
"""Dev: the pre-split all-DMA GEMM (256x256 tile) on the transformer shapes of BASELINE config 3, both the
three-term (f16x3) and the single-term (f16) kernels.  Env knobs are read by the library (CVX_GEMM_LOADERS, ...)."""
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
M = int(os.environ.get("M", "16000"))
tot = {3: 0.0, 1: 0.0}
only = os.environ.get("SHAPES")            # e.g. SHAPES=ff2 TERMS=3 for a PMC run on one kernel
for (N, K, name, cnt) in [(3072, 1024, "qkv", 8), (1024, 1024, "out", 8), (4096, 1024, "ff1", 8), (1024, 4096, "ff2", 8), (1024, 2048, "skip", 4)]:
    a, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / math.sqrt(K)
    if os.environ.get("ZERO") == "1":          # power probe: all-zero operands toggle no datapath bits
        a.zero_(); w.zero_(); w[0, 0] = 1.0
    c = torch.empty(M, N, device=dev)
    ref = None
    if only and name not in only.split(","):
        continue
    for terms in ([int(os.environ["TERMS"])] if os.environ.get("TERMS") else (3, 1)):
        ws = ops.split_f16(w, with_lo=(terms == 3))
        ah, al = ops.split_act_f16(a)
        pad = int(os.environ.get("PAD", "0"))       # row padding in halves (L2 / HBM channel spreading)
        if pad:
            def padded(t):
                if t is None:
                    return None
                buf = torch.empty(t.shape[0], t.shape[1] + pad, dtype=t.dtype, device=t.device)
                buf[:, : t.shape[1]].copy_(t)
                return buf[:, : t.shape[1]]
            ws = (padded(ws[0]), padded(ws[1]), ws[2])
            ah, al = padded(ah), padded(al)
        asp = (ah, al if terms == 3 else None)
        il = ops.split_f16_interleaved(ws) if (terms == 3 and os.environ.get("WIL") == "1" and not pad) else None
        t = timeit(lambda: ops.gemm(a, w, c, w_split=ws, a_split=asp, w_il=il))
        if ref is None:
            ref = (ah.double() + al.double()) @ w.double().T if N * K <= 1024 * 1024 else None
        err = float((c.double() - ref).norm() / ref.norm()) if ref is not None else float("nan")
        tot[terms] += t * cnt
        print(f"terms={terms} {name:5s} N={N:5d} K={K:5d}: {t:8.1f} us  {2*M*N*K/t/1e6:7.1f} TF algorithmic  {2*M*N*K*terms/t/1e6:7.1f} TF executed  err {err:.2e}")
print(f"per-eval GEMM total: f16x3 {tot[3]/1e3:.2f} ms, f16 {tot[1]/1e3:.2f} ms   (x32 evals: {tot[3]*32/1e3:.0f} / {tot[1]*32/1e3:.0f} ms)")
