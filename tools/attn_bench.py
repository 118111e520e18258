#!/usr/bin/env python3
"""Dev: split-precision attention at BASELINE config-3 size (Bt=16, T=1000, H=16), both the 3-term and 1-term kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
Bt, T, H = int(os.environ.get("BT", "16")), int(os.environ.get("T", "1000")), 16
Tp = (T + 31) // 32 * 32
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
q = torch.randn(Bt * T, 2 * H * 64, device=dev)
v = torch.randn(Bt * H * 64, Tp, device=dev)
if os.environ.get('ZERO') == '1':
    q.zero_(); v.zero_()          # power probe: same instruction stream, no operand toggling
qh, ql = ops.split_act_f16(q)
vh, vl = ops.split_act_f16(v)
oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
flops = 4.0 * Bt * H * T * T * 64
for terms in (3, 1):
    qk = (qh, ql if terms == 3 else None); vt = (vh, vl if terms == 3 else None)
    t = timeit(lambda: ops.attention_f16x3(qk, vt, None, Bt, T, H, 0.125, out_split=(oh, ol if terms == 3 else None)))
    print(f"attention terms={terms}: {t:7.1f} us  {flops/t/1e6:7.1f} TF algorithmic  {flops*terms/t/1e6:7.1f} TF executed")
