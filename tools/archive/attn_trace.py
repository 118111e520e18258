#!/usr/bin/env python3
"""Dev: per-segment cycle counts of the split-precision attention loop (library built with -DCVX_ATT_TRACE, selected through
CVX_LIB_PATH): wait + barrier, K reads + S MFMAs (to completion), softmax VALU, V reads + PV MFMAs (to completion)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
Bt, T, H = 16, 1000, 16
Tp = (T + 31) // 32 * 32
q = torch.randn(Bt * T, 2 * H * 64, device=dev); v = torch.randn(Bt * H * 64, Tp, device=dev)
qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
trace = torch.zeros(Bt * T, H * 64, device=dev)            # doubles as the (unused) fp32 output
for _ in range(3):
    trace.zero_()
    ops.attention_f16x3((qh, ql), (vh, vl), trace, Bt, T, H, 0.125, out_split=(oh, ol))
torch.cuda.synchronize()
t = trace.view(torch.int64).reshape(-1)[: 2048 * 4 * 8].view(2048, 4, 8).double().cpu()
nt = t[:, :, 4].mean()
for w in range(4):
    seg = t[:, w, :4].mean(dim=0) / nt
    print(f"wave {w}: per tile: wait+barrier {seg[0]:7.0f}  K reads + S MFMAs {seg[1]:7.0f}  softmax {seg[2]:7.0f}  V reads + PV MFMAs {seg[3]:7.0f}  total {seg.sum():7.0f} cycles")
# effective shader clock: cycles a CU spends on its share of the blocks / wall time of the launch (blocks per CU slot known)
import time
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.attention_f16x3((qh, ql), (vh, vl), trace, Bt, T, H, 0.125, out_split=(oh, ol))
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 10 * 1e3
per_block = float(t[:, :, :4].sum(dim=2).mean())
conc = int(os.environ.get("CONCURRENT", "2"))
print(f"launch {us:.1f} us; cycles per block {per_block:.0f}; {2048 / 256 / conc:.0f} rounds of {conc} concurrent blocks per CU -> "
      f"effective clock {per_block * 2048 / 256 / conc / us / 1e3:.2f} GHz (lower bound: launch tails not counted)")
