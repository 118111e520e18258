#!/usr/bin/env python3
"""Dev: AdaRMSNorm in the model's configuration (x fp32 [16000, 1024] -> interleaved split pair, device pre-scale) for a
library selected with CVX_LIB_PATH (A/B of load / store cache policies: -DCVX_NORM_NT=0..3).  Prints us per launch and TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
M, D = 16000, 1024
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, D, generator=g).to(dev) for _ in range(4)]            # rotate buffers: nothing stays in the 256 MB MALL by accident
outs = [ops.SplitIL(M, D, dev) for _ in range(4)]
gam, bet = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
sc = torch.tensor([16.0], device=dev)
def run(n):
    for i in range(n):
        ops.adarmsnorm(xs[i % 4], gam, bet, None, out_split=outs[i % 4], split_scale=sc)
run(8); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    s.record(); run(40); e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 40 * 1e3)
print(f"{os.environ.get('CVX_LIB_PATH', 'default'):60s} adarmsnorm {best:6.2f} us  {2 * M * D * 4 / best / 1e6:5.2f} TB/s")
