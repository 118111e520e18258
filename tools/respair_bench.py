#!/usr/bin/env python3
"""Dev: the fused ResBlock pair kernel (cvx_hifigan_resblock_pair_f16x3) on the two narrow stages of the bench shape
(B = 8, T = 1000: 31 ch x 160032 positions, 62 ch x 80016), every (k, dilation) of config_covomix: us per launch, the
HBM floor (read x + write x') and the matrix floor (3 x 2 convs at 1.2 PFLOP/s executed)."""
import os, sys, torch
from types import SimpleNamespace
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
B = 8
tot = 0.0
for C_, L in ((31, 160032), (62, 80016)):
    np_ = 32 if C_ <= 32 else 64
    Lp = ops.hifigan_cl_rows(L)
    x0 = torch.zeros(B, Lp, np_, device=dev); o = torch.zeros_like(x0)
    if not os.environ.get("ZERO"):          # ZERO=1: all-zero operands (no switching power in the matrix pipe): what the schedule alone costs
        x0[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L, :C_] = torch.randn(B, L, C_, device=dev)
    scale = torch.full((1,), 256.0, device=dev)
    for k in (3, 7, 11):
        for dil in (1, 3, 5):
            def conv(d):
                c = SimpleNamespace(k=k, dil=d)
                c.w16 = ops.hifigan_pack_weight_f16x3((torch.randn(C_, C_, k) / (C_ * k) ** 0.5).to(dev) * (0.0 if os.environ.get("ZERO") else 1.0))
                c.bias16 = torch.zeros(np_, device=dev)
                return c
            c1, c2 = conv(dil), conv(1)
            us = timeit(lambda: ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o, z_scale=scale, flags=int(os.environ.get('FLAGS', '0'))))
            hbm = 2 * B * L * np_ * 4 / 5.0e12 * 1e6
            mm = 2 * 3 * 2 * np_ * np_ * k * B * L / 1.2e15 * 1e6
            tot += us
            print(f"C={C_:3d} k={k:2d} dil={dil}: {us:7.1f} us   (HBM floor {hbm:5.1f}, matrix floor {mm:5.1f})")
print(f"sum over the 18 pairs of the two stages: {tot / 1e3:.2f} ms")
