#!/usr/bin/env python3
"""Dev: where a tile of the fused ResBlock pair kernel spends its cycles.  Needs a -DCVX_PAIR_TRACE build of the library
(CVX_HIPCC_FLAGS=-DCVX_PAIR_TRACE python -c "import build...", CVX_LIB_PATH): per wave, s_memtime cycles summed over the
block's tiles for prologue | pass 1 | epilogue 1 | pass 2 | next z tile (convert + LDS store) | epilogue 2."""
import os, sys, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
B = 8
for C_, L in ((31, 160032), (62, 80016)):
    np_ = 32 if C_ <= 32 else 64
    Lp = ops.hifigan_cl_rows(L)
    x0 = torch.zeros(B, Lp, np_, device=dev)
    x0[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L, :C_] = torch.randn(B, L, C_, device=dev)
    scale = torch.full((1,), 256.0, device=dev)
    for k, dil in ((3, 1), (7, 3), (11, 5)):
        def conv(d):
            c = SimpleNamespace(k=k, dil=d)
            c.w16 = ops.hifigan_pack_weight_f16x3((torch.randn(C_, C_, k) / (C_ * k) ** 0.5).to(dev))
            c.bias16 = torch.zeros(np_, device=dev)
            return c
        c1, c2 = conv(dil), conv(1)
        o = torch.zeros_like(x0)
        ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o, out_scale=0.0, z_scale=scale)
        torch.cuda.synchronize()
        t = o.view(-1).view(torch.int64)[: 256 * 8 * 8].view(256, 8, 8).double()
        tiles = B * ((L + 256 - (k - 1) - 1) // (256 - (k - 1))) / 256
        m = t.mean(dim=(0, 1))[:6] / tiles
        print(f"C={C_} k={k} dil={dil}: per tile cycles  pass1 {m[1]:7.0f}  epi1 {m[2]:6.0f}  pass2 {m[3]:7.0f}  next-z {m[4]:6.0f}  epi2 {m[5]:6.0f}   total {m[1:].sum():7.0f}  (prologue {t.mean(dim=(0,1))[0]:.0f})")
