#!/usr/bin/env python3
"""Dev: timeline of the wide-stage ResBlock convolution kernel (conv_f16x3_kernel).  Needs a -DCVX_CONV_TRACE build of the
library (CVX_LIB_PATH): every block leaves 100 MHz stamps (start | end of the main loop | end) in its own output rows.
Prints, per shape: blocks, makespan, and per block the start offset, main-loop and epilogue durations (mean / max, us)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "8"))
for C_, L in ((250, 5000), (125, 20000)):
    np_ = 256 if C_ > 128 else 128
    Lp = ops.hifigan_cl_rows(L)
    def split(x):
        hi = x.half(); return hi, (x - hi.float()).half()
    x0 = torch.zeros(B, Lp, np_, device=dev)
    x0[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L, :C_] = torch.randn(B, L, C_, device=dev)
    z = split(x0 * (0.0 if os.environ.get("ZERO") else 1.0))          # ZERO=1: all-zero operands (no switching power in the MFMAs)
    if np_ == 256:
        n256, n192 = ((L + 255) // 256) * B, ((L + 191) // 192) * B
        tmb = 192 if ((n192 + 255) // 256) * 3 < ((n256 + 255) // 256) * 4 else 256
    else:
        tmb = 256
    for k, dil in ((3, 1), (7, 3), (11, 5)):
        w16 = ops.hifigan_pack_weight_f16x3((torch.randn(C_, C_, k) / (C_ * k) ** 0.5).to(dev) * (0.0 if os.environ.get("ZERO") else 1.0))
        bias = torch.zeros(np_, device=dev)
        o = torch.zeros_like(x0)
        oz = (torch.zeros_like(z[0]), torch.zeros_like(z[1]))
        for _ in range(2):
            ops.hifigan_conv1d_f16x3(z, w16, bias, B, L, ksize=k, dil=dil, res=x0, out_x=o, out_scale=0.0, out_z=oz)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.hifigan_conv1d_f16x3(z, w16, bias, B, L, ksize=k, dil=dil, res=x0, out_x=o, out_scale=0.0, out_z=oz)
        e.record(); torch.cuda.synchronize()
        rows = o[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L:tmb, :8].contiguous().view(torch.int64).view(-1, 4).cpu()
        t0 = rows[:, 0].min()
        st, loop, epi = (rows[:, 0] - t0).double() / 100, (rows[:, 1] - rows[:, 0]).double() / 100, (rows[:, 2] - rows[:, 1]).double() / 100
        span = float((rows[:, 2].max() - t0)) / 100
        late = int((st > 5).sum())
        print(f"C={C_} L={L} k={k}: {rows.shape[0]} blocks of {tmb} rows, {s.elapsed_time(e) / 5 * 1e3:6.1f} us/launch, makespan {span:6.1f} us | "
              f"start mean {st.mean():5.1f} max {st.max():5.1f} ({late} blocks start > 5 us) | loop mean {loop.mean():5.1f} max {loop.max():5.1f} | "
              f"epilogue mean {epi.mean():5.1f} max {epi.max():5.1f}")
        if os.environ.get("ROUNDS"):
            cu = rows[:, 3]
            first = st < 5
            print(f"   first-round blocks: loop {loop[first].mean():5.1f}  epi {epi[first].mean():5.1f} | later: loop {loop[~first].mean() if late else 0:5.1f}  epi {epi[~first].mean() if late else 0:5.1f}")
