#!/usr/bin/env python3
"""Micro-benchmarks of the individual kernels at BASELINE config-3 sizes (B=8, T=1000, CFG batch 2B).
Prints one line per kernel: time, TFLOP/s or GB/s.  Dev tool (not part of the judged contract)."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    Bt, T, H, D = 16, 1000, 16, 1024
    M = Bt * T
    r = lambda *s: torch.randn(*s, device=dev)
    print(torch.cuda.get_device_name(0))
    for (n, k, name) in [(1024, 1024, "out-proj"), (3072, 1024, "qkv"), (4096, 1024, "ff1"), (1024, 4096, "ff2"),
                         (1024, 2048, "skip"), (1024, 80, "embed-x"), (80, 1024, "to_pred")]:
        a, w, c = r(M, k), r(n, k) / math.sqrt(k), torch.empty(M, n, device=dev)
        t = timeit(lambda: ops.gemm(a, w, c))
        print(f"gemm {name:9s} M={M} N={n} K={k}: {t*1e3:8.3f} ms  {2*M*n*k/t/1e12:7.2f} TFLOP/s")
    a, w, c, b = r(M, 1024), r(4096, 1024) / 32, torch.empty(M, 4096, device=dev), r(4096)
    t = timeit(lambda: ops.gemm(a, w, c, bias=b, act=ops.ACT_GELU))
    print(f"gemm ff1+bias+gelu: {t*1e3:8.3f} ms  {2*M*4096*1024/t/1e12:7.2f} TFLOP/s")
    qkv, o = r(Bt, T, 3 * H * 64), torch.empty(Bt, T, H * 64, device=dev)
    t = timeit(lambda: ops.attention(qkv, o, Bt, T, H, 0.125))
    print(f"attention Bt={Bt} T={T} H={H}: {t*1e3:8.3f} ms  {4*Bt*H*T*T*64/t/1e12:7.2f} TFLOP/s")
    x, g, be, y = r(M, D), r(D), r(D), torch.empty(M, D, device=dev)
    t = timeit(lambda: ops.adarmsnorm(x, g, be, y))
    print(f"adarmsnorm: {t*1e6:8.1f} us  {2*M*D*4/t/1e9:7.1f} GB/s")
    wd, bd = r(D, 31), r(D)
    t = timeit(lambda: ops.dwconv31_gelu_res(x, wd, bd, y, Bt, T))
    print(f"dwconv31: {t*1e6:8.1f} us  {2*M*D*4/t/1e9:7.1f} GB/s (algorithmic)")
    # vocoder convs at B=8, T=1000
    for (c, L, k, d) in [(250, 5001, 11, 5), (125, 20004, 7, 3), (62, 80016, 3, 1), (31, 160032, 11, 1)]:
        xx = r(8, c, L)
        wp = ops.hifigan_pack_weight(torch.randn(c, c, k) / math.sqrt(c * k), False).to(dev)
        bb, oo = r(c), torch.empty(8, c, L, device=dev)
        pad = (k * d - d) // 2
        t = timeit(lambda: ops.hifigan_conv1d(xx, wp, bb, oo, cout=c, ksize=k, dil=d, pad=pad, in_slope=0.1, res=xx))
        print(f"conv1d C={c} L={L} k={k} d={d}: {t*1e3:8.3f} ms  {2*8*c*c*k*L/t/1e12:7.2f} TFLOP/s")


if __name__ == "__main__":
    main()


def f16x3():
    Bt, T = 16, 1000
    M = Bt * T
    r = lambda *s: torch.randn(*s, device=dev)
    for (n, k, name) in [(1024, 1024, "out-proj"), (3072, 1024, "qkv"), (4096, 1024, "ff1"), (1024, 4096, "ff2"), (1024, 2048, "skip")]:
        a, w, c = r(M, k), r(n, k) / math.sqrt(k), torch.empty(M, n, device=dev)
        ws = ops.split_f16(w)
        t = timeit(lambda: ops.gemm(a, w, c, w_split=ws))
        print(f"f16x3 {name:9s} M={M} N={n} K={k}: {t*1e3:8.3f} ms  {2*M*n*k/t/1e12:7.2f} TFLOP/s (fp32-equivalent)")


if __name__ == "__main__" and os.environ.get("F16X3", "1") == "1":
    f16x3()


def f16x3_dma():
    Bt, T = 16, 1000
    M = Bt * T
    r = lambda *s: torch.randn(*s, device=dev)
    for (n, k, name) in [(1024, 1024, "out-proj"), (3072, 1024, "qkv"), (4096, 1024, "ff1"), (1024, 4096, "ff2")]:
        a, w, c = r(M, k), r(n, k) / math.sqrt(k), torch.empty(M, n, device=dev)
        ws, asp = ops.split_f16(w), ops.split_act_f16(a)
        t = timeit(lambda: ops.gemm(a, w, c, w_split=ws, a_split=asp))
        print(f"f16x3-dma[{os.environ.get('CVX_GEMM_STAGES','4')}st] {name:9s} N={n} K={k}: {t*1e3:8.3f} ms  {2*M*n*k/t/1e12:7.2f} TFLOP/s")


if __name__ == "__main__" and os.environ.get("F16X3", "1") == "1":
    f16x3_dma()


def attn_f16x3():
    Bt, T, H = 16, 1000, 16
    M = Bt * T
    qk = (torch.randn(M, 2 * H * 64, device=dev).half(), (torch.randn(M, 2 * H * 64, device=dev) * 1e-3).half())
    Tp = 1024
    vt = (torch.randn(Bt * H * 64, Tp, device=dev).half(), (torch.randn(Bt * H * 64, Tp, device=dev) * 1e-3).half())
    oh = torch.empty(Bt, T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
    t = timeit(lambda: ops.attention_f16x3(qk, vt, None, Bt, T, H, 0.125, out_split=(oh, ol)))
    print(f"attention f16x3 Bt={Bt} T={T} H={H}: {t*1e3:8.3f} ms  {4*Bt*H*T*T*64/t/1e12:7.2f} TFLOP/s (fp32-equivalent)")


if __name__ == "__main__" and os.environ.get("F16X3", "1") == "1":
    attn_f16x3()
