#!/usr/bin/env python3
"""Dev: board power and shader clock (rocm-smi, sampled in a thread) while ONE kernel of the hot path runs back to back for a few
seconds: the large-problem GEMM (ff1 shape, model epilogue), the split-precision attention (random and all-zero operands),
the adaptive RMSNorm and the fused ResBlock pair kernel.  Which kernels sit at the 1400 W cap."""
import os, re, subprocess, sys, threading, time, math, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import ops
dev = torch.device("cuda:0")
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([\d.]+)", out); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        if p and c: samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        time.sleep(0.2)
th = threading.Thread(target=sampler, daemon=True); th.start()
def measure(name, fn, secs=3.0):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    t1 = time.time()
    sel = [x for x in samples if t0 + 0.8 < x[0] < t1 - 0.1]
    pw = sum(x[1] for x in sel) / max(len(sel), 1); ck = sum(x[2] for x in sel) / max(len(sel), 1)
    us = s.elapsed_time(e) / n * 1e3
    print(f"{name:44s} {us:8.1f} us per launch   {pw:6.0f} W   sclk {ck:5.0f} MHz   ({len(sel)} samples)   {pw * us * 1e-6:.3f} J per launch")
M, T = 16000, 1000
g = torch.Generator().manual_seed(0)
for zero in (False, True):
    a = torch.randn(M, 1024, generator=g).to(dev); w = (torch.randn(4096, 1024, generator=g) / 32).to(dev); b = torch.randn(4096, generator=g).to(dev)
    if zero: a.zero_(); w.zero_()
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, 1024, dev); ops.split_act_f16(a, il)
    o = ops.SplitIL(M, 4096, dev); dummy = torch.empty(M, 4096, device=dev)
    measure("GEMM ff1 (bias + GELU + split)" + (" ZERO operands" if zero else ""), lambda: ops.gemm(a, w, dummy, w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=o, write_f32=False))
Bt, H = 16, 16
for zero in (False, True):
    q = torch.randn(Bt * T, 2 * H * 64, device=dev); v = torch.randn(Bt * H * 64, 1024, device=dev)
    if zero: q.zero_(); v.zero_()
    qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
    oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
    measure("attention f16x3" + (" ZERO operands" if zero else ""), lambda: ops.attention_f16x3((qh, ql), (vh, vl), None, Bt, T, H, 0.125, out_split=(oh, ol)))
for what in ("q, k zero (v random)", "v zero (q, k random)"):
    q = torch.randn(Bt * T, 2 * H * 64, device=dev); v = torch.randn(Bt * H * 64, 1024, device=dev)
    if what.startswith("q"): q.zero_()
    else: v.zero_()
    qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
    measure("attention f16x3, " + what, lambda: ops.attention_f16x3((qh, ql), (vh, vl), None, Bt, T, H, 0.125, out_split=(oh, ol)))
q = torch.randn(Bt * T, 2 * H * 64, device=dev); v = torch.randn(Bt * H * 64, 1024, device=dev)
qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
ql.zero_(); vl.zero_()
measure("attention f16x3, lo halves zero", lambda: ops.attention_f16x3((qh, ql), (vh, vl), None, Bt, T, H, 0.125, out_split=(oh, ol)))
x = torch.randn(M, 1024, device=dev); gam = torch.randn(16, 1024, device=dev); bet = torch.randn(16, 1024, device=dev)
n16 = ops.SplitIL(M, 1024, dev)
try:
    measure("adaptive RMSNorm (fp32 in, split out)", lambda: ops.adarmsnorm(x, gam[0], bet[0], None, out_split=n16))
except Exception as ex:
    print("adarmsnorm skipped:", ex)
C_, L, B = 31, 160032, 8
Lp = ops.hifigan_cl_rows(L)
x0 = torch.zeros(B, Lp, 32, device=dev); o2 = torch.zeros_like(x0)
x0[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L, :C_] = torch.randn(B, L, C_, device=dev)
def conv(d, k=7):
    c = SimpleNamespace(k=k, dil=d); c.w16 = ops.hifigan_pack_weight_f16x3((torch.randn(C_, C_, k) / (C_ * k) ** 0.5).to(dev)); c.bias16 = torch.zeros(32, device=dev); return c
c1, c2 = conv(3), conv(1)
sc = torch.full((1,), 256.0, device=dev)
measure("fused ResBlock pair, 31 ch, k = 7", lambda: ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o2, z_scale=sc))
stop.set()
