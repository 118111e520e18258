#!/usr/bin/env python3
"""Dev: joules per launch of the large-problem GEMM variants (cvx_gemm_split_io.flags: 0 eight-phase 16x16x32 persistent = default,
4 the same with one tile per block, 2 eight-phase 32x32x16, 1 round-1 two-stage) on the ff1 and to_qkv shapes with the model's
epilogues, random operands: board power and shader clock from rocm-smi while ONE variant runs back to back for ~3 s."""
import os, re, subprocess, sys, threading, time, math, torch
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
threading.Thread(target=sampler, daemon=True).start()
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
    print(f"{name:52s} {us:7.1f} us  {pw:6.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:.3f} J per launch", flush=True)
M, T = 16000, 1000
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 1024, generator=g).to(dev)
il = ops.SplitIL(M, 1024, dev); ops.split_act_f16(a, il)
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)); ang = torch.arange(T).float()[:, None] * inv[None, :]
rope = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
for name, N in (("ff1 (bias + GELU + split)", 4096), ("to_qkv (RoPE + split q|k + V^T)", 3072)):
    w = (torch.randn(N, 1024, generator=g) / 32).to(dev); b = torch.randn(N, generator=g).to(dev)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    dummy = torch.empty(M, N, device=dev)
    if N == 4096:
        kw = dict(bias=b, act=1, out_split=ops.SplitIL(M, N, dev), write_f32=False)
    else:
        qk = (torch.empty(M, 2048, dtype=torch.float16, device=dev), torch.empty(M, 2048, dtype=torch.float16, device=dev))
        vt = (torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev), torch.zeros(16 * 16 * 64, 1024, dtype=torch.float16, device=dev))
        kw = dict(rope=rope, rope_cols=2048, out_split=qk, vt_split=vt, write_f32=False)
    for flags, tag in ((0, "eight-phase 16x16x32, persistent (default)"), (4, "eight-phase 16x16x32, one tile per block"),
                       (2, "eight-phase 32x32x16"), (1, "two-stage (round 1)")):
        ops._GEMM_FLAGS = flags
        measure(f"{name[:18]:18s} {tag}", lambda: ops.gemm(a, w, dummy, w_split=ws, w_il=wil, a_split=il, **kw))
ops._GEMM_FLAGS = 0
stop.set()
