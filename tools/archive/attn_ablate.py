#!/usr/bin/env python3
"""Dev: energy ablations of the split-precision attention kernel (Bt = 16, T = 1000, H = 16, random operands): time, board
power, shader clock and joules per launch for the library selected with CVX_LIB_PATH (builds with -DCVX_ATT_ABLATE=<bits>,
attention_f16x3.hip: 1 no DMA after tile 0, 2 no LDS fragment reads after tile 0, 4 no v_exp_f32, 8 no P.V MFMAs, 16 no K.Q
MFMAs; the ablated results are wrong by construction - only the cost matters)."""
import os, re, subprocess, sys, threading, time, torch
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
Bt, T, H = 16, 1000, 16
q = torch.randn(Bt * T, 2 * H * 64, device=dev); v = torch.randn(Bt * H * 64, 1024, device=dev)
if os.environ.get("ZERO") == "1": q.zero_(); v.zero_()
qh, ql = ops.split_act_f16(q); vh, vl = ops.split_act_f16(v)
oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
fn = lambda: ops.attention_f16x3((qh, ql), (vh, vl), None, Bt, T, H, 0.125, out_split=(oh, ol))
for _ in range(5): fn()
torch.cuda.synchronize()
t0 = time.time(); n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < float(os.environ.get("SECS", "3")):
    for _ in range(50): fn()
    n += 50
    torch.cuda.synchronize()
e.record(); torch.cuda.synchronize()
t1 = time.time(); stop.set()
sel = [x for x in samples if t0 + 0.8 < x[0] < t1 - 0.1]
pw = sum(x[1] for x in sel) / max(len(sel), 1); ck = sum(x[2] for x in sel) / max(len(sel), 1)
us = s.elapsed_time(e) / n * 1e3
tag = os.path.basename(os.environ.get("CVX_LIB_PATH", "default")) + (" ZERO" if os.environ.get("ZERO") == "1" else "")
print(f"{tag:34s} {us:7.1f} us  {pw:6.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:.3f} J per launch  ({len(sel)} samples)")
