#!/usr/bin/env python3
"""Dev: VectorField.prepare() at the bench shape (time MLP, adaptive-norm table, step-invariant to_embed product, activation
pre-scales): ms per call with the two large products on the split-precision GEMM vs on the fp32 kernels, and their distance."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd import acoustic
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
f = acoustic.VectorField(sd, device=dev)
B, T = 8, 1000
d = f.d
ids = torch.randint(0, 500, (B, T, d["streams"]) if d["streams"] > 1 else (B, T), device=dev)
cond = torch.randn(B, T, d["dim_cond"], device=dev)
times, dts = acoustic.evaluation_times(32, "midpoint")
tt = times.to(dev)
for _ in range(3): ctx = f.prepare(ids, cond, tt, True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ctx = f.prepare(ids, cond, tt, True)
e.record(); torch.cuda.synchronize()
print(f"prepare: {s.elapsed_time(e)/10:.3f} ms; table checksum {float(ctx['table'].double().abs().sum()):.6e} base {float(ctx['ws']['base'].double().abs().sum()):.6e}")
tab_new, base_new = ctx["table"].double().clone(), ctx["ws"]["base"].double().clone()
f.split.pop("ada", None); f.split.pop("to_embed.rest"); os.environ["CVX_SKINNY"] = "0"; os.environ["CVX_ADA_F16X3"] = "0"
for _ in range(3): ctx = f.prepare(ids, cond, tt, True)
torch.cuda.synchronize()
s.record()
for _ in range(10): ctx = f.prepare(ids, cond, tt, True)
e.record(); torch.cuda.synchronize()
tab_old, base_old = ctx["table"].double(), ctx["ws"]["base"].double()
print(f"prepare (fp32 GEMMs): {s.elapsed_time(e)/10:.3f} ms")
rel = lambda a, b: float((a - b).norm() / b.norm())
print("table new vs fp32-kernel:", rel(tab_new, tab_old), " base new vs fp32-kernel:", rel(base_new, base_old), " max abs", float((tab_new - tab_old).abs().max()), float((base_new - base_old).abs().max()))
