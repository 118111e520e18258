#!/usr/bin/env python3
"""BASELINE config 2: VoSingle acoustic model, 32 NFE, B=1, 500-frame utterance (200-frame prompt) on one MI355X.
Latency-oriented companion of bench.py (which measures config 3/4).  Prints frames/s and ms per utterance."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.conditional_model import CoVoMixModel
shapes = syn.acoustic_param_shapes(dim_cond=80, streams=1)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
model = CoVoMixModel.from_state_dict(sd, nfe=32).eval().to("cuda:0")
B = int(os.environ.get("B", "1")); T = int(os.environ.get("T", "500"))
inp = syn.synthetic_inputs("vosingle", B, T, 200, seed=1234)
ids, cond, mask = inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda()
for _ in range(2):
    model.synthesis_sample(ids, cond, mask, 0.7)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    model.synthesis_sample(ids, cond, mask, 0.7)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"VoSingle B={B} T={T} 32 NFE: {dt*1e3:.1f} ms per call, {B*T/dt:.0f} mel-frames/s (precision {model.precision})")
