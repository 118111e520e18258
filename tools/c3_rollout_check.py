#!/usr/bin/env python3
"""Dev: the WHOLE 32-NFE rollout of BASELINE config 3 at its own size (B = 8 x T = 1000: 16,000 rows per evaluation, the deferred-norm
path) against the CPU oracle - about four minutes of CPU for the oracle, which is why the test suite compares 2 and 8 NFE at this
size and the 32-NFE rollout at B = 2.  Env: B, NFE."""
import os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import covomix_oracle as orc
import covomix_amd.synthetic as syn
from covomix_amd.conditional_model import CoVoMixModel
B, nfe = int(os.environ.get("B", "8")), int(os.environ.get("NFE", "32"))
shapes = syn.acoustic_param_shapes(dim=1024, dim_cond=160, dim_emb=1024, depth=8, heads=16, streams=2)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
inp = syn.synthetic_inputs("vomix", B, 1000, 400, seed=97531)
model = CoVoMixModel.from_state_dict(sd, nfe=nfe).eval().to("cuda:0")
out = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), 0.7, y0=inp["y0"]).cpu()
t0 = time.time()
ref = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=nfe)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print(f"C3 VoMix B={B} T=1000 {nfe}-NFE rollout, CVX_DEFER_NORM={os.environ.get('CVX_DEFER_NORM', '1')}: rel-L2 vs oracle {rel(out, ref):.3e}, "
      f"worst utterance {max(rel(out[b], ref[b]) for b in range(B)):.3e} (oracle {time.time() - t0:.0f} s)")
