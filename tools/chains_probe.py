"""A/B of the two-chain schedule (CVX_CHAINS=2: the batch cut into two halves on two streams) against the single chain:
bit-identity of the sampled mel and time per 32-NFE solve at BASELINE config 3 (B=8, T=1000)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import covomix_amd.synthetic as syn
from covomix_amd.acoustic import VectorField, FlowMatchingSampler

dev = torch.device("cuda")
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
field = VectorField(sd, dev)
B, T = 8, 1000
inp = syn.synthetic_inputs("vomix", B, T, 400, seed=1234)
ids, cond = inp["phoneme_ids"].to(dev), inp["cond"].to(dev)
y0 = torch.randn(B, T, 80, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
out = {}
for chains in ("1", "2", "1", "2"):
    os.environ["CVX_CHAINS"] = chains
    smp = FlowMatchingSampler(field, nfe=32)
    smp.sample(phoneme_ids=ids, cond=cond, cond_scale=0.7, y0=y0)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(2):
        y = smp.sample(phoneme_ids=ids, cond=cond, cond_scale=0.7, y0=y0)
    torch.cuda.synchronize(); dt = (time.time() - t) / 2
    out[chains] = y
    print(f"chains={chains}: {dt * 1e3:.1f} ms per solve = {B * T / dt:.0f} frames/s (acoustic only)", flush=True)
print("bit-identical:", torch.equal(out["1"], out["2"]), "max abs diff", float((out["1"] - out["2"]).abs().max()))
