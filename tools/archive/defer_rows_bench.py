#!/usr/bin/env python3
"""Dev: one 32-NFE VoMix solve (acoustic model only, CFG) at B x T frames - ms per solve with and without the deferred norm, to
place the row-count rule of VectorField._defers (the deferred forms need the large-problem kernel; the round-3 flow may run the
N = 1024 products on the medium-problem kernel where the large one's rounds come out part-empty).  Env: SHAPES="8x1000,10x1024,...", REPS."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.conditional_model import CoVoMixModel
shapes = syn.acoustic_param_shapes(dim=1024, dim_cond=160, dim_emb=1024, depth=8, heads=16, streams=2)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
model = CoVoMixModel.from_state_dict(sd, nfe=32).eval().to("cuda:0")
reps = int(os.environ.get("REPS", "3"))
for spec in os.environ.get("SHAPES", "8x1000,10x1024,5x1000,6x1000,12x1000").split(","):
    B, T = (int(v) for v in spec.split("x"))
    inp = syn.synthetic_inputs("vomix", B, T, int(0.4 * T), seed=1)
    a = [inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda()]
    out = []
    for mode in ("1", "0", "1", "0"):
        os.environ["CVX_DEFER_NORM"] = mode
        model.synthesis_sample(*a, 0.7, y0=inp["y0"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model.synthesis_sample(*a, 0.7, y0=inp["y0"])
        torch.cuda.synchronize()
        out.append((mode, (time.perf_counter() - t0) / reps * 1e3))
    print(f"B={B} T={T} ({2 * B * T} rows): " + "  ".join(f"defer={m}: {t:.1f} ms" for m, t in out), flush=True)
