#!/usr/bin/env python3
"""Dev: N generator calls on the bench shape and nothing else (for rocprofv3 traces of the vocoder alone).  Env: B, T, CALLS."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.vocoder import AttrDict, Generator
dev = torch.device("cuda:0")
B, T, calls = int(os.environ.get("B", "8")), int(os.environ.get("T", "1000")), int(os.environ.get("CALLS", "10"))
vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()}
g = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)).to(dev)
g.load_state_dict(vsd); g.eval(); g.remove_weight_norm()
mel = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(calls): y = g(mel)
torch.cuda.synchronize()
