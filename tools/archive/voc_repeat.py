#!/usr/bin/env python3
"""Dev: is Generator.__call__ bit-identical across repeated calls on the same mel (and across the first call)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa
import torch
import covomix_amd.synthetic as syn
from covomix_amd.vocoder import AttrDict, Generator
h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
gen = Generator(AttrDict(h)).to("cuda:0"); gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
for T in (80, 200):
    mel = (torch.randn(1, 80, T, generator=torch.Generator().manual_seed(2)) * 2 - 6).clamp(-11.52, 2.0).cuda()
    outs = [gen(mel).clone() for _ in range(4)]
    print(T, [float((o - outs[0]).abs().max()) for o in outs], [float((o - outs[-1]).abs().max()) for o in outs])
