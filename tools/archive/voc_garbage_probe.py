#!/usr/bin/env python3
"""Dev: HiFi-GAN with every torch.empty buffer pre-poisoned (the caching allocator hands back NaN-filled blocks): finds reads of
uninitialised memory that a fresh process hides."""
import os, sys, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import covomix_oracle as orc
import covomix_amd.synthetic as syn
from covomix_amd.vocoder import AttrDict, Generator
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return float((a - b).norm() / b.norm())
def dirty_lds():
    """leave every CU's LDS full of non-zero fp16 data (the large-problem GEMM uses 136 KB per block)"""
    import math
    from covomix_amd import ops
    M, K, N = 16000, 1024, 1024
    a = torch.randn(M, K, device="cuda") * 100; w = torch.randn(N, K, device="cuda")
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, K, a.device); ops.split_act_f16(a, il)
    ops.gemm(a, w, torch.empty(M, N, device="cuda"), w_split=ws, w_il=wil, a_split=il)
    torch.cuda.synchronize()
def poison(val):
    dirty_lds()
    blocks = [torch.full((64 << 20,), val, device="cuda") for _ in range(48)]      # 12 GB
    blocks += [torch.full((n,), val, device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22) for _ in range(16)]
    del blocks
warnings.simplefilter("always")
for c0 in (500, 64):
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG); h["upsample_initial_channel"] = c0
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    folded = orc.fold_weight_norm(vsd)
    for val in (float("nan"), 3.0e4):
        gen = Generator(AttrDict(h)).to("cuda:0"); gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
        g = torch.Generator().manual_seed(c0)
        for T in (57, 120, 88):
            m = (torch.randn(80, T, generator=g) * 2 - 6).clamp(-11.52, 2.0)
            ref = orc.hifigan_forward(folded, h, m[None])[0]
            poison(val)
            w = gen(m.cuda())
            print(f"c0={c0} poison={val} T={T}: rel-L2 vs oracle {rel(w, ref):.3e}", flush=True)
