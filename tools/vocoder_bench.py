#!/usr/bin/env python3
"""Dev: HiFi-GAN generator alone on the bench shape (B = 8 mels of T = 1000 frames, config_covomix, synthetic weights):
ms per call, the fp32-kernel generator as the accuracy reference (rel. l2), and a per-kernel breakdown from torch's profiler.
Env: B, T, REPS, PROFILE=1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.vocoder import AttrDict, Generator
dev = torch.device("cuda:0")
B, T, reps = int(os.environ.get("B", "8")), int(os.environ.get("T", "1000")), int(os.environ.get("REPS", "10"))
vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()}
def make(precision):
    g = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG), precision=precision).to(dev)
    g.load_state_dict(vsd); g.eval(); g.remove_weight_norm()
    return g
gen = make("f16x3")
mel = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(3): y = gen(mel)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps): y = gen(mel)
e.record(); torch.cuda.synchronize()
print(f"generator f16x3: {s.elapsed_time(e) / reps:.2f} ms per call (B={B}, T={T})")
ref = make("fp32")(mel)
print(f"rel l2 vs the fp32-kernel generator: {float((y - ref).norm() / ref.norm()):.2e}")
if os.environ.get("PROFILE", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        y = gen(mel); torch.cuda.synchronize()
    rows = sorted(((ev.key, ev.count, ev.device_time_total) for ev in prof.key_averages() if ev.device_time_total > 0), key=lambda r: -r[2])
    for k, c, t in rows[:14]:
        print(f"  {k[:90]:90s} x{c:4d} {t / 1e3:8.2f} ms")
