#!/usr/bin/env python3
"""Dev: find what makes tests/test_ragged_gpu.py::test_vocoder_ragged_batch_vs_b1_and_oracle[500-f16x3] fail inside the full suite:
run a set of predecessor tests IN THIS PROCESS (PRE=<pytest args>), then the vocoder scenario with per-item diagnostics."""
import os, sys, warnings, torch, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pre = os.environ.get("PRE", "")
if pre:
    rc = pytest.main(pre.split() + ["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"])
    print("predecessors rc", rc, flush=True)
import covomix_oracle as orc
import covomix_amd.synthetic as syn
from covomix_amd import ops
from covomix_amd.vocoder import AttrDict, Generator
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return float((a - b).norm() / b.norm())
c0 = 500
h = dict(syn.HIFIGAN_COVOMIX_CONFIG); h["upsample_initial_channel"] = c0
vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
folded = orc.fold_weight_norm(vsd)
def make(prec="f16x3"):
    g_ = Generator(AttrDict(h), precision=prec).to("cuda:0"); g_.load_state_dict(vsd); g_.eval(); g_.remove_weight_norm(); return g_
gen = make()
T = [57, 120, 3, 88, 119]
g = torch.Generator().manual_seed(c0)
mels = [(torch.randn(80, t, generator=g) * 2 - 6).clamp(-11.52, 2.0) for t in T]
warnings.simplefilter("always")
gen((torch.randn(len(T), 80, max(T), generator=g) * 2 - 6).cuda())
wavs = gen.ragged([m.cuda() for m in mels])
refs = [orc.hifigan_forward(folded, h, m[None])[0] for m in mels]
folded64 = {k: v.double() for k, v in folded.items()}
refs64 = [orc.hifigan_forward(folded64, h, m[None].double())[0] for m in mels]
print("torch threads", torch.get_num_threads(), "mkldnn", torch.backends.mkldnn.enabled)
for t, r, r64 in zip(T, refs, refs64):
    print(f"T={t}: oracle fp32 vs oracle fp64 {rel(r, r64):.2e}", flush=True)
refs = [r.float() for r in refs64]
gen32 = make("fp32")
for t, m, w, ref in zip(T, mels, wavs, refs):
    single = gen(m.cuda()); fresh = make()(m.cuda()); s32 = gen32(m.cuda())
    d = (single.cpu().double() - ref.double()).abs()[0]
    bad = (d > 1e-3 * ref.abs().max()).nonzero().flatten()
    print(f"T={t}: ragged {rel(w, ref):.2e} single {rel(single, ref):.2e} fresh-gen {rel(fresh, ref):.2e} fp32-gen {rel(s32, ref):.2e}"
          f"  bad samples {bad.numel()} range {(int(bad.min()), int(bad.max())) if bad.numel() else None} of {ref.shape[-1]}", flush=True)
print("sat flag now:", ops.saturation_query(reset=False))
