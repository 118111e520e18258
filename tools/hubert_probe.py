"""HuBERT tokeniser (row N4): distance of this build's and of the reference's features from an fp64 evaluation of the
oracle at layers 1 / 6 / 12, and time per 10-s utterance.  HUBERT_PRECISION=fp32 selects the fp32 GEMM mode."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import hubert_oracle as ho
from covomix_amd import synthetic
from covomix_amd.hubert import HubertEncoder
sd = synthetic.hubert_state_dict(seed=0); g = np.load(os.path.join(ROOT, "tests", "golden", "hubert_base.npz"))
enc = HubertEncoder(sd, precision=os.environ.get('HUBERT_PRECISION', 'f16x3'))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for tag in "abc":
    wav = g[f"{tag}_wav"]
    with torch.no_grad():
        c64 = ho.conv_features(sd, torch.from_numpy(wav).view(1, -1), torch.float64)[0].numpy()
    conv = enc.conv_features(torch.from_numpy(wav).cuda()).cpu().numpy()
    print(tag, "conv: ours-f64", rel(conv, c64), "ref-f64", rel(g[f"{tag}_conv"], c64))
    for L in (1, 6, 12):
        with torch.no_grad():
            f64 = ho.get_feats(sd, wav, layer=L, dtype=torch.float64).numpy()
        f = enc.extract_features(torch.from_numpy(wav).cuda(), L).cpu().numpy()
        print(tag, L, "ours-f64", rel(f, f64), "ref-f64", rel(g[f"{tag}_feat{L}"], f64), "ours-ref", rel(f, g[f"{tag}_feat{L}"].astype(np.float64)))
import time
wav = torch.randn(160000).cuda() * 0.1
for stepped in (False, True):
    e = HubertEncoder(sd, precision=enc.precision, stepped=stepped)
    for _ in range(30): e.extract_features(wav, 12)          # also lets the clocks ramp after the CPU-only phase above
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): e.extract_features(wav, 12)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print(f"10 s utterance, {'Python-stepped' if e.stepped else 'one C call'}: host enqueue {(t1 - t0) * 50:.2f} ms, total {(t2 - t0) * 50:.2f} ms")
