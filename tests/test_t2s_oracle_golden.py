"""CPU: the text2semantic oracle (oracle/t2s_oracle.py) against the golden vectors the REFERENCE TextToSemantic produced
in the build container (tests/golden/make_golden_t2s.py): sampled tokens from the recorded uniform draws (bit-exact),
teacher-forced logits and encoder output (<= 1e-5 rel-L2).  SURVEY.md section 8f row N1."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

import t2s_oracle as orc
import covomix_amd.synthetic as syn

KW = {
    "cosingle": dict(two_output=False, dim=512, dim_target=512),
    "comix": dict(two_output=True, dim=512, dim_target=1024),
}


def load_case(name):
    g = np.load(os.path.join(GOLDEN, f"t2s_{name}.npz"))
    if name.endswith("_small"):
        sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    else:
        sd = {k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(syn.t2s_param_shapes(**KW[name]), seed=0).items()}
    return g, sd


def test_t2s_param_counts_match_reference_probe():
    n = lambda s: sum(int(np.prod(v)) for v in s.values())
    assert n(syn.t2s_param_shapes(**KW["cosingle"])) == 45_287_312          # SURVEY section 8c probe
    assert n(syn.t2s_param_shapes(**KW["comix"])) == 76_758_584


@pytest.mark.parametrize("name", ["cosingle_small", "comix_small", "cosingle", "comix"])
def test_t2s_oracle_vs_reference_golden(name):
    g, sd = load_case(name)
    src = torch.from_numpy(g["source_ids"])
    uni = torch.from_numpy(g["uniforms"])
    torch.set_num_threads(8)
    o = orc.generate(sd, src, uni, max_length=uni.shape[0])
    assert torch.equal(o["tokens"], torch.from_numpy(g["tokens"]))
    assert torch.equal(o["streams"], torch.from_numpy(g["streams"]))
    assert int(o["streams"][0, :, -1].max()) == 501                  # every fixture ends with a sampled eos
    of = orc.generate(sd, src, uni, forced=torch.from_numpy(g["streams"]))
    assert rel_l2(of["logits"], torch.from_numpy(g["logits"])) < 1e-5
    assert rel_l2(orc.encode(sd, src)[0], torch.from_numpy(g["encoder"])) < 1e-5


@pytest.mark.parametrize("name", ["cosingle_small", "cosingle"])
def test_t2s_oracle_guidance_vs_reference_golden(name):
    """Classifier-free guidance (text2semantic.py:780-792): the tokens the REFERENCE (built with cond_drop_prob > 0) sampled at
    cond_scale = 1.5 from the recorded draws, reproduced bit-exactly; they differ from the unguided run's."""
    _, sd = load_case(name)
    g = np.load(os.path.join(GOLDEN, f"t2s_{name}_cfg.npz"))
    src, uni = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"])
    torch.set_num_threads(8)
    o = orc.generate(sd, src, uni, max_length=uni.shape[0], cond_scale=float(g["cond_scale"]))
    assert torch.equal(o["tokens"], torch.from_numpy(g["tokens"])) and int(o["streams"][0, 0, -1]) == 501
    assert not torch.equal(orc.generate(sd, src, uni, max_length=uni.shape[0])["tokens"], o["tokens"])


def test_t2s_helpers():
    t = torch.tensor([[5, 7, 0, 0], [3, 4, 6, 9]])
    out = orc.set_eos_id(t.clone(), 99, 0)
    assert out.tolist() == [[5, 7, 99, 0, 0], [3, 4, 6, 9, 99]]
    m = orc.mask_after_eos(torch.tensor([[1, 501, 7, 8], [2, 3, 4, 501]]), 501, -1)
    assert m.tolist() == [[1, 501, -1, -1], [2, 3, 4, 501]]
    f = orc.top_k_filter(torch.arange(502, dtype=torch.float32)[None])
    assert int(torch.isfinite(f).sum()) == 51 and bool(torch.isfinite(f[0, -51:]).all())
