"""CPU: the HuBERT tokeniser oracle (oracle/hubert_oracle.py) against the golden vectors
tests/golden/hubert_base.npz, produced by the reference's own HubertModel / ApplyKmeans classes
(tests/golden/make_golden_hubert.py)."""
import os

import numpy as np
import pytest
import torch

import hubert_oracle as ho
from covomix_amd import synthetic

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hubert_base.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def sd():
    return synthetic.hubert_state_dict(seed=0)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_recipe_covers_the_reference_state_dict(gold, sd):
    assert list(sd.keys()) == [str(n) for n in gold["param_names"]]


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_matches_reference_features_and_codes(gold, sd, tag):
    torch.set_num_threads(8)
    wav = gold[f"{tag}_wav"]
    assert ho.frames_for(len(wav)) == gold[f"{tag}_feat12"].shape[0]
    with torch.no_grad():
        conv = ho.conv_features(sd, torch.from_numpy(wav).view(1, -1))[0]
        assert rel(conv.numpy(), gold[f"{tag}_conv"]) < 2e-6
        for layer in (1, 6, 12):
            f = ho.get_feats(sd, wav, layer=layer)
            # two fp32 evaluations of a 12-layer post-LN stack with peaky attention: the reference itself is
            # 1e-6 / 4e-6 / 1.1e-5 from an fp64 evaluation at layers 1 / 6 / 12 (measured), so that is the floor
            assert rel(f.numpy(), gold[f"{tag}_feat{layer}"]) < {1: 3e-6, 6: 1.2e-5, 12: 3e-5}[layer], layer
        codes = ho.apply_kmeans(synthetic.hubert_kmeans_centers(seed=0), f)
    np.testing.assert_array_equal(codes, gold[f"{tag}_codes"])
    # second fixture: centres near the features themselves (91 distinct labels over the three waveforms, none dominant - the
    # N(0, 1) centres above put almost every frame into cell 452); labelled by the reference's ApplyKmeans
    pool = np.concatenate([gold[f"{t}_feat12"] for t in ("a", "b", "c")], 0)
    near = ho.apply_kmeans(synthetic.hubert_kmeans_centers_near(pool, seed=0), f)
    safe = gold[f"{tag}_margin_near"] > 3e-2
    assert safe.mean() > 0.9 and len(np.unique(gold[f"{tag}_codes_near"])) >= min(len(near), 16)
    np.testing.assert_array_equal(near[safe], gold[f"{tag}_codes_near"][safe])


def test_fp64_oracle_is_the_common_limit(gold, sd):
    """The fp64 evaluation of the restatement is as close to the reference's fp32 output as fp32 rounding allows."""
    with torch.no_grad():
        f64 = ho.get_feats(sd, gold["b_wav"], layer=12, dtype=torch.float64).numpy()
    assert rel(gold["b_feat12"], f64) < 2e-5


def test_oracle_normalized_waveform_and_chunking(gold, sd):
    wav = gold["a_wav"]
    with torch.no_grad():
        f = ho.get_feats(sd, wav, layer=12, normalize=True)
        assert rel(f.numpy(), gold["a_feat12_normalized"]) < 3e-5
        # max_chunk splits the waveform and concatenates the per-chunk features (hubert_feature_reader.py:70-78)
        parts = ho.get_feats(sd, wav, layer=1, max_chunk=4000)
        want = torch.cat([ho.get_feats(sd, wav[:4000], layer=1), ho.get_feats(sd, wav[4000:], layer=1)], 0)
    assert parts.shape == want.shape and torch.equal(parts, want)


@pytest.mark.parametrize("orig,new", [(8000, 16000), (44100, 16000), (16000, 8000)])
def test_resampler_restatement_properties(orig, new):
    """torchaudio is absent (parity unpinned): check what the windowed-sinc bank must satisfy, and scipy's polyphase
    resampler (a different low-pass design) as an independent reference well inside the pass band."""
    from scipy.signal import resample_poly
    import math
    n = orig // 4
    t = np.arange(n) / orig
    f0 = 0.11 * min(orig, new)
    wav = (0.5 * np.sin(2 * np.pi * f0 * t) + 0.25).astype(np.float32)
    y = ho.sinc_resample(wav, orig, new)
    assert y.shape == (math.ceil(new * n / orig),) and y.dtype == np.float32
    tt = np.arange(len(y)) / new
    want = 0.5 * np.sin(2 * np.pi * f0 * tt) + 0.25
    edge = 64
    assert np.abs(y[edge:-edge] - want[edge:-edge]).max() < 2e-3            # in-band tone + DC pass with unit gain, no delay
    g = math.gcd(orig, new)
    sp = resample_poly(wav.astype(np.float64), new // g, orig // g)
    assert np.abs(y[edge:-edge] - sp[edge:len(y) - edge]).max() < 5e-3
    assert np.array_equal(ho.sinc_resample(wav, orig, orig), wav)
