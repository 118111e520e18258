"""CPU: the oracle (oracle/covomix_oracle.py) against the golden vectors the REFERENCE modules
produced in the build container (tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

import covomix_oracle as orc
import covomix_amd.synthetic as syn


def _state(kind, **kw):
    two = kind in ("vomix", "vomix2out")
    shapes = syn.acoustic_param_shapes(dim=kw.get("dim", 1024), dim_cond=160 if two else 80,
                                       dim_emb=kw.get("dim_emb", 1024), depth=kw.get("depth", 8),
                                       heads=kw.get("heads", 16), streams=2 if two else 1,
                                       dim_out=160 if kind == "vomix2out" else 80)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    return sd


def test_param_counts_match_reference_probe():
    n = lambda s: sum(int(np.prod(v)) for v in s.values())
    assert n(syn.acoustic_param_shapes()) == 250_521_248                         # VoMix  (SURVEY section 8c probe)
    assert n(syn.acoustic_param_shapes(dim_cond=80, streams=1)) == 249_390_672   # VoSingle
    hs = syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG)
    assert len(hs) == 234 and n(hs) == 12_037_340


def test_recipe_is_deterministic():
    a = syn.synth_array("transformer.layers.3.2.to_qkv.weight", (8, 4), seed=0)
    b = syn.synth_array("transformer.layers.3.2.to_qkv.weight", (8, 4), seed=0)
    assert np.array_equal(a, b)
    assert abs(float(a[0, 0]) - float(syn.synth_array("transformer.layers.3.2.to_qkv.weight", (8, 4), seed=1)[0, 0])) > 0


@pytest.mark.parametrize("name,kind", [("vomix_small", "vomix"), ("vomix2out_small", "vomix2out")])
def test_oracle_small_vs_reference_golden(name, kind):
    """vomix2out = twocondition_twooutput (acoustic.py:375-376): 160-wide state and output."""
    sd = _state(kind, dim=128, dim_emb=64, depth=4, heads=2)
    g = np.load(os.path.join(GOLDEN, f"acoustic_{name}.npz"))
    ids, cond, y0 = (torch.from_numpy(g[k]) for k in ("phoneme_ids", "cond", "y0"))
    t = torch.tensor(float(g["times"]))
    assert rel_l2(orc.acoustic_forward(sd, y0, t, ids, cond, False), torch.from_numpy(g["fwd_cond"])) < 1e-5
    assert rel_l2(orc.acoustic_forward(sd, y0, t, ids, cond, True), torch.from_numpy(g["fwd_null"])) < 1e-5
    assert rel_l2(orc.forward_with_cond_scale(sd, y0, t, ids, cond, 0.7), torch.from_numpy(g["cfg07"])) < 1e-5
    assert rel_l2(orc.forward_with_cond_scale(sd, y0, t, ids, cond, 1.0), torch.from_numpy(g["cfg10"])) < 1e-5
    roll = orc.sample(sd, ids[:1], cond[:1], y0[:1], 0.7, nfe=int(g["rollout_nfe"]))
    assert rel_l2(roll, torch.from_numpy(g["rollout"])) < 1e-5


@pytest.mark.parametrize("name,kind", [("vomix_full", "vomix"), ("vosingle_full", "vosingle")])
def test_oracle_full_width_vs_reference_golden(name, kind):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    sd = _state(kind)
    g = np.load(os.path.join(GOLDEN, f"acoustic_{name}.npz"))
    ids, cond, y0 = (torch.from_numpy(g[k]) for k in ("phoneme_ids", "cond", "y0"))
    t = torch.tensor(float(g["times"]))
    assert rel_l2(orc.forward_with_cond_scale(sd, y0, t, ids, cond, 0.7), torch.from_numpy(g["cfg07"])) < 1e-5
    assert rel_l2(orc.acoustic_forward(sd, y0, t, ids, cond, True), torch.from_numpy(g["fwd_null"])) < 1e-5


def test_midpoint_integrator_analytic():
    """dy/dt = -2y + t: midpoint is 2nd order; check the value and the convergence order."""
    exact = lambda t: (1.0 + 0.25) * np.exp(-2 * t) + t / 2 - 0.25
    f = lambda t, y: -2 * y + t
    errs = []
    for n in (16, 32):
        y = orc.odeint_fixed(f, torch.tensor([1.0], dtype=torch.float64), orc.fixed_grid(1.0 / n).double(), "midpoint")
        errs.append(abs(float(y) - exact(1.0)))
    assert errs[0] < 2e-3 and 3.5 < errs[0] / errs[1] < 4.5
    g = orc.fixed_grid(0.0625)
    assert g.numel() == 17 and float(g[-1]) == 1.0 and float(g[1]) == 0.0625
    ye = orc.odeint_fixed(f, torch.tensor([1.0], dtype=torch.float64), orc.fixed_grid(1 / 64).double(), "euler")
    assert abs(float(ye) - exact(1.0)) < 2e-2


@pytest.mark.parametrize("tag,c0", [("covomix", 500), ("small64", 64)])
def test_oracle_hifigan_vs_reference_golden(tag, c0):
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    h["upsample_initial_channel"] = c0
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    g = np.load(os.path.join(GOLDEN, f"hifigan_{tag}.npz"))
    folded = orc.fold_weight_norm(sd)
    assert rel_l2(folded["conv_post.weight"], torch.from_numpy(g["conv_post_weight"])) < 1e-6
    assert rel_l2(folded["ups.0.weight"][0], torch.from_numpy(g["ups0_weight_row0"])) < 1e-6
    mel = torch.from_numpy(g["mel"])
    yb = orc.hifigan_forward(folded, h, mel)
    assert yb.shape == g["wav_batched"].shape == (mel.shape[0], 1, 160 * mel.shape[2] + 32)
    assert rel_l2(yb, torch.from_numpy(g["wav_batched"])) < 1e-5
    yu = orc.hifigan_forward(folded, h, mel[0])
    assert yu.shape == g["wav_unbatched"].shape
    pcm = orc.wav_to_int16(yu)
    assert np.abs(pcm.astype(np.int32) - g["int16_unbatched"].astype(np.int32)).max() <= 1


def test_assembly_bit_exact_vs_golden():
    from covomix_amd import assembly as asm
    g = np.load(os.path.join(GOLDEN, "assembly.npz"))
    for i in range(int(g["n_cases"])):
        t = lambda k: torch.from_numpy(g[f"c{i}_{k}"])
        for fn in (orc.assemble_dialogue, asm.build_dialogue_inputs):
            ids, mel, mask = fn(t("sem_a"), t("sem_b"), t("pred_a"), t("pred_b"), t("mel_a"), t("mel_b"))
            assert torch.equal(ids, t("ids")) and torch.equal(mel, t("mel")) and torch.equal(mask, t("mask"))
            assert ids.dtype == torch.int64 and mask.dtype == torch.bool
        for fn in (orc.assemble_monologue, asm.build_monologue_inputs):
            ids, mel, mask = fn(t("sem_a"), t("pred_a"), t("mel_a"))
            assert torch.equal(ids, t("mono_ids")) and torch.equal(mel, t("mono_mel")) and torch.equal(mask, t("mono_mask"))
        for fn in (orc.select_generated, asm.select_generated_frames):
            assert torch.equal(fn(t("sampled"), t("mask")), t("valid"))
