"""Model-level parity on the GPU: the HIP path against (i) the committed golden vectors that the
REFERENCE modules produced (tests/golden/*.npz, see make_golden.py) and (ii) the CPU oracle on
the same seeded inputs.  Tolerance for the fp32-class paths (f16x3 default and fp32): 1e-5 rel-L2 per evaluation /
2e-5 after a 32-NFE rollout - 10x above what is measured (5e-7 ... 1.9e-6; the fp32 floor the survey measured is
2-7e-7), 50x below the north_star budget of 1e-3, so a regression to 5e-5 fails here.  Only the opt-in f16 mode is
held to the 1e-3 budget."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

EVAL_TOL = 1e-5
ROLL_TOL = 2e-5


def _state(kind, **kw):
    import covomix_amd.synthetic as syn
    two = kind in ("vomix", "vomix2out")
    shapes = syn.acoustic_param_shapes(dim=kw.get("dim", 1024), dim_cond=160 if two else 80,
                                       dim_emb=kw.get("dim_emb", 1024), depth=kw.get("depth", 8),
                                       heads=kw.get("heads", 16), streams=2 if two else 1,
                                       dim_out=160 if kind == "vomix2out" else 80)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    return sd


CASES = {
    "vomix_full": ("vomix", {}),
    "vosingle_full": ("vosingle", {}),
    "vomix_small": ("vomix", dict(dim=128, dim_emb=64, depth=4, heads=2)),
    "vomix2out_small": ("vomix2out", dict(dim=128, dim_emb=64, depth=4, heads=2)),    # twocondition_twooutput (row N2)
}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    from covomix_amd.conditional_model import CoVoMixModel
    kind, kw = CASES[request.param]
    sd = _state(kind, **kw)
    g = np.load(os.path.join(GOLDEN, f"acoustic_{request.param}.npz"))
    model = CoVoMixModel.from_state_dict(sd).eval().to("cuda:0")
    return request.param, model, sd, g


def _single_eval(model, g, cond_scale):
    """One CFG evaluation through the product path: an Euler step of size 1 from y0 at t = times
    is not expressible, so drive VectorField directly (same code path the sampler uses)."""
    from covomix_amd import ops
    f = model._get_field()
    dev = f.device
    ids = torch.from_numpy(g["phoneme_ids"]).to(dev)
    cond = torch.from_numpy(g["cond"]).to(dev)
    y0 = torch.from_numpy(g["y0"]).to(dev)
    t = torch.tensor([float(g["times"])], dtype=torch.float32, device=dev)
    use_null = cond_scale != 1.0
    ctx = f.prepare(ids, cond, t, use_null)
    M1 = ctx["M1"]
    ctx["ws"]["xin"][:M1].copy_(y0.reshape(M1, -1))
    if use_null:
        ctx["ws"]["xin"][M1:].copy_(y0.reshape(M1, -1))
    pred = f.evaluate(ctx, 0)
    return pred[:M1].reshape(y0.shape).clone(), (pred[M1:].reshape(y0.shape).clone() if use_null else None), y0


def test_forward_branches_vs_reference_golden(case):
    name, model, sd, g = case
    fc, fn, _ = _single_eval(model, g, 0.7)
    e_c, e_n = rel_l2(fc, torch.from_numpy(g["fwd_cond"])), rel_l2(fn, torch.from_numpy(g["fwd_null"]))
    print(name, "cond", e_c, "null", e_n)
    assert e_c < EVAL_TOL and e_n < EVAL_TOL


def test_cfg_combine_vs_reference_golden(case):
    from covomix_amd import ops
    name, model, sd, g = case
    fc, fn, y0 = _single_eval(model, g, 0.7)
    out = torch.empty_like(y0)
    zero = torch.zeros_like(y0)
    ops.cfg_combine_axpy(fc.contiguous(), fn.contiguous(), zero, 0.7, 1.0, out)
    assert rel_l2(out, torch.from_numpy(g["cfg07"])) < EVAL_TOL
    fc1, none, _ = _single_eval(model, g, 1.0)          # s == 1.0 skips the null branch
    assert none is None
    assert rel_l2(fc1, torch.from_numpy(g["cfg10"])) < EVAL_TOL


def test_rollout_vs_reference_golden(case):
    name, model, sd, g = case
    nfe = int(g["rollout_nfe"])
    model.nfe = nfe
    ids = torch.from_numpy(g["phoneme_ids"][:1]).cuda()
    cond = torch.from_numpy(g["cond"][:1]).cuda()
    mask = torch.from_numpy(g["mask"][:1]).cuda()
    y0 = torch.from_numpy(g["y0"][:1])
    out = model.synthesis_sample(phoneme_ids=ids, cond=cond, mask=mask, cond_scale=0.7, y0=y0)
    assert out.shape == (1, ids.shape[1], g["y0"].shape[-1]) and out.is_cuda
    e = rel_l2(out, torch.from_numpy(g["rollout"]))
    print(name, "rollout", nfe, e)
    assert e < ROLL_TOL
    # inputs are not mutated and the call is deterministic
    assert torch.equal(y0, torch.from_numpy(g["y0"][:1]))
    out2 = model.synthesis_sample(phoneme_ids=ids, cond=cond, mask=mask, cond_scale=0.7, y0=y0)
    assert torch.equal(out, out2)


def test_batched_equals_single_and_oracle():
    """Equal-length batching must not change per-utterance results (no key-padding mask exists,
    acoustic.py:313) and must match the CPU oracle on fresh seeded inputs (ragged T, B=3)."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _state("vomix", dim=128, dim_emb=64, depth=4, heads=2)
    model = CoVoMixModel.from_state_dict(sd, nfe=8).eval().to("cuda:0")
    inp = syn.synthetic_inputs("vomix", 3, 131, 40, seed=99)
    out = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), 0.7, y0=inp["y0"])
    one = model.synthesis_sample(inp["phoneme_ids"][1:2].cuda(), inp["cond"][1:2].cuda(), None, 0.7, y0=inp["y0"][1:2])
    assert rel_l2(out[1:2], one) < 1e-5
    ref = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=8)
    assert rel_l2(out, ref) < ROLL_TOL
    # euler extra (not in the reference): against the oracle only
    model.ode_method, model.nfe = "euler", 6
    oe = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), None, 0.7, y0=inp["y0"])
    re = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=6, method="euler")
    assert rel_l2(oe, re) < ROLL_TOL


@pytest.mark.parametrize("tag,c0", [("covomix", 500), ("small64", 64)])
def test_hifigan_vs_reference_golden(tag, c0):
    import covomix_amd.synthetic as syn
    from covomix_amd.vocoder import AttrDict, Generator, mel_decode_to_wav
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    h["upsample_initial_channel"] = c0
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    g = np.load(os.path.join(GOLDEN, f"hifigan_{tag}.npz"))
    gen = Generator(AttrDict(h)).to("cuda:0")
    gen.load_state_dict(sd)
    gen.eval()
    gen.remove_weight_norm()
    mel = torch.from_numpy(g["mel"])
    yb = gen(mel.cuda())
    assert yb.shape == g["wav_batched"].shape
    e_b = rel_l2(yb, torch.from_numpy(g["wav_batched"]))
    yu = gen(mel[0].cuda())                               # unbatched [80,T] -> [1,L]
    assert yu.shape == g["wav_unbatched"].shape
    e_u = rel_l2(yu, torch.from_numpy(g["wav_unbatched"]))
    print(tag, "batched", e_b, "unbatched", e_u)
    assert e_b < 1e-5 and e_u < 1e-5
    pcm = mel_decode_to_wav(gen, mel[0].cuda())
    assert pcm.dtype == np.int16 and pcm.shape == g["int16_unbatched"].shape
    assert np.abs(pcm.astype(np.int32) - g["int16_unbatched"].astype(np.int32)).max() <= 1


def test_full_size_properties():
    """BASELINE-size (B=8, T=1000) checks through size-independent properties: batch-permutation
    equivariance, zero CFG sensitivity at s=1 to the null inputs, and finite outputs."""
    from covomix_amd.conditional_model import CoVoMixModel
    import covomix_amd.synthetic as syn
    sd = _state("vomix")
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    inp = syn.synthetic_inputs("vomix", 8, 1000, 400, seed=3)
    args = (inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda())
    out = model.synthesis_sample(*args, 0.7, y0=inp["y0"])
    assert out.shape == (8, 1000, 80) and torch.isfinite(out).all()
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    outp = model.synthesis_sample(args[0][perm.cuda()], args[1][perm.cuda()], args[2][perm.cuda()], 0.7,
                                  y0=inp["y0"][perm])
    assert rel_l2(outp, out[perm.cuda()]) < 1e-5


# ---------------------------------------------------------------- opt-in precision="f16" (single-term fp16 operands)
F16_ROLL_TOL = 1e-3          # BASELINE.json north_star budget: <= 1e-3 rel-L2 on the final mel


@pytest.mark.parametrize("name", ["vomix_full", "vosingle_full", "vomix_small"])
def test_f16_precision_rollout_within_north_star_budget(name):
    """precision='f16' is NOT fp32-class (the default f16x3 is); it must stay inside the 1e-3 rel-L2 budget the
    north star states, measured against the REFERENCE golden rollout."""
    from covomix_amd.conditional_model import CoVoMixModel
    kind, kw = CASES[name]
    g = np.load(os.path.join(GOLDEN, f"acoustic_{name}.npz"))
    model = CoVoMixModel.from_state_dict(_state(kind, **kw), precision="f16").eval().to("cuda:0")
    model.nfe = int(g["rollout_nfe"])
    ids = torch.from_numpy(g["phoneme_ids"][:1]).cuda()
    cond = torch.from_numpy(g["cond"][:1]).cuda()
    mask = torch.from_numpy(g["mask"][:1]).cuda()
    out = model.synthesis_sample(phoneme_ids=ids, cond=cond, mask=mask, cond_scale=0.7, y0=torch.from_numpy(g["y0"][:1]))
    e = rel_l2(out, torch.from_numpy(g["rollout"]))
    print(name, "f16 rollout", int(g["rollout_nfe"]), e)
    assert e < F16_ROLL_TOL


# ---------------------------------------------------------------- edge shapes (ragged tiles, tiny and long sequences)
@pytest.mark.parametrize("precision", ["f16x3", "fp32", "f16"])
@pytest.mark.parametrize("kind,B,T,prompt", [("vomix", 1, 1, 0), ("vomix", 2, 3, 1), ("vosingle", 1, 33, 10),
                                             ("vomix", 1, 130, 130), ("vosingle", 2, 257, 0)])
def test_edge_shapes_vs_oracle(kind, B, T, prompt, precision):
    """T = 1 (one key), T % 4 != 0 (the transposed-V epilogue is not usable: fp32-attention fallback), T % 32 != 0
    (masked last key tile), prompt = T (nothing to generate) and prompt = 0, at B = 1 and 2 - against the CPU oracle."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _state(kind, dim=128, dim_emb=64, depth=4, heads=2)
    model = CoVoMixModel.from_state_dict(sd, nfe=4, precision=precision).eval().to("cuda:0")
    inp = syn.synthetic_inputs(kind, B, T, prompt, seed=5)
    for s in (0.7, 1.0):                                       # s == 1.0 skips the null branch (acoustic.py:421-423)
        out = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), s, y0=inp["y0"])
        ref = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], s, nfe=4)
        assert out.shape == ref.shape == (B, T, 80)
        e = rel_l2(out, ref)
        assert e < (F16_ROLL_TOL if precision == "f16" else ROLL_TOL), (kind, B, T, prompt, precision, s, e)


def test_long_sequence_and_argument_errors():
    """T = 2500 (79 key tiles, 10 query blocks per head) against the oracle on a narrow model, and the reference's
    own argument checks (acoustic.py:441-443 cond width; midpoint needs an even NFE)."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _state("vosingle", dim=128, dim_emb=64, depth=4, heads=2)
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    inp = syn.synthetic_inputs("vosingle", 1, 2500, 700, seed=6)
    out = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), None, 0.7, y0=inp["y0"])
    torch.set_num_threads(8)
    ref = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=2)
    assert rel_l2(out, ref) < ROLL_TOL
    with pytest.raises((AssertionError, ValueError)):
        model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"][..., :40].cuda(), None, 0.7)
    model.nfe = 3
    with pytest.raises((AssertionError, ValueError)):
        model.synthesis_sample(inp["phoneme_ids"][:, :8].cuda(), inp["cond"][:, :8].cuda(), None, 0.7)


def test_vocoder_full_size_properties_and_precisions():
    """config_covomix generator at BASELINE size (T = 1000 frames): the split-precision path against the all-fp32 path
    (same kernels as round 0, pinned by the goldens at T = 50), batched == unbatched, batch-permutation equivariance."""
    import covomix_amd.synthetic as syn
    from covomix_amd.vocoder import AttrDict, Generator
    h = AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    gens = {}
    for prec in ("f16x3", "fp32"):
        g = Generator(h, precision=prec).to("cuda:0")
        g.load_state_dict(vsd); g.eval(); g.remove_weight_norm()
        gens[prec] = g
    mel = (torch.randn(3, 80, 1000, generator=torch.Generator().manual_seed(5)) * 2 - 6).clamp(-11.52, 2.0).cuda()
    y16, y32 = gens["f16x3"](mel), gens["fp32"](mel)
    assert y16.shape == (3, 1, 160032) and torch.isfinite(y16).all()
    assert rel_l2(y16, y32) < 3e-6
    assert rel_l2(gens["f16x3"](mel[1]), y16[1]) < 1e-6
    perm = torch.tensor([2, 0, 1]).cuda()
    assert rel_l2(gens["f16x3"](mel[perm]), y16[perm]) < 1e-6


@pytest.mark.parametrize("B,T", [(1, 1000), (3, 333)])
def test_vocoder_grouped_resblock_stages_are_bit_identical_to_block_after_block(B, T):
    """Round 6: on the wide stages (250 / 125 channels) the three ResBlocks of a stage (kernel sizes 3 / 7 / 11, models.py:104-110) share
    launches - their convolutions are independent until the xs accumulate (cvx_hifigan_resblock_stage_f16x3 / cvx_hifigan_conv1d_group_f16x3:
    18 -> 8 launches per stage).  Same arithmetic per output element, same accumulation order into xs: the waveform must equal, bit for
    bit, the one the block-after-block schedule gives (and both the ragged and the unragged call)."""
    import covomix_amd.synthetic as syn
    from covomix_amd import vocoder
    from covomix_amd.vocoder import AttrDict, Generator
    h = AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    gen = Generator(h).to("cuda:0")
    gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(9)) * 2 - 6).clamp(-11.52, 2.0).cuda()
    lens = [T - 37 * i for i in range(B)]
    assert vocoder.GROUP_STAGE
    grouped = gen(mel).clone()
    grouped_r = gen(mel, lengths=lens).clone() if B > 1 else None
    vocoder.GROUP_STAGE = False
    try:
        plain = gen(mel).clone()
        plain_r = gen(mel, lengths=lens).clone() if B > 1 else None
    finally:
        vocoder.GROUP_STAGE = True
    assert torch.equal(grouped, plain)
    if B > 1:
        assert torch.equal(grouped_r, plain_r)


def test_vocoder_repeated_calls_are_bit_identical_and_independent_of_the_previous_input():
    """Round 4: cvx_amax_pow2_scale_f32 left its maximum in the scratch word that the NEXT call's upsampler max-accumulates
    into, so a call's activation pre-scale depended on the previous call's input (first call of a shape != later calls by an
    fp32 rounding).  Now: the same mel gives the same bits on every call, whatever ran in between."""
    import covomix_amd.synthetic as syn
    from covomix_amd.vocoder import AttrDict, Generator
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    gen = Generator(AttrDict(h)).to("cuda:0")
    gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
    g = torch.Generator().manual_seed(2)
    mel = (torch.randn(2, 80, 120, generator=g) * 2 - 6).clamp(-11.52, 2.0).cuda()
    loud = (torch.randn(2, 80, 120, generator=g) * 0.5 + 1.5).clamp(-11.52, 2.0).cuda()     # a much "louder" input in between
    first = gen(mel).clone()
    second = gen(mel).clone()
    gen(loud)
    third = gen(mel).clone()
    assert torch.equal(first, second) and torch.equal(first, third)


def test_time_tables_are_cached_per_model_and_grid_not_across_checkpoints(monkeypatch):
    """Everything that depends on the evaluation times alone (time MLP, adaLN table, activation pre-scales, deferred-norm weight
    tables - reference acoustic.py:198-204: gamma, beta are functions of the time embedding) is computed once per (VectorField, solver
    grid) and reused: a second call launches no table kernels and returns the same bits; another grid gets its own entry; OTHER
    WEIGHTS (a new checkpoint in the same model object: EMA on / off) never see the old tables; at most TIME_CACHE grids stay resident."""
    from covomix_amd.conditional_model import CoVoMixModel
    import covomix_amd.acoustic as ac
    import covomix_amd.ops as ops
    import covomix_amd.synthetic as syn
    monkeypatch.setattr(ac.VectorField, "DEFER_MIN_ROWS", 2048)          # the deferred-norm tables too (2 x 2 x 600 = 2400 rows)
    monkeypatch.setattr(ac.VectorField, "DEFER_RULE", False)
    monkeypatch.setenv("CVX_GRAPH", "0")
    sd = _state("vomix")
    model = CoVoMixModel.from_state_dict(sd, nfe=4).eval().to("cuda:0")
    inp = syn.synthetic_inputs("vomix", 2, 600, 200, seed=5)
    args = (inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda())
    calls = {"skinny": 0, "colscale": 0}
    real_sk, real_cs = ops.gemm_skinny, ops.split_f16_colscale_il

    def sk(*a, **k):
        calls["skinny"] += 1
        return real_sk(*a, **k)

    def cs(*a, **k):
        calls["colscale"] += 1
        return real_cs(*a, **k)
    monkeypatch.setattr(ops, "gemm_skinny", sk)
    monkeypatch.setattr(ops, "split_f16_colscale_il", cs)
    one = model.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()
    f = model._get_field()
    assert list(f._time_cache) == [(4, "midpoint")] and f._time_cache[(4, "midpoint")]["dn"] is not None
    first = dict(calls)
    assert first["skinny"] > 0 and first["colscale"] == 15
    two = model.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()
    assert calls == first and torch.equal(one, two)                          # nothing rebuilt, same bits
    model.nfe = 2
    model.synthesis_sample(*args, 0.7, y0=inp["y0"])
    assert list(f._time_cache) == [(4, "midpoint"), (2, "midpoint")] and calls["colscale"] == 30
    model.nfe = 6
    model.synthesis_sample(*args, 0.7, y0=inp["y0"])                         # a third grid evicts the least recently used one
    assert list(f._time_cache) == [(2, "midpoint"), (6, "midpoint")]
    assert all(k[2] in f._time_cache for k in f._dn_bufs)
    model.nfe = 4
    again = model.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()         # rebuilt after the eviction: same bits
    assert torch.equal(one, again)
    # other weights: a fresh field, fresh tables - the result is the other checkpoint's, bit for bit what a new model object gives
    sd2 = {k: (v * 1.25 if "to_gamma" in k or "to_beta" in k else v) for k, v in sd.items()}
    other = CoVoMixModel.from_state_dict(sd2, nfe=4).eval().to("cuda:0")
    want = other.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()
    assert not torch.equal(want, one)
    assert other._get_field() is not f and list(other._get_field()._time_cache) == [(4, "midpoint")]
    # ... and inside ONE model object: switching the weights (EMA off -> the raw state dict) drops the field with its tables
    from covomix_amd.conditional_model import parameter_order
    both = CoVoMixModel(sd, ema_shadow=[sd2[n] for n in parameter_order(sd.keys())]).eval().to("cuda:0")
    both.nfe = 4
    ema_out = both.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()
    f_ema = both._get_field()
    both.eval(no_ema=True)
    raw_out = both.synthesis_sample(*args, 0.7, y0=inp["y0"]).clone()
    assert both._get_field() is not f_ema
    assert torch.equal(ema_out, want) and torch.equal(raw_out, one)
