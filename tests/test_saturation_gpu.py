"""Parity robustness of the split-precision path (round-2 verdict, "What's weak" #2 / next-round item 3a).

Split activations are stored as (fp16 hi, fp16 lo) pairs times a power-of-two pre-scale from a weights-only gain model
(transformer) or from the measured stage input (vocoder); every such store clamps to +-65504.  Before this round a
checkpoint or an input outside the model's window gave a wrong result with rc 0.  Now every saturating store raises the
device's sticky flag (cvx_saturation_flag_*), the host reads it once per call and re-runs a flagged call on the exact-fp32
kernels (or raises, CVX_ON_SATURATION=raise).  Checked here:
  * kernel level: each family of split stores raises the flag when (and only when) it clamps;
  * outlier checkpoints - ONE to_embed row / FF channel / to_qkv row / HiFi-GAN conv_pre channel times 2^14 - stay
    <= 1e-5 from the CPU oracle (flagged or not: never a silent 1e-2);
  * inputs / weights that do leave the window: warning + fp32 re-run <= 1e-5, or a loud error under CVX_ON_SATURATION=raise.
"""
import math
import warnings

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import covomix_amd.ops as o
    return o


def dev():
    return torch.device("cuda:0")


def test_flag_is_raised_by_every_family_of_split_stores(ops):
    dev_ = dev()
    g = torch.Generator().manual_seed(0)
    one = lambda v: torch.tensor([v], dtype=torch.float32, device=dev_)

    def flagged(fn):
        ops.saturation_reset()
        fn()
        return ops.saturation_query()
    x = torch.randn(300, 256, generator=g).to(dev_)
    # cvx_split_f16_dev
    assert flagged(lambda: ops.split_act_f16(x)) == 0
    assert flagged(lambda: ops.split_act_f16(x, scale=one(2.0 ** 15))) != 0
    big = x.clone(); big[7, 3] = 1e5
    assert flagged(lambda: ops.split_act_f16(big)) != 0
    # sticky: stays up across later clean launches until it is reset; query(reset=True) clears it
    ops.saturation_reset(); ops.split_act_f16(big); ops.split_act_f16(x)
    assert ops.saturation_query(reset=False) != 0 and ops.saturation_query() != 0 and ops.saturation_query() == 0
    # AdaRMSNorm split output
    gam, bet = torch.ones(256, device=dev_), torch.zeros(256, device=dev_)
    pair = (torch.empty(300, 256, dtype=torch.float16, device=dev_), torch.empty(300, 256, dtype=torch.float16, device=dev_))
    assert flagged(lambda: ops.adarmsnorm(x, gam, bet, None, out_split=pair, split_scale=one(16.0))) == 0
    assert flagged(lambda: ops.adarmsnorm(x, gam, bet, None, out_split=pair, split_scale=one(2.0 ** 15))) != 0
    # GEMM epilogues (small-problem kernel and large-problem kernel): split output and transposed V^T
    for M in (300, 2304):
        a = torch.randn(M, 256, generator=g).to(dev_)
        w = (torch.randn(512, 256, generator=g) / 16).to(dev_)
        ws = ops.split_f16(w)
        kw = dict(w_split=ws, a_split=ops.split_act_f16(a))
        if M >= 2048:
            il = ops.SplitIL(M, 256, dev_); ops.split_act_f16(a, il)
            kw = dict(w_split=ws, w_il=ops.split_f16_interleaved(ws), a_split=il)
        o = (torch.empty(M, 512, dtype=torch.float16, device=dev_), torch.empty(M, 512, dtype=torch.float16, device=dev_))
        c = torch.empty(M, 512, device=dev_)
        assert flagged(lambda: ops.gemm(a, w, c, out_split=o, **kw)) == 0
        assert flagged(lambda: ops.gemm(a, w, c, out_split=o, c_scale=one(2.0 ** 17), **kw)) != 0
        assert torch.isfinite(o[0].float()).all() and float(o[0].float().abs().max()) == 65504.0      # clamped, not inf
    # attention split output
    Bt, T, H = 1, 64, 1
    qkv = (torch.randn(T, 192, generator=g) * 0.5).to(dev_)
    ah, al = ops.split_act_f16(qkv.contiguous())
    qk = (ah[:, :128].contiguous(), al[:, :128].contiguous())
    vt = (torch.zeros(64, 64, dtype=torch.float16, device=dev_), torch.zeros(64, 64, dtype=torch.float16, device=dev_))
    slots = ops.vt_frame_slots(T, dev_)
    vt[0][:, slots] = ah[:, 128:].T; vt[1][:, slots] = al[:, 128:].T
    op = (torch.empty(T, 64, dtype=torch.float16, device=dev_), torch.empty(T, 64, dtype=torch.float16, device=dev_))
    assert flagged(lambda: ops.attention_f16x3(qk, vt, None, Bt, T, H, 0.125, out_split=op)) == 0
    assert flagged(lambda: ops.attention_f16x3(qk, vt, None, Bt, T, H, 0.125, out_split=op, out_scale=one(2.0 ** 26))) != 0


def test_nan_in_an_fp32_operand_raises_the_flag(ops):
    """The clamps turn NaN into -65504 and v_max3 / fmaxf skip NaN operands (round-3 advice, cvx_common.h): without its own
    predicate a NaN residual / bias / input would be stored as a finite pair with the flag down.  Every family of split stores
    that takes an fp32 operand: one NaN in it -> flag up (and a clean repeat -> flag down)."""
    dev_ = dev()
    g = torch.Generator().manual_seed(1)
    nan = float("nan")

    def flagged(fn):
        ops.saturation_reset()
        fn()
        return ops.saturation_query()
    x = torch.randn(300, 256, generator=g).to(dev_)
    bad = x.clone(); bad[11, 5] = nan
    assert flagged(lambda: ops.split_act_f16(x)) == 0 and flagged(lambda: ops.split_act_f16(bad)) != 0
    il = ops.SplitIL(300, 256, dev_)
    assert flagged(lambda: ops.split_act_f16(x, il)) == 0 and flagged(lambda: ops.split_act_f16(bad, il)) != 0
    gam, bet = torch.ones(256, device=dev_), torch.zeros(256, device=dev_)
    pair = (torch.empty(300, 256, dtype=torch.float16, device=dev_), torch.empty(300, 256, dtype=torch.float16, device=dev_))
    assert flagged(lambda: ops.adarmsnorm(x, gam, bet, None, out_split=pair)) == 0
    assert flagged(lambda: ops.adarmsnorm(bad, gam, bet, None, out_split=pair)) != 0
    badg = gam.clone(); badg[200] = nan
    assert flagged(lambda: ops.adarmsnorm(x, badg, bet, None, out_split=pair)) != 0
    # GEMM epilogues: generic (300 rows, flag 16), medium (300 rows interleaved) and large (2304 rows): NaN bias / NaN residual
    for M, flags in ((300, 16), (300, 0), (2304, 16)):
        a = torch.randn(M, 256, generator=g).to(dev_)
        w = (torch.randn(512, 256, generator=g) / 16).to(dev_)
        ws = ops.split_f16(w)
        kw = dict(w_split=ws, a_split=ops.split_act_f16(a))
        if M >= 2048 or flags == 0:
            ail = ops.SplitIL(M, 256, dev_); ops.split_act_f16(a, ail)
            kw = dict(w_split=ws, w_il=ops.split_f16_interleaved(ws), a_split=ail)
        o = ops.SplitIL(M, 512, dev_) if "w_il" in kw else (torch.empty(M, 512, dtype=torch.float16, device=dev_), torch.empty(M, 512, dtype=torch.float16, device=dev_))
        c = torch.empty(M, 512, device=dev_)
        bias = torch.randn(512, generator=g).to(dev_)
        res = torch.randn(M, 512, generator=g).to(dev_)
        bbad = bias.clone(); bbad[300] = nan
        rbad = res.clone(); rbad[M - 1, 17] = nan
        old = ops._GEMM_FLAGS
        ops._GEMM_FLAGS = old | flags
        try:
            assert flagged(lambda: ops.gemm(a, w, c, bias=bias, residual=res, out_split=o, **kw)) == 0, (M, flags)
            assert flagged(lambda: ops.gemm(a, w, c, bias=bbad, residual=res, out_split=o, **kw)) != 0, (M, flags)
            assert flagged(lambda: ops.gemm(a, w, c, bias=bias, residual=rbad, out_split=o, **kw)) != 0, (M, flags)
            assert flagged(lambda: ops.gemm(a, w, c, bias=bbad, act=ops.ACT_GELU, out_split=o, write_f32=False, **kw)) != 0, (M, flags)
        finally:
            ops._GEMM_FLAGS = old


def _full_width_state(kind="vomix"):
    import covomix_amd.synthetic as syn
    two = kind == "vomix"
    shapes = syn.acoustic_param_shapes(dim_cond=160 if two else 80, streams=2 if two else 1)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    return sd


OUTLIERS = {
    "to_embed_row": ("to_embed.weight", 77),
    "ff1_channel": ("transformer.layers.3.4.0.weight", 1234),
    "to_qkv_v_row": ("transformer.layers.5.2.to_qkv.weight", 2 * 1024 + 99),
    "to_out_row": ("transformer.layers.2.2.to_out.weight", 500),
}


@pytest.mark.parametrize("which", list(OUTLIERS))
def test_single_outlier_row_times_2_14_is_never_silently_wrong(which):
    """Full-width VoMix, one weight ROW (= one output channel) times 2^14: the gain model only sees the Frobenius norm move.
    The result must be fp32-class against the CPU oracle - by staying inside the window or by the flagged fp32 re-run."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _full_width_state()
    key, row = OUTLIERS[which]
    sd[key] = sd[key].clone()
    sd[key][row] *= 2.0 ** 14
    inp = syn.synthetic_inputs("vomix", 2, 96, 40, seed=21)
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), 0.7, y0=inp["y0"])
    rerun = any("saturat" in str(w.message) for w in rec)
    ref = orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=2)
    e = rel_l2(out, ref)
    print(f"outlier {which}: rel-L2 vs oracle {e:.3e} ({'flagged -> fp32 re-run' if rerun else 'inside the window'})")
    assert torch.isfinite(out).all() and e < 1e-5


def test_input_outside_the_window_is_rerun_in_fp32_or_raises(monkeypatch):
    """The gain model takes the prompt mel at RMS 4; a cond tensor 4000x that leaves the residual stream's window.  Default:
    warning + the exact-fp32 result; CVX_ON_SATURATION=raise: CovomixHipError; CVX_SAT_CHECK=0 shows what the unguarded path
    would have returned."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd._lib import CovomixHipError
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _full_width_state()
    inp = syn.synthetic_inputs("vomix", 1, 80, 40, seed=5)
    cond = inp["cond"] * 4000.0
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    args = (inp["phoneme_ids"].cuda(), cond.cuda(), inp["mask"].cuda(), 0.7)
    ref = orc.sample(sd, inp["phoneme_ids"], cond, inp["y0"], 0.7, nfe=2)
    with pytest.warns(UserWarning, match="saturat"):
        out = model.synthesis_sample(*args, y0=inp["y0"])
    assert rel_l2(out, ref) < 1e-5
    # ragged call, same mechanism
    with pytest.warns(UserWarning, match="saturat"):
        outs = model.synthesis_sample([inp["phoneme_ids"][0].cuda()], [cond[0].cuda()], None, 0.7, y0=[inp["y0"][0]])
    assert rel_l2(outs[0], ref[0]) < 1e-5
    monkeypatch.setenv("CVX_ON_SATURATION", "raise")
    with pytest.raises(CovomixHipError, match="saturat"):
        model.synthesis_sample(*args, y0=inp["y0"])
    monkeypatch.delenv("CVX_ON_SATURATION")
    monkeypatch.setenv("CVX_SAT_CHECK", "0")
    silent = model.synthesis_sample(*args, y0=inp["y0"])
    print("unguarded split-precision result on the out-of-window input: rel-L2", rel_l2(silent, ref))
    monkeypatch.delenv("CVX_SAT_CHECK")
    # a normal input right afterwards is neither flagged nor re-run
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ok = model.synthesis_sample(inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), 0.7, y0=inp["y0"])
    assert rel_l2(ok, orc.sample(sd, inp["phoneme_ids"], inp["cond"], inp["y0"], 0.7, nfe=2)) < 1e-5


def test_time_tables_built_by_a_flagged_call_are_not_cached():
    """Round-5 advice: the per-(nfe, method) tables (time MLP, adaLN table, activation scales, deferred-norm weights) are built inside
    the first call by split-precision kernels; if THAT call is flagged it alone is re-run in fp32 - a clamped table left in the cache
    would be reused by every later call under a clean flag.  A checkpoint whose time MLP leaves the window (64 evaluation times: the
    table product runs on the split-precision GEMM with split_act_f16(temb)): every call warns and returns the exact-fp32 result, and
    nothing is cached; the healthy checkpoint caches its grid after the first call."""
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _full_width_state()
    inp = syn.synthetic_inputs("vomix", 1, 80, 40, seed=5)
    args = (inp["phoneme_ids"].cuda(), inp["cond"].cuda(), inp["mask"].cuda(), 0.7)
    good = CoVoMixModel.from_state_dict(sd, nfe=64).eval().to("cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        good.synthesis_sample(*args, y0=inp["y0"])
    assert len(good._get_field()._time_cache) == 1
    bad_sd = dict(sd)
    bad_sd["sinu_pos_emb.1.bias"] = sd["sinu_pos_emb.1.bias"] + 3.0e5          # SiLU(3e5) = 3e5 > 65504: split_act_f16(temb) clamps
    bad = CoVoMixModel.from_state_dict(bad_sd, nfe=64).eval().to("cuda:0")
    ref = CoVoMixModel.from_state_dict(bad_sd, nfe=64, precision="fp32").eval().to("cuda:0").synthesis_sample(*args, y0=inp["y0"])
    for _ in range(2):                                                         # the second call must not ride on a cached clamped table
        with pytest.warns(UserWarning, match="saturat"):
            out = bad.synthesis_sample(*args, y0=inp["y0"])
        assert rel_l2(out, ref) < 1e-5
        assert len(bad._get_field()._time_cache) == 0


def test_nan_input_is_not_returned_as_a_finite_result(monkeypatch):
    """One NaN in the prompt mel / in the vocoder's mel: the fp32 reference returns NaN; the split path used to return finite
    numbers with the flag down (the clamps swallow NaN).  Now: flag -> fp32 re-run (NaN, like the reference) or a loud error."""
    import covomix_amd.synthetic as syn
    from covomix_amd._lib import CovomixHipError
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _full_width_state()
    inp = syn.synthetic_inputs("vomix", 1, 80, 40, seed=6)
    cond = inp["cond"].clone(); cond[0, 33, 7] = float("nan")
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    args = (inp["phoneme_ids"].cuda(), cond.cuda(), inp["mask"].cuda(), 0.7)
    with pytest.warns(UserWarning, match="saturat"):
        out = model.synthesis_sample(*args, y0=inp["y0"])
    assert not torch.isfinite(out).all()
    monkeypatch.setenv("CVX_ON_SATURATION", "raise")
    with pytest.raises(CovomixHipError, match="saturat"):
        model.synthesis_sample(*args, y0=inp["y0"])
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    mel = (torch.randn(1, 80, 50, generator=torch.Generator().manual_seed(4)) * 2 - 6).clamp(-11.52, 2.0)
    mel[0, 40, 25] = float("nan")
    with pytest.raises(CovomixHipError, match="saturat"):
        _vocoder(h, vsd)(mel.cuda())


def test_deferred_check_reads_the_flag_once_for_a_block_of_calls():
    """ops.saturation_deferred (round-3 advice: the per-call flag reads are host synchronisations between acoustic model and
    vocoder): inside the block the entry points neither reset nor read; one read at exit covers every call of the block."""
    import covomix_amd.ops as ops
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    sd = _full_width_state()
    inp = syn.synthetic_inputs("vomix", 1, 80, 40, seed=5)
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to("cuda:0")
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    gen = _vocoder(h, vsd)
    ids, mask = inp["phoneme_ids"].cuda(), inp["mask"].cuda()
    good = model.synthesis_sample(ids, inp["cond"].cuda(), mask, 0.7, y0=inp["y0"])
    wav_good = gen(good.transpose(1, 2).contiguous())
    calls = dict(reset=0, query=0)
    r0, q0 = ops.saturation_reset, ops.saturation_query

    def counting(fn, key):
        def f(*a, **k):
            calls[key] += 1
            return fn(*a, **k)
        return f
    ops.saturation_reset, ops.saturation_query = counting(r0, "reset"), counting(q0, "query")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")               # no re-run inside a deferred block
            with ops.saturation_deferred() as guard:
                mel = model.synthesis_sample(ids, inp["cond"].cuda(), mask, 0.7, y0=inp["y0"])
                wav = gen(mel.transpose(1, 2).contiguous())
            assert guard.flagged is False and calls == dict(reset=1, query=1), (guard.flagged, calls)
            print("deferred block vs checked calls: mel rel-L2", rel_l2(mel, good), "wav rel-L2", rel_l2(wav, wav_good))
            assert torch.equal(mel, good), "mel differs"
            assert torch.equal(wav, wav_good), "wav differs"
            with ops.saturation_deferred() as guard:     # the acoustic model saturates, the vocoder call afterwards is clean:
                mel = model.synthesis_sample(ids, (inp["cond"] * 4000.0).cuda(), mask, 0.7, y0=inp["y0"])
                gen(good.transpose(1, 2).contiguous())   # ... the flag is sticky across the block
            assert guard.flagged is True and calls == dict(reset=2, query=2), (guard.flagged, calls)
    finally:
        ops.saturation_reset, ops.saturation_query = r0, q0
    with pytest.warns(UserWarning, match="saturat"):     # outside the block the calls check themselves again
        model.synthesis_sample(ids, (inp["cond"] * 4000.0).cuda(), mask, 0.7, y0=inp["y0"])


def test_snapshot_reads_the_flag_without_waiting_and_h2d_goes_through_pinned_memory():
    """Round 5, host one batch ahead of the device: ops.saturation_snapshot() is an asynchronous copy of the stream's flag into pinned
    memory - two batches enqueued back to back (deferred, no read), each followed by its snapshot and an event: the first snapshot
    says 'saturated', the second 'clean' (the reset of the second block sits between them in stream order), with no flag query in
    between.  ops.h2d copies host tensors through pinned memory (values and dtype conversion intact, device tensors pass through)."""
    import covomix_amd.ops as ops
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    dev = torch.device("cuda:0")
    x = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    d = ops.h2d(x, dev)
    assert d.device.type == "cuda" and d.dtype == torch.int64 and torch.equal(d.cpu(), x)
    d32 = ops.h2d(x[:, ::2], dev, dtype=torch.float32)            # (non-contiguous source, conversion)
    assert d32.dtype == torch.float32 and torch.equal(d32.cpu(), x[:, ::2].float())
    assert ops.h2d(d, dev) is d
    sd = _full_width_state()
    inp = syn.synthetic_inputs("vomix", 1, 80, 40, seed=5)
    model = CoVoMixModel.from_state_dict(sd, nfe=2).eval().to(dev)
    ids, mask = inp["phoneme_ids"].cuda(), inp["mask"].cuda()
    good = model.synthesis_sample(ids, inp["cond"].cuda(), mask, 0.7, y0=inp["y0"])
    q0, queries = ops.saturation_query, [0]

    def counting(*a, **k):
        queries[0] += 1
        return q0(*a, **k)
    ops.saturation_query = counting
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            with ops.saturation_deferred(read=False):
                model.synthesis_sample(ids, (inp["cond"] * 4000.0).cuda(), mask, 0.7, y0=inp["y0"])
            snap_hot = ops.saturation_snapshot()
            ev_hot = torch.cuda.Event(); ev_hot.record()
            with ops.saturation_deferred(read=False):
                mel = model.synthesis_sample(ids, inp["cond"].cuda(), mask, 0.7, y0=inp["y0"])
            snap_ok = ops.saturation_snapshot()
            ev_ok = torch.cuda.Event(); ev_ok.record()
        ev_hot.synchronize()
        assert int(snap_hot[0]) != 0 and snap_hot.is_pinned()
        ev_ok.synchronize()
        assert int(snap_ok[0]) == 0 and queries[0] == 0
        assert torch.equal(mel, good)
    finally:
        ops.saturation_query = q0


def _vocoder(h, vsd, precision=None):
    from covomix_amd.vocoder import AttrDict, Generator
    gen = Generator(AttrDict(h), precision=precision).to("cuda:0")
    gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
    return gen


def test_vocoder_conv_pre_channel_times_2_14():
    """One conv_pre output channel times 2^14.  The problem itself becomes ill-conditioned in fp32 (the CPU oracle sits 7.5e-6
    from an fp64 evaluation, 5e-7 without the outlier), so the yardstick is fp64: the GPU result must be fp32-class, i.e. no
    further from fp64 than twice the fp32 oracle is."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    vsd["conv_pre.weight_g"] = vsd["conv_pre.weight_g"].clone()
    vsd["conv_pre.weight_g"][123] *= 2.0 ** 14                     # weight-norm gain of ONE output channel
    mel = (torch.randn(2, 80, 60, generator=torch.Generator().manual_seed(2)) * 2 - 6).clamp(-11.52, 2.0)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        wav = _vocoder(h, vsd)(mel.cuda())
    folded = orc.fold_weight_norm(vsd)
    ref32 = orc.hifigan_forward(folded, h, mel)
    ref64 = orc.hifigan_forward({k: v.double() for k, v in folded.items()}, h, mel.double())
    e_gpu, e_cpu = rel_l2(wav, ref64), rel_l2(ref32, ref64)
    print(f"vocoder conv_pre channel x 2^14: rel-L2 vs fp64 {e_gpu:.3e} (fp32 CPU oracle: {e_cpu:.3e}; "
          f"re-run: {any('saturat' in str(w.message) for w in rec)})")
    assert torch.isfinite(wav).all() and e_gpu < max(1e-5, 2 * e_cpu)


def test_vocoder_resblock_gain_outside_the_window_is_rerun_or_raises(monkeypatch):
    """First convolutions of one stage-1 ResBlock times 2^10: its intermediates sit 2^10 above the measured stage input,
    outside the 2^6 headroom of that stage's pre-scale -> flag -> all-fp32 re-run (<= 1e-5 vs the oracle) / loud error."""
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    from covomix_amd._lib import CovomixHipError
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    for m in range(3):
        k = f"resblocks.3.convs1.{m}.weight_g"
        vsd[k] = vsd[k] * 2.0 ** 10
    mel = (torch.randn(1, 80, 50, generator=torch.Generator().manual_seed(3)) * 2 - 6).clamp(-11.52, 2.0)
    gen = _vocoder(h, vsd)
    ref = orc.hifigan_forward(orc.fold_weight_norm(vsd), h, mel)
    with pytest.warns(UserWarning, match="saturat"):
        wav = gen(mel.cuda())
    assert rel_l2(wav, ref) < 1e-5
    monkeypatch.setenv("CVX_ON_SATURATION", "raise")
    with pytest.raises(CovomixHipError, match="saturat"):
        gen(mel.cuda())


def test_flags_are_per_stream_and_caller_owned(ops):
    """Round-3 review: the flag was ONE library-owned word per device, so a reset on one stream could clear what another
    stream's kernels had raised.  Now the caller owns one word per (device, stream), carried by the stream's launch context
    (cvx_ctx.sat_flag): a saturating launch on stream A flags A only, a reset / clean launch / query on stream B neither sees nor
    clears it, a stream the Python front end meets for the first time gets its own flag, and a shared flag (capture stream, side
    stream) collects both streams."""
    dev_ = dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 256, generator=g).to(dev_)
    big = x.clone(); big[5, 7] = 1e5
    sa, sb, sc = torch.cuda.Stream(device=dev_), torch.cuda.Stream(device=dev_), torch.cuda.Stream(device=dev_)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ops.saturation_reset()
        ops.split_act_f16(big)                    # raises A's flag
    with torch.cuda.stream(sb):
        ops.saturation_reset()                    # must not clear A's
        ops.split_act_f16(x)
        assert ops.saturation_query() == 0        # B: clean, and B's query does not consume A's
    with torch.cuda.stream(sa):
        assert ops.saturation_query(reset=False) != 0
        assert ops.saturation_flag().data_ptr() != ops.saturation_flag(sb).data_ptr()
    # a stream met for the first time: its own flag (allocated by the front end), nobody else's flag moves
    with torch.cuda.stream(sb):
        ops.saturation_reset()
    with torch.cuda.stream(sc):
        ops.split_act_f16(big)
    sc.synchronize()
    with torch.cuda.stream(sb):
        assert ops.saturation_query() == 0
    # shared flag: kernels on sc report into A's
    with torch.cuda.stream(sa):
        ops.saturation_reset()
    sa.synchronize()
    ops.saturation_share(sa, sc)
    with torch.cuda.stream(sc):
        ops.split_act_f16(big)
    sc.synchronize()
    with torch.cuda.stream(sa):
        assert ops.saturation_query() != 0 and ops.saturation_query() == 0


def test_c_caller_without_a_flag_is_refused_not_silently_unchecked(ops):
    """Round-5 review: a C caller that launched a split-pair kernel on a stream nobody had bound a flag to got SILENT clamping (only
    the Python front end refused).  Version 107: the flag travels in the launch context (cvx_ctx) of every call; a call that writes
    split pairs with a context that carries none returns CVX_EINVAL - unless the context waives the bookkeeping explicitly
    (CVX_CTX_NO_SATURATION_FLAG) - and the library keeps no per-stream state at all (two contexts on one stream with different flags
    report into their own flags)."""
    import ctypes as C
    from covomix_amd import _lib
    lib = _lib.load()
    dev_ = dev()
    x = torch.randn(64, 256, device=dev_)
    x[3, 3] = 1e6
    hi, lo = torch.empty(64, 256, dtype=torch.float16, device=dev_), torch.empty(64, 256, dtype=torch.float16, device=dev_)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda ctx: lib.cvx_split_f16_dev(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), 1.0, None, C.addressof(ctx))
    bare = _lib.Ctx(st, None, 0, 0)
    assert call(bare) != 0 and b"saturation flag" in lib.cvx_last_error_string()
    assert lib.cvx_saturation_flag_reset(C.addressof(bare)) != 0
    waived = _lib.Ctx(st, None, 0, _lib.CTX_NO_SATURATION_FLAG)
    assert call(waived) == 0
    torch.cuda.synchronize()
    assert float(hi.float().abs().max()) == 65504.0                   # clamped - the caller asked for no bookkeeping
    f1, f2 = torch.zeros(1, dtype=torch.int32, device=dev_), torch.zeros(1, dtype=torch.int32, device=dev_)
    torch.cuda.synchronize()
    c1, c2 = _lib.Ctx(st, f1.data_ptr(), 0, 0), _lib.Ctx(st, f2.data_ptr(), 0, 0)
    assert call(c1) == 0
    x[3, 3] = 1.0
    assert call(c2) == 0
    v = C.c_uint32(0)
    assert lib.cvx_saturation_flag_query(C.byref(v), 1, C.addressof(c1)) == 0 and v.value != 0
    assert lib.cvx_saturation_flag_query(C.byref(v), 0, C.addressof(c2)) == 0 and v.value == 0
    assert int(f1.item()) == 0                                        # (the query of c1 reset it)
    assert lib.cvx_split_f16_dev(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), 1.0, None, None) != 0      # NULL context: no flag either


def test_two_threads_two_models_one_device():
    """Two host threads on one device, each with its own stream and model: only the thread whose input leaves the window is
    flagged (fp32 re-run, warning); the other thread's result is its undisturbed split-precision result."""
    import threading
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    dev_ = dev()
    shapes = syn.acoustic_param_shapes(dim=128, dim_emb=64, depth=4, heads=2, dim_cond=80, streams=1)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    inp = syn.synthetic_inputs("vosingle", 1, 200, 80, seed=5)
    ids, cond, mask = inp["phoneme_ids"].to(dev_), inp["cond"].to(dev_), inp["mask"].to(dev_)
    y0 = torch.randn(1, 200, 80, device=dev_)
    ref = CoVoMixModel.from_state_dict(sd, nfe=4).eval().to(dev_).synthesis_sample(ids, cond, mask, 0.7, y0=y0)
    res, reran = {}, {}

    def run(name, scale):
        st = torch.cuda.Stream(device=dev_)
        with torch.cuda.stream(st):
            m = CoVoMixModel.from_state_dict(sd, nfe=4).eval().to(dev_)
            for _ in range(6):
                out = m.synthesis_sample(ids, cond * scale, mask, 0.7, y0=y0)
            st.synchronize()
            res[name] = out
            reran[name] = m._field_fp32 is not None      # (the exact-fp32 twin is built by the first flagged call; the warnings
                                                         #  module's recorder is process-global, so it cannot tell the threads apart)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ta, tb = threading.Thread(target=run, args=("hot", 4000.0)), threading.Thread(target=run, args=("cool", 1.0))
        ta.start(); tb.start(); ta.join(); tb.join()
    assert reran["hot"] and not reran["cool"]
    assert rel_l2(res["cool"], ref) < 1e-6
