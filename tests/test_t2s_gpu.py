"""GPU: text2semantic through the C ABI against the golden vectors the REFERENCE
TextToSemantic produced (tests/golden/make_golden_t2s.py):
  * encoder output           <= 1e-4 rel-L2 (fp32 kernels; measured ~1e-6);
  * per-step pre-filter logits of the free-running decode <= 1e-4 rel-L2 against the reference's teacher-forced logits;
  * sampled tokens from the recorded uniform draws: BIT-EXACT (every fixture records the smallest top-2 margin of the
    Gumbel-perturbed logits, >= 9e-3 - three orders of magnitude above the fp32 arithmetic noise)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

KW = {
    "cosingle": dict(two_output=False, dim=512, dim_target=512),
    "comix": dict(two_output=True, dim=512, dim_target=1024),
}
TOL = 1e-4


def load_case(name):
    import covomix_amd.synthetic as syn
    g = np.load(os.path.join(GOLDEN, f"t2s_{name}.npz"))
    if name.endswith("_small"):
        sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    else:
        sd = {k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(syn.t2s_param_shapes(**KW[name]), seed=0).items()}
    return g, sd


@pytest.fixture(scope="module", params=["cosingle_small", "comix_small", "cosingle", "comix"])
def case(request):
    from covomix_amd.t2s import TextToSemanticDecoder
    g, sd = load_case(request.param)
    return request.param, g, TextToSemanticDecoder(sd, torch.device("cuda:0"), max_length=256)


def test_encoder_vs_reference_golden(case):
    name, g, model = case
    enc = model.encode(torch.from_numpy(g["source_ids"]))
    e = rel_l2(enc, torch.from_numpy(g["encoder"])[0])
    print(name, "encoder", e)
    assert e < TOL


def test_step_logits_and_tokens_vs_reference_golden(case):
    name, g, model = case
    src, uni = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"])
    flat, streams, logits = model.generate(src, uniforms=uni, collect_logits=True)
    ref_logits = torch.from_numpy(g["logits"])[:, :, 0, :]                  # [L, S, V]
    assert logits.shape == ref_logits.shape
    e = rel_l2(logits, ref_logits)
    print(name, "step logits", e, "min margin", float(g["min_margin"]))
    assert e < TOL
    assert torch.equal(streams.cpu(), torch.from_numpy(g["streams"])[0])
    assert torch.equal(flat.cpu(), torch.from_numpy(g["tokens"]))


def test_graph_replay_matches_stepwise_and_is_reusable(case):
    name, g, model = case
    src, uni = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"])
    want = torch.from_numpy(g["tokens"])
    for _ in range(2):                                                     # second call reuses the captured graph
        got = model.generate(src, uniforms=uni)
        assert torch.equal(got.cpu(), want)
    # a different (shorter) text through the same graph: the context length is read on the device
    short = src[:, : src.shape[1] // 2]
    a = model.generate(short, uniforms=uni)
    os.environ["CVX_GRAPH"] = "0"
    try:
        b = model.generate(short, uniforms=uni)
    finally:
        os.environ["CVX_GRAPH"] = "1"
    assert torch.equal(a, b)


def test_random_draws_and_max_length(case):
    name, g, model = case
    src = torch.from_numpy(g["source_ids"])
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    out, streams = model.generate(src, max_length=20, generator=gen, return_streams=True)
    S = streams.shape[0]
    assert streams.shape[1] <= 20 and out.dtype == torch.int64 and int(out.max()) <= 501 and int(out.min()) >= 0
    assert out.numel() <= S * streams.shape[1]
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    again = model.generate(src, max_length=20, generator=gen)
    assert torch.equal(out, again)


def test_rejects_padded_batches(case):
    _, g, model = case
    with pytest.raises(NotImplementedError):
        model.generate(torch.tensor([[5, 6, 0, 0]]))
    with pytest.raises(NotImplementedError):
        model.generate(torch.tensor([[5, 6], [7, 8]]))


def test_facade_synthesis_sample_text2semantic():
    """CoVoMixModel.synthesis_sample_text2semantic (conditional_model.py:313-321) on the comix fixture."""
    from covomix_amd.conditional_model import CoVoMixModel
    g, sd = load_case("comix_small")
    m = CoVoMixModel.from_state_dict(sd).eval().to("cuda:0")
    assert m.is_text2semantic
    src = torch.from_numpy(g["source_ids"])
    out = m.synthesis_sample_text2semantic(src, uniforms=torch.from_numpy(g["uniforms"]))
    assert out.device == src.device and torch.equal(out, torch.from_numpy(g["tokens"]))
    half = out.shape[0] // 2                                   # comix_pred (monologue_generation.py:307-319) splits the halves
    assert torch.equal(out[:half], torch.from_numpy(g["streams"])[0, 0]) and torch.equal(out[half:], torch.from_numpy(g["streams"])[0, 1])


def test_batched_decode_is_bit_identical_to_one_by_one(case):
    """generate_batch advances several utterances together (shared weight streaming); every utterance must get exactly
    the tokens AND logits it gets alone - here the golden utterance next to shorter / longer texts, in slot 0 and 2."""
    name, g, model = case
    src, uni = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"])
    S, V = uni.shape[1], uni.shape[-1]
    gen = torch.Generator().manual_seed(11)
    other = [src[:, :5], torch.cat((src, src[:, :7]), dim=1), src[:, 2:9]]
    other_u = [torch.rand(uni.shape[0], S, V, generator=gen) for _ in other]
    alone = [model.generate(o, uniforms=u, return_streams=True) for o, u in zip(other, other_u)]
    for slot in (0, 2):
        sources = list(other); unis = list(other_u)
        sources.insert(slot, src); unis.insert(slot, uni)
        res = model.generate_batch(sources, unis, collect_logits=True)
        assert torch.equal(res[slot][0].cpu(), torch.from_numpy(g["tokens"]))
        assert rel_l2(res[slot][2], torch.from_numpy(g["logits"])[:, :, 0, :]) < TOL
        ref = list(alone); ref.insert(slot, None)
        for i, r in enumerate(res):
            if i != slot:
                assert torch.equal(r[0], ref[i][0]) and torch.equal(r[1], ref[i][1])
    # graph path, full batch of 8
    res8 = model.generate_batch([src] * 8, [uni] * 8)
    for r in res8:
        assert torch.equal(r[0].cpu(), torch.from_numpy(g["tokens"]))


def _varied(g, n, steps, seed):
    """n utterances: texts of different length cut from the golden one, own uniform draws"""
    src = torch.from_numpy(g["source_ids"])
    uni = torch.from_numpy(g["uniforms"])
    S, V = uni.shape[1], uni.shape[-1]
    gen = torch.Generator().manual_seed(seed)
    L = src.shape[1]
    srcs, unis = [], []
    for i in range(n):
        a = int(torch.randint(0, max(1, L // 2), (1,), generator=gen))
        e = int(torch.randint(a + 3, L + 1, (1,), generator=gen))
        srcs.append(src[:, a:e] if i % 5 else torch.cat((src, src[:, : 1 + i % 7]), dim=1))
        unis.append(torch.rand(steps, S, V, generator=gen))
    return srcs, unis


@pytest.mark.parametrize("nb", [16, 32, 64])
def test_large_decode_batches_are_bit_identical_to_one_by_one(case, nb):
    """Round 6: 16 / 32 / 64 decode slots per step (groups of 8 slots; the kernels read every slot's own position).  Every
    utterance must get exactly the tokens it gets alone - the golden utterance in slot nb - 3 against the reference's tokens."""
    name, g, model = case
    if nb > 16 and not name.endswith("_small") and name != "comix":
        pytest.skip("the wide batches run on both small fixtures and on the full CoMix shape")
    steps = g["uniforms"].shape[0]          # (a lock-step batch runs as many steps as its shortest list of draws)
    srcs, unis = _varied(g, nb, steps, seed=100 + nb)
    slot = nb - 3
    srcs[slot], unis[slot] = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"])
    res = model.generate_batch(srcs, unis)
    assert torch.equal(res[slot][0].cpu(), torch.from_numpy(g["tokens"]))
    for i in sorted({0, 1, 7, 8, 9, nb // 2, nb - 9, nb - 1}):
        alone = model.generate(srcs[i], uniforms=unis[i], return_streams=True)
        assert torch.equal(res[i][0], alone[0]) and torch.equal(res[i][1], alone[1]), (name, nb, i)


@pytest.mark.parametrize("slots", [8, 32])
def test_continuous_batching_tokens_equal_one_by_one(case, slots):
    """generate_many: utterances that end at DIFFERENT steps (their own eos, text2semantic.py:803-818, or their step limit) run
    through continuously refilled decode slots - the sampling kernel hands a finished slot the next pending utterance on the
    device.  Every utterance gets exactly the tokens the one-by-one decode gives it, whatever slot and neighbours it had."""
    name, g, model = case
    n, steps = 44, 96
    srcs, unis = _varied(g, n, steps, seed=7)
    gold = torch.from_numpy(g["uniforms"])[:, :, 0, :]
    srcs[5], unis[5] = torch.from_numpy(g["source_ids"]), torch.cat((gold, unis[5][gold.shape[0]:]))
    gen = torch.Generator().manual_seed(1)
    limits = torch.randint(10, steps + 1, (n,), generator=gen).tolist()
    limits[5] = gold.shape[0]
    done = []
    res = model.generate_many(srcs, unis, slots=slots, limits=limits, on_done=lambda j, r: done.append(j))
    assert sorted(done) == list(range(n))
    rec = model.last_records
    by_eos = sum(1 for j in range(n) if rec[j][3] == 2)
    by_limit = sum(1 for j in range(n) if rec[j][3] == 3)
    used = {rec[j][5] for j in range(n)}
    print(name, f"{slots} slots: {by_eos} utterances ended by their eos, {by_limit} by their limit; slots used: {len(used)}")
    assert by_eos + by_limit == n and by_limit > 0 and len(used) == min(slots, n)
    for j in range(n):
        alone = model.generate(srcs[j], uniforms=unis[j][: limits[j]], return_streams=True)
        assert torch.equal(res[j][0], alone[0].cpu()) and torch.equal(res[j][1], alone[1].cpu()), (name, slots, j, rec[j])
    assert torch.equal(res[5][0].cpu(), torch.from_numpy(g["tokens"]))          # the reference's tokens for the golden utterance


@pytest.mark.parametrize("name", ["cosingle_small", "cosingle"])
def test_classifier_free_guidance_vs_reference_golden(name):
    """cond_scale = 1.5 (text2semantic.py:780-792): tokens sampled from null + (cond - null) * scale, BIT-EXACT against the
    reference built with cond_drop_prob > 0 (tests/golden/t2s_*_cfg.npz; top-2 margin >= 2e-2), the combined logits against the
    oracle's, stepwise and graph-replayed alike; two utterances in one guided batch decode like each alone."""
    from covomix_amd.t2s import TextToSemanticDecoder
    _, sd = load_case(name)
    g = np.load(os.path.join(GOLDEN, f"t2s_{name}_cfg.npz"))
    model = TextToSemanticDecoder(sd, torch.device("cuda:0"), max_length=256)
    src, uni, scale = torch.from_numpy(g["source_ids"]), torch.from_numpy(g["uniforms"]), float(g["cond_scale"])
    flat, streams, logits = model.generate(src, uniforms=uni, collect_logits=True, cond_scale=scale)
    ref_logits = torch.from_numpy(g["logits"])[:, :, 0, :]
    e = rel_l2(logits[:, None, :] if logits.ndim == 2 else logits, ref_logits)
    print(name, "guided logits vs oracle", e, "min margin", float(g["min_margin"]))
    assert e < TOL
    assert torch.equal(flat.cpu(), torch.from_numpy(g["tokens"]))
    flat2 = model.generate(src, uniforms=uni, cond_scale=scale)                    # chunked graph replay
    assert torch.equal(flat2.cpu(), torch.from_numpy(g["tokens"]))
    unguided = model.generate(src, uniforms=uni)
    assert not torch.equal(unguided.cpu(), torch.from_numpy(g["tokens"]))
    g1 = np.load(os.path.join(GOLDEN, f"t2s_{name}.npz"))                          # a second utterance in the same guided batch
    src_b, uni_b = torch.from_numpy(g1["source_ids"]), torch.from_numpy(g1["uniforms"])
    alone_b = model.generate(src_b, uniforms=uni_b, cond_scale=scale)
    both = model.generate_batch([src, src_b], [uni, uni_b], cond_scale=scale)
    n = min(uni.shape[0], uni_b.shape[0])          # (a batch runs as many steps as its shortest list of draws provides)
    assert torch.equal(both[0][0].cpu()[:n], torch.from_numpy(g["tokens"])[:n]) and torch.equal(both[1][0][:n], alone_b[:n])


def test_guidance_follows_the_reference_asserts():
    from covomix_amd.conditional_model import CoVoMixModel
    _, sd = load_case("cosingle_small")
    ids = torch.from_numpy(np.load(os.path.join(GOLDEN, "t2s_cosingle_small_cfg.npz"))["source_ids"])
    plain = CoVoMixModel.from_state_dict(sd).eval().to("cuda:0")
    with pytest.raises(AssertionError):             # text2semantic.py:684: cond_drop_prob == 0 (the default) forbids guidance
        plain.synthesis_sample_text2semantic(ids, cond_scale=1.5)
    g = np.load(os.path.join(GOLDEN, "t2s_cosingle_small_cfg.npz"))
    guided = CoVoMixModel(sd, hparams={"cond_drop_prob": 0.25, "text2semantic": True}).eval().to("cuda:0")
    out = guided.synthesis_sample_text2semantic(ids, cond_scale=float(g["cond_scale"]), uniforms=torch.from_numpy(g["uniforms"]))
    assert torch.equal(out.cpu(), torch.from_numpy(g["tokens"]))
    _, sd2 = load_case("comix_small")
    two = CoVoMixModel(sd2, hparams={"cond_drop_prob": 0.25, "text2semantic": True}).eval().to("cuda:0")
    with pytest.raises(NotImplementedError):
        two.synthesis_sample_text2semantic(ids, cond_scale=1.5)
