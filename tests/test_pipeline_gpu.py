"""CU-partitioned streams and the two-stage schedule of BASELINE config 5 (covomix_amd/pipeline.py, config5.py): the text2semantic
decode of batch k + 1 under the acoustic solve + vocoder of batch k (reference loop dialogue_generation.py:272-329, decode
covomix/covomix_model/text2semantic.py:749-848)."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cu_partition_masks_and_cu_counts():
    """The partition's streams own disjoint CU sets (28 + 4 per XCD on MI355X); the library sizes persistent grids from the stream's
    CU count; a capture / side stream inherits it through saturation_share; a GEMM on either stream equals the default stream's bits."""
    from covomix_amd import ops
    dev = torch.device("cuda:0")
    n = torch.cuda.get_device_properties(dev).multi_processor_count
    part = ops.cu_partition(dev)
    assert part is ops.cu_partition(dev)
    assert part.n_main + part.n_side == n and part.n_side == 32 and part.n_main % 32 == 0
    assert ops.stream_cus(part.main) == part.n_main and ops.stream_cus(part.side) == part.n_side
    assert ops.stream_cus(torch.cuda.current_stream()) == n
    cap = torch.cuda.Stream(device=dev)
    ops.saturation_share(part.main, cap)
    assert ops.stream_cus(cap) == part.n_main
    ops.saturation_share(torch.cuda.current_stream(), cap)
    assert ops.stream_cus(cap) == n
    with pytest.raises(ValueError):
        ops.CUPartition(dev, side_per_xcd=2)           # shader engines would be left with different CU counts
    g = torch.Generator().manual_seed(0)
    M, N, K = 4608, 1024, 1024                           # 18 row panels x 4: more tiles than the side stream's CUs
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / 32).to(dev)
    ws = ops.split_f16(w)
    wil = ops.split_f16_interleaved(ws)
    il = ops.SplitIL(M, K, dev)
    ops.split_act_f16(x, il)
    outs = []
    for st in (torch.cuda.current_stream(), part.main, part.side):
        with torch.cuda.stream(st), ops.gemm_flags(16):             # (pinned to the large-problem kernel: the choice depends on the CU count)
            c = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il)
            st.synchronize()
            outs.append(c)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_two_stage_runner_order_errors_and_overlap():
    """run_two_stage: results in item order, stage1 at most `depth` ahead, an exception of either stage reaches the caller, and the
    stages really run on the partition's two streams in two threads."""
    from covomix_amd import ops, pipeline
    dev = torch.device("cuda:0")
    part = ops.cu_partition(dev)
    seen = {"s1": set(), "s2": set(), "t1": set(), "t2": set()}
    ahead = []
    done2 = [0]

    def s1(i):
        seen["s1"].add(torch.cuda.current_stream().cuda_stream); seen["t1"].add(threading.get_ident())
        ahead.append(i - done2[0])
        return torch.full((4,), float(i), device=dev)

    def s2(x):
        seen["s2"].add(torch.cuda.current_stream().cuda_stream); seen["t2"].add(threading.get_ident())
        r = float((x * 2).sum())
        done2[0] += 1
        return r
    for overlap in (True, False):
        for k in seen:
            seen[k].clear()
        ahead.clear(); done2[0] = 0
        out = pipeline.run_two_stage(range(9), s1, s2, dev, overlap=overlap, depth=2)
        assert out == [8.0 * i for i in range(9)]
        # (pipelined: the first item runs on a plain stream - nothing to hide under yet - the others on the side stream)
        assert part.side.cuda_stream in seen["s1"] and part.main.cuda_stream not in seen["s1"] and len(seen["s1"]) == (2 if overlap else 1)
        assert seen["s2"] == {part.main.cuda_stream}
        assert (seen["t1"] != seen["t2"]) == overlap
        assert max(ahead) <= 4          # queue depth 2 + the item in flight on either side

    def bad1(i):
        if i == 3:
            raise RuntimeError("stage one failed")
        return i
    with pytest.raises(RuntimeError, match="stage one failed"):
        pipeline.run_two_stage(range(6), bad1, lambda x: x, dev)

    def bad2(x):
        if x == 2:
            raise ValueError("stage two failed")
        return x
    with pytest.raises(ValueError, match="stage two failed"):
        pipeline.run_two_stage(range(50), lambda i: i, bad2, dev)
    assert pipeline.run_two_stage(range(3), lambda i: i, lambda x: x + 1, dev) == [1, 2, 3]      # (and the runner is reusable afterwards)
    # finish: stage 2 only enqueues and hands back a handle; the handle of batch k is finished after batch k + 1 was enqueued
    log = []
    out = pipeline.run_two_stage(range(4), lambda i: i, lambda x: (log.append(("launch", x)), x)[1], dev,
                                 finish=lambda h: (log.append(("finish", h)), 10 * h)[1])
    assert out == [0, 10, 20, 30]
    assert log == [("launch", 0), ("launch", 1), ("finish", 0), ("launch", 2), ("finish", 1), ("launch", 3), ("finish", 2), ("finish", 3)]


@pytest.fixture(scope="module")
def c5():
    from covomix_amd.config5 import Config5
    return Config5(torch.device("cuda:0"), tokens=48, prompt=80, nfe=4)


def test_config5_pipelined_bits_equal_alternate_schedule(c5):
    """Config 5 at test size (full-width CoMix / VoMix / HiFi-GAN, 48 decoded steps, 4 NFE; text2semantic in groups of 4, the solve in
    batches of 3): the pipelined schedule returns the SAME tokens and the SAME PCM as the same calls run one after the other on the
    same two streams; against the unpartitioned single-stream schedule with another acoustic batch size the tokens are identical and
    the PCM within one LSB (the GEMM kernel choice depends on rows and CU count)."""
    n = 10
    alt = c5.run(n, 3, overlap=False, B1=4)
    pip = c5.run(n, 3, overlap=True, B1=4)
    ser = c5.run(n, 5, overlap=False, partitioned=False, B1=4)
    assert [r["j"] for r in alt] == [r["j"] for r in pip] == [r["j"] for r in ser] == list(range(n))
    for a, p, s in zip(alt, pip, ser):
        assert a["streams"].shape == (2, c5.tokens) and a["pcm"].dtype == torch.int16
        assert torch.equal(a["streams"], p["streams"]) and torch.equal(a["pcm"], p["pcm"])
        assert torch.equal(a["streams"], s["streams"])
        assert int((a["pcm"].int() - s["pcm"].int()).abs().max()) <= 1
        assert int(a["pcm"].int().abs().max()) > 0
    assert not torch.equal(alt[0]["streams"], alt[1]["streams"])          # (different dialogues decode different tokens)


def test_config5_64_per_decode_pass_bits_equal_8_per_pass(c5):
    """Round 6: the decode batch is no longer capped at 8.  The schedule bench.py's `c5` reports and `--pipeline auto` runs - every
    dialogue's text2semantic in ONE wide lock-step pass on the whole chip, then the acoustic batches, all on one stream - returns the SAME
    tokens and the SAME PCM as the same schedule with small decode passes (here 24 dialogues: one pass of 24 slots against passes of 4)."""
    n = 24
    small = c5.run(n, 5, overlap=False, partitioned=False, B1=4)
    big = c5.run(n, 5, overlap=False, partitioned=False, B1=24)
    huge = c5.run(n, 5, overlap=False, partitioned=False, B1=64)
    assert [r["j"] for r in small] == [r["j"] for r in big] == [r["j"] for r in huge] == list(range(n))
    for a, b, c in zip(small, big, huge):
        assert torch.equal(a["streams"], b["streams"]) and torch.equal(a["pcm"], b["pcm"])
        assert torch.equal(a["streams"], c["streams"]) and torch.equal(a["pcm"], c["pcm"])


def test_continuous_batching_on_dialogues_that_end_at_different_steps():
    """Real dialogues sample their eos at different steps (text2semantic.py:803-818); a lock-step decode batch then runs half empty.
    Full-size CoMix, dialogues whose last step is spread over 100 ... 608, 8 decode slots refilled on the device (t2s.generate_many):
    every dialogue gets the tokens of its one-by-one decode, and the decode stage delivers >= 0.85 of the useful tokens/s of the
    fixed-length case (every dialogue 608 steps through the same 8 slots) on a queue of 64 dialogues.  What is lost is the TAIL: when
    the queue is empty the last dialogues finish in half-empty steps, and a step costs the same whatever runs in it - with only 32
    dialogues (4 per slot) that alone caps the ratio at ~0.8 (measured 0.79), which is asserted too, lower."""
    import time
    import covomix_amd.synthetic as syn
    from covomix_amd.t2s import TextToSemanticDecoder
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(syn.t2s_param_shapes(two_output=True, dim=512, dim_target=1024), seed=0).items()}
    m = TextToSemanticDecoder(sd, dev, max_length=608)
    g = torch.Generator().manual_seed(5)
    N, slots, steps = 64, 8, 608
    lims = torch.randint(100, steps + 1, (N,), generator=g).tolist()
    srcs = [torch.randint(1, 30000, (1, 40 + j % 30), generator=g) for j in range(N)]
    unis = [torch.rand(steps, 2, 502, generator=torch.Generator().manual_seed(100 + j)) for j in range(N)]

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return r, time.perf_counter() - t0
    m.generate_many(srcs, unis, slots=slots, ignore_eos=True, limits=[20] * N)                  # graphs and buffers of the timed shapes
    m.generate_batch(srcs[:slots], unis[:slots], max_length=16, ignore_eos=True)
    for n, bar in ((64, 0.85), (32, 0.72)):
        res, t_many = timed(lambda: m.generate_many(srcs[:n], unis[:n], slots=slots, ignore_eos=True, limits=lims[:n]))
        _, t_fix = timed(lambda: [m.generate_batch(srcs[w:w + slots], unis[w:w + slots], ignore_eos=True) for w in range(0, n, slots)])
        ratio = (sum(lims[:n]) / t_many) / (n * steps / t_fix)
        print(f"continuous batching, {n} dialogues on {slots} slots: {sum(lims[:n])} useful steps in {t_many * 1e3:.0f} ms; fixed length "
              f"{n * steps} steps in {t_fix * 1e3:.0f} ms; ratio {ratio:.3f}")
        for j in (0, 7, 8, 19, n - 1):
            alone = m.generate_batch([srcs[j]], [unis[j][: lims[j]]], ignore_eos=True)[0]
            assert res[j][1].shape == (2, lims[j]) and torch.equal(res[j][1], alone[1].cpu()), j
        assert ratio >= bar, (n, ratio)


def test_regroup_collate():
    from covomix_amd import pipeline
    out = list(pipeline.regroup(3)(iter([[1, 2], [3, 4, 5, 6], [7]])))
    assert out == [[1, 2, 3], [4, 5, 6], [7]]
