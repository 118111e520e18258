#!/usr/bin/env python3
"""Dev: text2semantic decode rate (us/step, tokens/s) for the CoSingle / CoMix configurations with recipe weights at decode batches
1 ... 64: encoder once, then N token steps through the graph-replayed decode loop (eos ignored so that the step count is fixed), then
the continuous-batching path (generate_many) on utterances that END at different steps against the lock-step batch.
Also prints the weight bytes a token step has to stream (the HBM/MALL roofline of a batch-1 decode).
    TOKENS=512 SIDE=0 BATCHES=1,8,16,32,64 python tools/bench_t2s.py [comix|cosingle]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.t2s import TextToSemanticDecoder, CHUNK
from covomix_amd import ops
dev = torch.device("cuda:0")
N = int(os.environ.get("TOKENS", "512"))
BATCHES = [int(x) for x in os.environ.get("BATCHES", "1,2,4,8,16,32,64").split(",")]
which = sys.argv[1:] or ["cosingle", "comix"]
if os.environ.get("SIDE", "0") == "1":          # on the 32-CU side stream of the CU partition (pipeline.py)
    torch.cuda.set_stream(ops.cu_partition(dev).side)
    print("on the side stream of the CU partition:", ops.stream_cus(), "CUs")
CFG = {"cosingle": dict(two_output=False, dim=512, dim_target=512), "comix": dict(two_output=True, dim=512, dim_target=1024)}
for name in which:
    kw = CFG[name]
    sd = {k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(syn.t2s_param_shapes(**kw), seed=0).items()}
    m = TextToSemanticDecoder(sd, dev, max_length=2048)
    src = torch.randint(1, 30000, (1, 48))
    wbytes = sum(L[k].numel() * 4 for L in m.dec for k in ("wqkv_s", "wo_s", "wq_c", "wo_c", "w1", "w2")) + m.emb.numel() * 4
    t0 = time.perf_counter(); enc = m.encode(src); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); enc = m.encode(src); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
    print(f"{name}: encoder {t_enc*1e3:.2f} ms, weights per token step {wbytes/1e6:.1f} MB")
    for nb in BATCHES:                                      # slots decoded together, lock step
        m._ensure(nb, nb, N)
        m._graph(1.0, nb)
        ctx = m._contexts([src] * nb)
        m.buf["uniforms"].uniform_(1e-6, 1 - 1e-6)
        m.buf["x"][:nb].copy_(m.start[None, :].expand(nb, -1))
        m.buf["state"].copy_(m._slot_records(ctx))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N // CHUNK):
            m._run_chunk(1.0, nb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"   batch {nb:2d}: {dt/N*1e6:7.1f} us/step = {nb*N/dt:8.0f} tokens/s; weight streaming {wbytes/(dt/N)/1e12:.2f} TB/s", flush=True)
    if os.environ.get("MANY", "1") == "1":
        # utterances that end at different steps: limits spread over [100, 608]
        g = torch.Generator().manual_seed(3)
        n = int(os.environ.get("UTTS", "64"))
        lims = torch.randint(100, 609, (n,), generator=g).tolist()
        srcs = [torch.randint(1, 30000, (1, 48), generator=g) for _ in range(n)]
        for slots in (8, 32, 64):
            m.generate_many(srcs, max_length=608, slots=slots, ignore_eos=True, limits=[20] * n)      # graph + buffers of the timed shape
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.generate_many(srcs, max_length=608, slots=slots, ignore_eos=True, limits=lims)
            torch.cuda.synchronize(); t_many = time.perf_counter() - t0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for w in range(0, n, slots):
                m.generate_batch(srcs[w:w + slots], max_length=max(lims[w:w + slots]), ignore_eos=True)
            torch.cuda.synchronize(); t_lock = time.perf_counter() - t0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for w in range(0, n, slots):
                m.generate_batch(srcs[w:w + slots], max_length=608, ignore_eos=True)
            torch.cuda.synchronize(); t_fix = time.perf_counter() - t0
            print(f"   {n} utterances, limits 100..608 (sum {sum(lims)}), {slots} slots: continuous {t_many*1e3:7.1f} ms = {sum(lims)/t_many:8.0f} useful tokens/s;"
                  f" lock step {t_lock*1e3:7.1f} ms = {sum(lims)/t_lock:8.0f}; fixed 608 for all {t_fix*1e3:7.1f} ms = {n*608/t_fix:8.0f} tokens/s", flush=True)
