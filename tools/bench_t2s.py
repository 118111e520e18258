#!/usr/bin/env python3
"""Dev: text2semantic decode rate (tokens/s, us/token) for the CoSingle / CoMix configurations with recipe weights:
encoder once, then N token steps through the graph-replayed decode loop (eos ignored so that the step count is fixed).
Also prints the weight bytes a token step has to stream (the HBM/MALL roofline of a batch-1 decode)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd.t2s import TextToSemanticDecoder, CHUNK
from covomix_amd import ops
dev = torch.device("cuda:0")
N = int(os.environ.get("TOKENS", "512"))
if os.environ.get("SIDE", "0") == "1":          # on the 32-CU side stream of the CU partition (pipeline.py)
    torch.cuda.set_stream(ops.cu_partition(dev).side)
    print("on the side stream of the CU partition:", ops.stream_cus(), "CUs")
for name, kw in [("cosingle", dict(two_output=False, dim=512, dim_target=512)), ("comix", dict(two_output=True, dim=512, dim_target=1024))]:
    sd = {k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(syn.t2s_param_shapes(**kw), seed=0).items()}
    m = TextToSemanticDecoder(sd, dev, max_length=2048)
    src = torch.randint(1, 30000, (1, 48))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.generate(src, max_length=CHUNK)                       # encoder + graph capture + one chunk
    torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    t0 = time.perf_counter(); enc = m.encode(src); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
    m.buf["x"].copy_(m.start); m.buf["state"].copy_(torch.tensor([0, 0, 0, enc.shape[0] + 1], dtype=torch.int32))
    m.buf["uniforms"].uniform_(1e-6, 1 - 1e-6)
    wbytes = sum(L[k].numel() * 4 for L in m.dec for k in ("wqkv_s", "wo_s", "wq_c", "wo_c", "w1", "w2")) + m.emb.numel() * 4
    print(f"{name}: first call {t_first*1e3:.1f} ms (capture), encoder {t_enc*1e3:.2f} ms, weights per token step {wbytes/1e6:.1f} MB")
    for nb in (1, 2, 4, 8):                                   # utterances decoded together
        for i in range(nb):
            for L in m.dec:
                L["kv_c"][i, : enc.shape[0] + 1].copy_(m.dec[0]["kv_c"][0, : enc.shape[0] + 1])
        m.buf["x"][:nb].copy_(m.start[None, :].expand(nb, -1))
        m.buf["state"].copy_(torch.tensor([[0, 0, 0, enc.shape[0] + 1]] * 8, dtype=torch.int32))
        m._run_chunk(1.0, nb)                                 # capture for this batch size
        m.buf["state"].copy_(torch.tensor([[0, 0, 0, enc.shape[0] + 1]] * 8, dtype=torch.int32))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N // CHUNK):
            m._run_chunk(1.0, nb)
            m.buf["state"].tolist()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"   batch {nb}: {dt/N*1e6:7.1f} us/step = {nb*N/dt:7.0f} tokens/s; weight streaming {wbytes/(dt/N)/1e12:.2f} TB/s")
