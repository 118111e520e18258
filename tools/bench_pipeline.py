#!/usr/bin/env python3
"""Dev: BASELINE config 5 in one process - CoMix text2semantic AR decode + VoMix 64-NFE + HiFi-GAN for 8 dialogues per
GPU (recipe weights, synthetic inputs).  Stage times and dialogues/s; the text2semantic stage runs the utterances one
after the other at batch 1 like the reference scripts do.  TOKENS = decoded steps per utterance (eos ignored so the
work is fixed), T = TOKENS + 400 prompt frames."""
import os, sys, time, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd import ops
from covomix_amd.conditional_model import CoVoMixModel
from covomix_amd.t2s import TextToSemanticDecoder, CHUNK
from covomix_amd.vocoder import AttrDict, Generator
dev = torch.device("cuda:0")
B, TOK, PROMPT = 8, int(os.environ.get("TOKENS", "608")), 400
T = TOK + PROMPT
with contextlib.redirect_stdout(sys.stderr):
    t2s = TextToSemanticDecoder({k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(
        syn.t2s_param_shapes(two_output=True, dim=512, dim_target=1024), seed=0).items()}, dev)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    model = CoVoMixModel.from_state_dict(sd, nfe=64).eval().to(dev)
    gen = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)).to(dev)
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()})
    gen.eval(); gen.remove_weight_norm()
inp = syn.synthetic_inputs("vomix", B, T, PROMPT, seed=1)
ids, cond, mask = inp["phoneme_ids"].to(dev), inp["cond"].to(dev), inp["mask"].to(dev)
texts = [torch.randint(1, 30000, (1, 64)) for _ in range(B)]

T2S_BATCH = int(os.environ.get("T2S_BATCH", "8"))          # utterances decoded together (1 = one by one, like the reference)


def t2s_stage():
    for g0 in range(0, B, T2S_BATCH):
        group = texts[g0:g0 + T2S_BATCH]
        nb = len(group)
        rows = []
        for i, src in enumerate(group):
            enc = t2s.encode(src)
            rows.append(enc.shape[0] + 1)
            for L in t2s.dec:
                L["kv_c"][i, 0].copy_(L["null"]); ops.gemm(enc, L["wkv_c"], L["kv_c"][i, 1:enc.shape[0] + 1])
        t2s.buf["uniforms"].uniform_(1e-6, 1 - 1e-6)
        t2s.buf["x"][:nb].copy_(t2s.start[None, :].expand(nb, -1))
        t2s.buf["state"].copy_(torch.tensor([[0, 0, 0, rows[i] if i < nb else 1] for i in range(8)], dtype=torch.int32))
        for _ in range(TOK // CHUNK):
            t2s._run_chunk(1.0, nb); t2s.buf["state"].tolist()


def acoustic_stage():
    return model.synthesis_sample(ids, cond, mask, 0.7)

def vocoder_stage(mel):
    return ops.wav_to_int16(gen(mel.permute(0, 2, 1).contiguous()).squeeze(1).contiguous())

def timed(fn, *a):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a); torch.cuda.synchronize()
    return r, time.perf_counter() - t0
t2s_stage(); mel = acoustic_stage(); vocoder_stage(mel)          # warm-up (graph capture, buffers)
_, a = timed(t2s_stage); mel, b = timed(acoustic_stage); _, c = timed(vocoder_stage, mel)
tot = a + b + c
print(f"config 5 per GPU, {B} dialogues, T={T} frames ({TOK} decoded steps x 2 streams): text2semantic {a*1e3:.0f} ms, "
      f"(T2S batch {T2S_BATCH}) VoMix 64-NFE {b*1e3:.0f} ms, HiFi-GAN {c*1e3:.0f} ms -> {B/tot:.2f} dialogues/s = {B*T/tot:.0f} mel-frames/s end to end")
