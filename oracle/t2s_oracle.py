"""CPU restatement (torch fp32, functional) of the reference text2semantic AR decode - SURVEY.md section 8f row N1.

TEST INFRASTRUCTURE ONLY: nothing under neurips2024-covomix_amd/ imports this file; only tests/, __graft_entry__.smoke()
and bench tools may, and only as the checker.

Follows (reference file:line, all under covomix/covomix_model/):
  * TextToSemantic.generate, sampling branch            text2semantic.py:662-848 (loop :748-820)
  * Transformer.forward (cache handling, layer order)   text2semantic.py:308-383
  * Attention.forward (to_q / to_kv, cache of UN-rotated k, rotary, learned null kv for cross-attention)  :225-270
  * RMSNorm :143-151, GEGLU feed-forward :154-167
  * Attend.forward (mask, causal mask for q_len != k_len) attend_t2s.py:126-171
  * RotaryEmbedding.rotate_queries_with_cached_keys (interleaved pairs, queries take the LAST q_len positions)
    rotary_embedding_torch.py:25-41, :146-157
  * top_k :126-132, gumbel_noise / gumbel_sample :105-113, set_eos_id :59-67, mask_after_eos :73-76
  * TextToSemanticWrapper.sample (target[target_mask]) :1237-1251

Pinned against the imported reference by tests/golden/make_golden_t2s.py (teacher-forced logits and sampled tokens
with injected uniform noise).  Restated: the sampling branch (no beam search, no speculative decoding, B = 1 or equal-length
batches), with classifier-free guidance (cond_scale > 1, text2semantic.py:780-792: a second decode with the context masked out,
i.e. cross-attention over the learned null key / value only, its own cache) for one-output models - the reference's two-output
guidance feeds the full-width hidden state to the half-width logit head and cannot run.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def t2s_dims(sd: SD) -> dict:
    dim = sd["token_emb.text.weight"].shape[1]
    dim_t = sd["start_token.speech"].shape[0]
    emb = sd["semantic_token_emb.weight"].shape[1]
    heads = sd["target_transformer.layers.0.1.null_kv"].shape[1]
    depth = lambda pre: len({k.split(".")[2] for k in sd if k.startswith(pre + ".layers.")})
    return dict(dim=dim, dim_target=dim_t, dim_emb=emb, two_output=(emb * 2 == dim_t), heads=heads, dim_head=64,
                source_depth=depth("source_transformer"), target_depth=depth("target_transformer"),
                vocab=sd["semantic_token_emb.weight"].shape[0], eos_id=sd["semantic_token_emb.weight"].shape[0] - 1,
                text_eos_id=sd["token_emb.text.weight"].shape[0] - 1)


def rmsnorm(x, gamma):
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * gamma


def rotate_interleaved(t, positions, freqs):
    """t [..., n, 64]; positions [n] float; pairs (2i, 2i+1) rotate by positions * freqs[i]."""
    ang = positions[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1)
    x = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def attention(sd: SD, p: str, x, heads: int, freqs=None, context=None, mask=None, causal=False, cache=None):
    """One Attention.forward.  Returns (out, new_cache) with new_cache = (k, v) UN-rotated, [B, H, n, 64]."""
    B = x.shape[0]
    xn = rmsnorm(x, sd[p + ".norm.gamma"])
    ctx = xn if context is None else context
    q = (xn @ sd[p + ".to_q.0.weight"].T).reshape(B, -1, heads, 64).transpose(1, 2)
    kv = ctx @ sd[p + ".to_kv.0.weight"].T
    k, v = [t.reshape(B, -1, heads, 64).transpose(1, 2) for t in kv.chunk(2, dim=-1)]
    if cache is not None:
        k = torch.cat((cache[0], k), dim=-2)
        v = torch.cat((cache[1], v), dim=-2)
    new_cache = (k, v)
    if freqs is not None:
        n_q, n_k = q.shape[-2], k.shape[-2]
        pos = torch.arange(n_k, dtype=torch.float32)
        q = rotate_interleaved(q, pos[n_k - n_q:], freqs)
        k = rotate_interleaved(k, pos, freqs)
    if (p + ".null_kv") in sd:
        nk, nv = sd[p + ".null_kv"]
        k = torch.cat((nk[None].expand(B, -1, -1, -1), k), dim=-2)
        v = torch.cat((nv[None].expand(B, -1, -1, -1), v), dim=-2)
        if mask is not None:
            mask = F.pad(mask, (1, 0), value=True)
    sim = (q @ k.transpose(-1, -2)) * (64 ** -0.5)
    neg = -torch.finfo(sim.dtype).max
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, None, :], neg)
    if causal:
        i, j = sim.shape[-2:]
        n = max(i, j)
        cm = torch.ones((n, n), dtype=torch.bool).triu(1)[-i:, :]
        sim = sim.masked_fill(cm, neg)
    out = sim.softmax(dim=-1) @ v
    out = out.transpose(1, 2).reshape(B, -1, heads * 64)
    return out @ sd[p + ".to_out.weight"].T, new_cache


def feedforward(sd: SD, p: str, x):
    h = rmsnorm(x, sd[p + ".0.gamma"]) @ sd[p + ".1.weight"].T + sd[p + ".1.bias"]
    a, gate = h.chunk(2, dim=-1)
    return (F.gelu(gate) * a) @ sd[p + ".4.weight"].T + sd[p + ".4.bias"]


def transformer(sd: SD, pre: str, x, d: dict, depth: int, mask=None, context=None, context_mask=None, causal=False,
                cache: Optional[List] = None):
    """Transformer.forward with return_cache semantics: x holds ALL positions; with a cache only the new ones run."""
    freqs = sd[pre + ".layers.0.0.rotary_emb.freqs"]
    if cache is not None:
        x = x[:, cache[0][0].shape[-2]:]
    new_cache = []
    for i in range(depth):
        p = f"{pre}.layers.{i}"
        a, kv = attention(sd, p + ".0", x, d["heads"], freqs=freqs, mask=mask, causal=causal,
                          cache=None if cache is None else cache[i])
        x = a + x
        new_cache.append(kv)
        if context is not None:
            c, _ = attention(sd, p + ".1", x, d["heads"], context=context, mask=context_mask)
            x = c + x
        x = feedforward(sd, p + ".2", x) + x
    return rmsnorm(x, sd[pre + ".final_norm.gamma"]), new_cache


def set_eos_id(t, eos_id: int, pad_id: int):
    idx = ((t == pad_id).cumsum(dim=-1) == 0).sum(dim=-1, keepdim=True).long()
    t = F.pad(t, (0, 1), value=pad_id)
    t[torch.arange(t.shape[0])[:, None], idx] = eos_id
    return t


def mask_after_eos(target, eos_id: int, pad_id: int):
    m = (target == eos_id).cumsum(dim=-1) > 0
    m = F.pad(m, (1, -1), value=False)
    return target.masked_fill(m, pad_id)


def encode(sd: SD, source_ids: torch.Tensor):
    """text2semantic.py:716-741: append the text eos, mask = (id != 0), embed, source transformer."""
    d = t2s_dims(sd)
    src = set_eos_id(source_ids.clone(), d["text_eos_id"], 0)
    mask = src != 0
    emb = sd["token_emb.text.weight"][src]
    enc, _ = transformer(sd, "source_transformer", emb, d, d["source_depth"], mask=mask)
    return enc, mask


def top_k_filter(logits, thres: float = 0.1):
    k = math.ceil(thres * logits.shape[-1])
    val, ind = torch.topk(logits, k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(-1, ind, val)
    return out


def gumbel_from_uniform(u):
    log = lambda t: torch.log(t.clamp(min=1e-20))
    return -log(-log(u))


def generate(sd: SD, source_ids: torch.Tensor, uniforms: torch.Tensor, max_length: int = 2048, temperature: float = 1.0,
             forced: Optional[torch.Tensor] = None, on_step=None, cond_scale: float = 1.0):
    """Sampling branch of TextToSemantic.generate + TextToSemanticWrapper.sample.
    uniforms [max_length, S, B, V]: the U(0,1) draws the reference takes from torch's RNG (S = 2 for two_output, in
    the order stream 1 then stream 2 each step).  forced [B, S, L]: teacher forcing (tokens appended instead of the
    samples; the loop runs L steps).  Returns dict(tokens=flat target[target_mask] as the wrapper returns it,
    streams=[B, S, L] raw targets, logits=[L, S, B, V] pre-filter logits)."""
    d = t2s_dims(sd)
    S = 2 if d["two_output"] else 1
    enc, smask = encode(sd, source_ids)
    B = source_ids.shape[0]
    E = sd["semantic_token_emb.weight"]
    targets = [torch.empty((B, 0), dtype=torch.long) for _ in range(S)]
    start = sd["start_token.speech"][None, None, :].expand(B, 1, -1)
    cache = None
    null_cache = None
    assert cond_scale >= 1.0 and (cond_scale == 1.0 or S == 1), "guidance: one-output models only (see the module docstring)"
    all_logits = []
    steps = max_length if forced is None else forced.shape[-1]
    for t in range(steps):
        temb = torch.cat([E[tt] for tt in targets], dim=-1)
        temb = torch.cat((start, temb), dim=1)
        att, cache = transformer(sd, "target_transformer", temb, d, d["target_depth"], context=enc, context_mask=smask,
                                 causal=True, cache=cache)
        if cond_scale > 1.0:                   # text2semantic.py:780-792: the same decode with every context position masked
            att_n, null_cache = transformer(sd, "target_transformer", temb, d, d["target_depth"], context=enc,
                                            context_mask=torch.zeros_like(smask), causal=True, cache=null_cache)
            null_logits = (att_n @ E.T)[:, -1]
        half = att.shape[-1] // S
        step_logits, done = [], []
        for s in range(S):
            logits = (att[..., s * half:(s + 1) * half] @ E.T)[:, -1]
            if cond_scale > 1.0:
                logits = null_logits + (logits - null_logits) * cond_scale
            step_logits.append(logits)
            if forced is not None:
                sampled = forced[:, s, t]
            else:
                f = top_k_filter(logits)
                if on_step is not None:              # fixture generation only: lets the caller edit uniforms[t, s]
                    on_step(t, s, f)
                sampled = ((f / max(temperature, 1e-10)) + gumbel_from_uniform(uniforms[t, s])).argmax(dim=-1)
            targets[s] = torch.cat((targets[s], sampled[:, None]), dim=1)
            done.append(bool((targets[s] == d["eos_id"]).any(dim=-1).all()))
        all_logits.append(torch.stack(step_logits))
        if forced is None and any(done):         # one stream: stop on its eos; two streams: stop when EITHER has one
            break
    raw = torch.stack(targets, dim=1)
    if forced is None:
        targets = [mask_after_eos(tt, d["eos_id"], -1) for tt in targets]
    flat = torch.cat(targets, dim=1)
    return dict(tokens=flat[flat != -1], streams=raw, logits=torch.stack(all_logits))
