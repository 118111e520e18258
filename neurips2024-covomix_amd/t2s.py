"""text2semantic (CoSingle / CoMix) on the GPU - SURVEY.md section 8f row N1.

Host mirror of the reference's `TextToSemantic.generate` sampling branch + `TextToSemanticWrapper.sample`
(covomix/covomix_model/text2semantic.py:662-848, :1237-1251), the call `CoVoMixModel.synthesis_sample_text2semantic`
forwards to (covomix/conditional_model.py:313-321).  Only what the generation scripts reach is built: cond_scale == 1
(the reference asserts on anything else with its default cond_drop_prob = 0), no beam / speculative decoding, batch 1.

  encoder  (source transformer, once per utterance): the full-sequence kernels of the acoustic path - fp32 GEMM with
           the RoPE epilogue, flash attention, RMSNorm - plus a GEGLU kernel;
  decoder  (one token per step): csrc/t2s_decode.hip - cvx_t2s_decode_steps, 34 launches per step replayed from a HIP
           graph of CHUNK steps.  (Two single-launch persistent forms with grid barriers were built in round 4 and measured
           2.5-2.7x slower - a grid barrier with the L2 write-back / invalidate that cross-XCD visibility needs costs 4-7 us
           against 1.2-1.5 us for a dependent kernel boundary; removed in round 5, numbers in HISTORY.md.)  The host only
           looks at the eos flags between chunks.  `generate_batch` advances up to MAX_BATCH utterances together (the
           reference decodes them one by one): a token step is bound by streaming the decoder weights, which a batch
           shares, and the per-utterance arithmetic does not depend on the batch size - the tokens are bit-identical to
           the one-by-one decode.  The decode is a latency chain that needs a handful of CUs: pipeline.py runs it on a
           CU-masked side stream UNDER the acoustic solve of the previous batch.

The reference's rotary embedding rotates interleaved pairs (2i, 2i+1) (rotary_embedding_torch.py:25-41); the kernels
rotate half-split pairs (i, i+32).  Permuting the rows of to_q and to_k inside every head (the same permutation on
both, so q.k is unchanged) maps one onto the other - done once here at load time.
"""
import ctypes as C
import math
import os
from typing import Dict, Optional

import torch

from . import _lib, ops

PAD_ID = -1                    # semantic_pad_id (conditional_model.py:126)
TOP_K_THRES = 0.1              # top_k default (text2semantic.py:126)
CHUNK = 16                     # token steps per graph replay / host check
MAX_BATCH = 8                  # utterances per decode step (kernel limit)


def _dims(sd: Dict[str, torch.Tensor]) -> dict:
    dim = sd["token_emb.text.weight"].shape[1]
    dim_t = sd["start_token.speech"].shape[0]
    emb = sd["semantic_token_emb.weight"].shape[1]
    heads = sd["target_transformer.layers.0.1.null_kv"].shape[1]
    if sd["target_transformer.layers.0.1.null_kv"].shape[-1] != 64:
        raise ValueError("only dim_head == 64 is supported (reference default)")
    if emb not in (dim_t, dim_t // 2):
        raise ValueError("semantic embedding width must be the target width (one output) or half of it (two outputs)")
    depth = lambda pre: len({k.split(".")[2] for k in sd if k.startswith(pre + ".layers.")})
    ff = lambda pre: sd[pre + ".layers.0.2.4.weight"].shape[1]
    return dict(dim=dim, dim_target=dim_t, dim_emb=emb, streams=dim_t // emb, heads=heads, inner=heads * 64,
                source_depth=depth("source_transformer"), target_depth=depth("target_transformer"),
                vocab=sd["semantic_token_emb.weight"].shape[0], ff_src=ff("source_transformer"), ff_tgt=ff("target_transformer"),
                text_eos=sd["token_emb.text.weight"].shape[0] - 1)


def _half_split_rows(w: torch.Tensor, heads: int) -> torch.Tensor:
    """Rows of a [heads*64, K] projection reordered (0,2,..,62,1,3,..,63) inside every head."""
    perm = torch.cat((torch.arange(0, 64, 2), torch.arange(1, 64, 2))).to(w.device)
    return w.reshape(heads, 64, -1)[:, perm, :].reshape(heads * 64, -1).contiguous()


def _pad_cols(w: torch.Tensor, mult: int = 4) -> torch.Tensor:
    k = w.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros(w.shape[0], kp, dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


class TextToSemanticDecoder:
    """Device-resident packed weights of one TextToSemantic network + encode / generate."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device: torch.device, max_length: int = 2048, max_source: int = 1024):
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in state_dict.items()}
        self.device = device
        self.d = d = _dims(sd)
        self.max_length, self.max_source = int(max_length), int(max_source)
        H, I = d["heads"], d["inner"]
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)

        def attn_pack(p, self_attn):
            wq, wkv = sd[p + ".to_q.0.weight"], sd[p + ".to_kv.0.weight"]
            wk, wv = wkv[:I], wkv[I:]
            if self_attn:
                return torch.cat((_half_split_rows(wq, H), _half_split_rows(wk, H), wv), dim=0).contiguous()
            return wq.contiguous(), wkv.contiguous()

        self.emb_text = sd["token_emb.text.weight"]
        self.emb = sd["semantic_token_emb.weight"]
        self.start = sd["start_token.speech"]
        # ---- encoder
        self.enc = []
        for i in range(d["source_depth"]):
            p = f"source_transformer.layers.{i}"
            self.enc.append(dict(gamma_a=sd[p + ".0.norm.gamma"], wqkv=attn_pack(p + ".0", True), wo=sd[p + ".0.to_out.weight"],
                                 gamma_f=sd[p + ".2.0.gamma"], w1=sd[p + ".2.1.weight"], b1=sd[p + ".2.1.bias"],
                                 w2=_pad_cols(sd[p + ".2.4.weight"]), b2=sd[p + ".2.4.bias"]))
        self.enc_final = sd["source_transformer.final_norm.gamma"]
        self.freqs_src = sd["source_transformer.layers.0.0.rotary_emb.freqs"]
        # ---- decoder
        self.Fp = (d["ff_tgt"] + 3) // 4 * 4
        self.dec = []
        for i in range(d["target_depth"]):
            p = f"target_transformer.layers.{i}"
            wq_c, wkv_c = attn_pack(p + ".1", False)
            nkv = sd[p + ".1.null_kv"]                                    # [2, H, 1, 64]
            self.dec.append(dict(gamma_s=sd[p + ".0.norm.gamma"], wqkv_s=attn_pack(p + ".0", True), wo_s=sd[p + ".0.to_out.weight"],
                                 gamma_c=sd[p + ".1.norm.gamma"], wq_c=wq_c, wkv_c=wkv_c, wo_c=sd[p + ".1.to_out.weight"],
                                 null=torch.cat((nkv[0].reshape(I), nkv[1].reshape(I))).contiguous(),
                                 gamma_f=sd[p + ".2.0.gamma"], w1=sd[p + ".2.1.weight"], b1=sd[p + ".2.1.bias"],
                                 w2=_pad_cols(sd[p + ".2.4.weight"]), b2=sd[p + ".2.4.bias"],
                                 kv_c=f32(MAX_BATCH, self.max_source + 2, 2 * I), k_cache=f32(MAX_BATCH, self.max_length, I),
                                 v_cache=f32(MAX_BATCH, self.max_length, I)))
        self.dec_final = sd["target_transformer.final_norm.gamma"]
        pos = torch.arange(self.max_length, device=device, dtype=torch.float32)
        ang = pos[:, None] * sd["target_transformer.layers.0.0.rotary_emb.freqs"][None, :]
        self.rope = (ang.cos().contiguous(), ang.sin().contiguous())
        S, V = d["streams"], d["vocab"]
        self.top_k = math.ceil(TOP_K_THRES * V)
        self.buf = dict(x=f32(MAX_BATCH, d["dim_target"]), q=f32(MAX_BATCH, I), att=f32(MAX_BATCH, I), h=f32(MAX_BATCH, self.Fp),
                        logits=f32(MAX_BATCH, S, V), uniforms=f32(self.max_length * MAX_BATCH * S * V),
                        tokens=torch.zeros(MAX_BATCH, S, self.max_length, dtype=torch.int64, device=device))
        self.buf["state"] = torch.zeros(MAX_BATCH, 4, dtype=torch.int32, device=device)      # per utterance: pos, done, length, context rows
        self._layers = (_lib.T2SLayer * d["target_depth"])()
        for i, L in enumerate(self.dec):
            for name in ("gamma_s", "wqkv_s", "wo_s", "gamma_c", "wq_c", "wo_c", "kv_c", "gamma_f", "w1", "b1", "w2", "b2",
                         "k_cache", "v_cache"):
                setattr(self._layers[i], name, L[name].data_ptr())
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self._cap = None                                   # capture stream of the decode graphs
        self._stage = None                                 # staging copies of the state record (two in flight), _decode_chunks

    # ------------------------------------------------------------------ encoder (text2semantic.py:716-741)
    def encode(self, source_ids: torch.Tensor) -> torch.Tensor:
        d = self.d
        ids = source_ids.reshape(-1).to(self.device, torch.int64)
        if bool((ids == 0).any()):
            raise NotImplementedError("padded text batches (id 0) are not supported: one un-padded utterance per call")
        if ids.numel() + 1 > self.max_source:
            raise ValueError(f"text of {ids.numel()} tokens exceeds max_source = {self.max_source}")
        src = torch.cat((ids, torch.tensor([d["text_eos"]], device=self.device)))          # set_eos_id, no padding
        n, H, I, D = src.numel(), d["heads"], d["inner"], d["dim"]
        x = self.emb_text.index_select(0, src).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        normed, qkv, att = f(n, D), f(n, 3 * I), f(n, I)
        F_ = d["ff_src"]
        Fp = (F_ + 3) // 4 * 4
        h2, hg = f(n, 2 * F_), f(n, Fp)
        pos = torch.arange(n, device=self.device, dtype=torch.float32)
        ang = pos[:, None] * self.freqs_src[None, :]
        rope = (ang.cos().contiguous(), ang.sin().contiguous())
        for L in self.enc:
            ops.adarmsnorm(x, L["gamma_a"], None, normed)
            ops.gemm(normed, L["wqkv"], qkv, rope=rope, rope_cols=2 * I)
            ops.attention(qkv, att, 1, n, H, 64 ** -0.5)
            ops.gemm(att, L["wo"], x, residual=x)
            ops.adarmsnorm(x, L["gamma_f"], None, normed)
            ops.gemm(normed, L["w1"], h2, bias=L["b1"])
            ops.geglu(h2, hg, F_)
            ops.gemm(hg, L["w2"], x, bias=L["b2"], residual=x)
        enc = f(n, D)
        ops.adarmsnorm(x, self.enc_final, None, enc)
        return enc

    # ------------------------------------------------------------------ decoder
    def _descriptor(self, temperature: float, batch: int = 1, cfg_scale: float = 1.0) -> "_lib.T2SDecoder":
        d, b = self.d, self.buf
        dec = _lib.T2SDecoder()
        dec.batch, dec.ctx_rows = batch, self.max_source + 2
        dec.cfg_scale = float(cfg_scale)
        dec.dim, dec.inner, dec.heads = d["dim_target"], d["inner"], d["heads"]
        dec.ff_inner, dec.ff_inner_pad, dec.depth = d["ff_tgt"], self.Fp, d["target_depth"]
        dec.streams, dec.vocab, dec.dim_emb = d["streams"], d["vocab"], d["dim_emb"]
        dec.n_ctx, dec.max_len, dec.top_k, dec.temperature = 0, self.max_length, self.top_k, float(temperature)
        dec.layers = C.cast(self._layers, C.POINTER(_lib.T2SLayer))
        dec.final_gamma, dec.emb = self.dec_final.data_ptr(), self.emb.data_ptr()
        dec.rope_cos, dec.rope_sin = self.rope[0].data_ptr(), self.rope[1].data_ptr()
        for n in ("uniforms", "x", "q", "att", "h", "logits", "tokens", "state"):
            setattr(dec, n, b[n].data_ptr())
        return dec

    def _steps(self, temperature: float, batch: int, n: int, cfg_scale: float = 1.0) -> None:
        """n token steps on the current stream without a graph."""
        _lib.check(_lib.load().cvx_t2s_decode_steps(C.byref(self._descriptor(temperature, batch, cfg_scale)), n,
                                                    torch.cuda.current_stream().cuda_stream), "cvx_t2s_decode_steps")

    def _read_state(self, nb: int) -> list:
        """state rows of the first nb utterances (synchronises the current stream)."""
        return self.buf["state"].tolist()[:nb]

    def _decode_chunks(self, temperature: float, nb: int, max_len: int, cfg_scale: float, watch, ignore_eos: bool = False) -> list:
        """Graph-replayed chunks of CHUNK token steps until every utterance slot in `watch` has sampled its eos (or max_len steps).
        The host looks at the eos flags ONE CHUNK BEHIND the device: the state record of chunk i is copied to a device-side staging
        record between the replays of chunks i and i + 1 and read through a helper stream while chunk i + 1 runs - the decode
        chain never waits for a host round trip (nor for a host thread that is waiting for the interpreter lock while another thread
        drives the acoustic solve, pipeline.py); the price is at most one chunk decoded past the last eos (masked afterwards like every
        token behind an eos).  Returns the final state rows."""
        if self._stage is None:
            self._stage = [torch.empty_like(self.buf["state"]) for _ in range(2)]
            self._pin = [torch.empty(MAX_BATCH, 4, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._stage_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._pin_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._helper = torch.cuda.Stream(device=self.device)
        # How the record reaches the host.  On an ordinary stream: a non_blocking copy into pinned memory on the decode stream itself.
        # On a CU-masked stream of ops.CUPartition that form is NOT used: torch's pinned-memory allocator remembers the stream of such a
        # copy, and a masked stream destroyed at exit before the block is freed takes the process down (tools/cu_mask_exit_probe.py) -
        # there the record is copied device-to-device on the decode stream and a plain helper stream brings it to the host.  (The
        # helper form on the legacy null stream measured 3.4x slower per decode - 941 vs 274 ms - in one process layout and not in another:
        # it is confined to the streams that need it.)
        via_helper = ops.is_partition_stream()
        steps, i, pending = 0, 0, None
        while steps < max_len:
            self._run_chunk(temperature, nb, cfg_scale)
            steps += CHUNK
            k = i & 1
            if via_helper:
                self._stage[k].copy_(self.buf["state"])                 # (device to device, behind chunk i on this stream)
                self._stage_ev[k].record()
                with torch.cuda.stream(self._helper):                   # the helper waits for chunk i only
                    self._helper.wait_event(self._stage_ev[k])
                    self._pin[k].copy_(self._stage[k], non_blocking=True)
                    self._pin_ev[k].record()
            else:
                self._pin[k].copy_(self.buf["state"], non_blocking=True)
                self._pin_ev[k].record()
            if pending is not None:
                self._pin_ev[pending].synchronize()
                st = self._pin[pending].tolist()
                if all(st[r][1] for r in watch) and not ignore_eos:
                    break
            pending = k
            i += 1
        return self._read_state(nb)

    def _run_chunk(self, temperature: float, batch: int = 1, cfg_scale: float = 1.0) -> None:
        """CHUNK token steps on the current stream (a graph replay of the per-launch path)."""
        def launch():
            _lib.check(_lib.load().cvx_t2s_decode_steps(C.byref(self._descriptor(temperature, batch, cfg_scale)), CHUNK,
                                                        torch.cuda.current_stream().cuda_stream), "cvx_t2s_decode_steps")
        if os.environ.get("CVX_GRAPH", "1") != "1":
            launch()
            return
        key = (temperature, batch, cfg_scale, ops.stream_cus())   # (the kernels' shape follows the CUs the stream owns)
        g = self._graphs.get(key)
        if g is None:
            saved = {k: v.clone() for k, v in self.buf.items()}
            caches = [(L["k_cache"].clone(), L["v_cache"].clone()) for L in self.dec]
            launch()                                       # warm-up outside capture (module load, attributes)
            cur = torch.cuda.current_stream()
            if self._cap is None:
                self._cap = torch.cuda.Stream(device=self.device)
            ops.saturation_share(cur, self._cap)           # (the capture stream belongs to this call: flag and CU count of `cur`)
            with ops.CAPTURE_GATE.exclusive():             # (no other entry point of the package syncs / copies meanwhile)
                cur.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self._cap, capture_error_mode="thread_local"):
                    launch()
            for k, v in saved.items():                     # capture does not execute, the warm-up did: restore
                self.buf[k].copy_(v)
            for L, (kc, vc) in zip(self.dec, caches):
                L["k_cache"].copy_(kc); L["v_cache"].copy_(vc)
            if len(self._graphs) >= 6:
                self._graphs.clear()
            self._graphs[key] = g
        g.replay()

    @ops.gated
    @torch.no_grad()          # (not inference_mode: tensors torch creates lazily during the first graph capture,
                              #  e.g. the generator's graph-safe state, would become inference tensors)
    def generate_batch(self, sources, uniforms=None, max_length: Optional[int] = None, temperature: float = 1.0,
                       generator: Optional[torch.Generator] = None, collect_logits: bool = False, cond_scale: float = 1.0,
                       ignore_eos: bool = False):
        """Decode up to MAX_BATCH utterances together.  sources: list of [n] / [1, n] id tensors; uniforms: optional list
        of [steps, streams, vocab] tensors (one per utterance).  Returns a list of (flat tokens, streams[, logits])
        tuples, each exactly what `generate` returns for that utterance alone.
        ignore_eos (benchmarks: a fixed amount of work): decode max_length steps whatever is sampled; `streams` then holds all of them.
        cond_scale > 1: classifier-free guidance (text2semantic.py:780-792; one-output models, up to MAX_BATCH / 2 utterances):
        every utterance takes two decode slots - the text context and the context masked out (cross-attention then sees the
        learned null key / value only) - and each step samples from null + (cond - null) * cond_scale; logits returned under
        collect_logits are the COMBINED ones, null + (cond - null) * cond_scale (what the reference filters and samples from)."""
        d, b = self.d, self.buf
        S, V = d["streams"], d["vocab"]
        cfg = float(cond_scale) > 1.0
        if cfg:
            if S != 1:
                raise NotImplementedError("guidance (cond_scale > 1) on a two-output model: the reference feeds the full-width hidden "
                                          "state to the half-width logit head there (text2semantic.py:783-785) and cannot run")
            return self._generate_guided(sources, uniforms, max_length, temperature, generator, collect_logits, float(cond_scale))
        nb = len(sources)
        if not 1 <= nb <= MAX_BATCH:
            raise ValueError(f"1..{MAX_BATCH} utterances per decode batch, got {nb}")
        max_len = min(int(max_length or self.max_length), self.max_length)
        if uniforms is not None:
            us = [u.to(self.device, torch.float32).reshape(u.shape[0], S, V) for u in uniforms]
            max_len = min([max_len] + [u.shape[0] for u in us])
        ctx = []
        for i, src in enumerate(sources):
            if src.ndim == 2 and src.shape[0] != 1:
                raise NotImplementedError("one utterance per entry (the generation scripts run batch 1)")
            enc = self.encode(src)
            n = enc.shape[0]
            ctx.append(n + 1)
            for L in self.dec:                                          # context k/v once: [null | to_kv(enc)]
                L["kv_c"][i, 0].copy_(L["null"])
                ops.gemm(enc, L["wkv_c"], L["kv_c"][i, 1:n + 1])
        uview = b["uniforms"][: max_len * nb * S * V].view(max_len, nb, S, V)
        if uniforms is None:
            uview.copy_(torch.rand(max_len, nb, S, V, device=self.device, generator=generator))
        else:
            for i, u in enumerate(us):
                uview[:, i].copy_(u[:max_len])
        b["x"][:nb].copy_(self.start[None, :].expand(nb, -1))
        b["state"].copy_(torch.tensor([[0, 0, 0, ctx[i] if i < nb else 1] for i in range(MAX_BATCH)], dtype=torch.int32))
        logits = []
        if collect_logits:                                              # (tests: one step at a time without a graph)
            for _ in range(max_len):
                self._steps(float(temperature), nb, 1)
                logits.append(b["logits"][:nb].clone())
                st = self._read_state(nb)
                if all(row[1] for row in st) and not ignore_eos:
                    break
        else:
            st = self._decode_chunks(float(temperature), nb, max_len, 1.0, range(nb), ignore_eos)
        eos = V - 1
        out = []
        for i in range(nb):
            length = min(st[i][2] if st[i][1] and st[i][2] <= max_len and not ignore_eos else max_len, max_len)
            streams = b["tokens"][i, :, :length].clone()
            after = (streams == eos).cumsum(dim=-1) > 0                  # mask_after_eos (text2semantic.py:73-76)
            after = torch.nn.functional.pad(after, (1, -1), value=False)
            flat = streams.masked_fill(after, PAD_ID).reshape(-1)
            item = (flat[flat != PAD_ID], streams)
            if collect_logits:
                item = item + (torch.stack([lg[i] for lg in logits])[:length],)
            out.append(item)
        return out

    def _generate_guided(self, sources, uniforms, max_length, temperature, generator, collect_logits, cond_scale):
        """generate_batch with cond_scale > 1: slots 2u (text context) / 2u + 1 (null context) per utterance u."""
        d, b = self.d, self.buf
        V, nu = d["vocab"], len(sources)
        nb = 2 * nu
        if not 1 <= nu <= MAX_BATCH // 2:
            raise ValueError(f"1..{MAX_BATCH // 2} utterances per guided decode batch, got {nu}")
        max_len = min(int(max_length or self.max_length), self.max_length)
        us = None
        if uniforms is not None:
            us = [u.to(self.device, torch.float32).reshape(u.shape[0], 1, V) for u in uniforms]
            max_len = min([max_len] + [u.shape[0] for u in us])
        ctx = []
        for u_, src in enumerate(sources):
            if src.ndim == 2 and src.shape[0] != 1:
                raise NotImplementedError("one utterance per entry (the generation scripts run batch 1)")
            enc = self.encode(src)
            n = enc.shape[0]
            ctx += [n + 1, 1]                                            # the null slot: row 0 (null k/v) only = every context key masked
            for L in self.dec:
                L["kv_c"][2 * u_, 0].copy_(L["null"])
                ops.gemm(enc, L["wkv_c"], L["kv_c"][2 * u_, 1:n + 1])
                L["kv_c"][2 * u_ + 1, 0].copy_(L["null"])
        uview = b["uniforms"][: max_len * nb * V].view(max_len, nb, 1, V)
        if us is None:
            uview[:, 0::2].copy_(torch.rand(max_len, nu, 1, V, device=self.device, generator=generator))
        else:
            for u_, u in enumerate(us):
                uview[:, 2 * u_].copy_(u[:max_len])
        b["x"][:nb].copy_(self.start[None, :].expand(nb, -1))
        b["state"].copy_(torch.tensor([[0, 0, 0, ctx[i] if i < nb else 1] for i in range(MAX_BATCH)], dtype=torch.int32))
        logits = []
        if collect_logits:
            for _ in range(max_len):
                self._steps(float(temperature), nb, 1, cond_scale)
                lg = b["logits"][:nb].clone()
                logits.append(lg[1::2] + (lg[0::2] - lg[1::2]) * cond_scale)
                st = self._read_state(nb)
                if all(st[2 * u_][1] for u_ in range(nu)):
                    break
        else:
            st = self._decode_chunks(float(temperature), nb, max_len, cond_scale, [2 * u_ for u_ in range(nu)])
        eos, out = V - 1, []
        for u_ in range(nu):
            i = 2 * u_
            length = min(st[i][2] if st[i][1] and st[i][2] <= max_len else max_len, max_len)
            streams = b["tokens"][i, :, :length].clone()
            after = (streams == eos).cumsum(dim=-1) > 0
            after = torch.nn.functional.pad(after, (1, -1), value=False)
            flat = streams.masked_fill(after, PAD_ID).reshape(-1)
            item = (flat[flat != PAD_ID], streams)
            if collect_logits:
                item = item + (torch.stack([lg[u_] for lg in logits])[:length],)
            out.append(item)
        return out

    @ops.gated
    def generate(self, source_ids: torch.Tensor, uniforms: Optional[torch.Tensor] = None, max_length: Optional[int] = None,
                 temperature: float = 1.0, generator: Optional[torch.Generator] = None, return_streams: bool = False,
                 collect_logits: bool = False, cond_scale: float = 1.0):
        """== TextToSemanticWrapper.sample(grapheme_token_ids): flat int64 tensor, stream 1 then stream 2 (two-output
        models), each cut after its eos.  uniforms [steps, streams, vocab] (or [steps, streams, 1, vocab]) replaces
        the random draws of gumbel_noise (text2semantic.py:108-110); default: torch.rand from `generator`.
        collect_logits (tests): step one token at a time without a graph and also return the pre-filter logits
        [steps, streams, vocab]."""
        if source_ids.ndim == 2 and source_ids.shape[0] != 1:
            raise NotImplementedError("one utterance per call (the generation scripts run batch 1); see generate_batch")
        res = self.generate_batch([source_ids], None if uniforms is None else [uniforms], max_length, temperature, generator,
                                  collect_logits, cond_scale)[0]
        if collect_logits:
            return res
        return res if return_streams else res[0]
