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
           looks at the eos flags between chunks.  `generate_batch` advances up to MAX_BATCH (64) utterances together (the
           reference decodes them one by one): a token step is a latency chain of dependent launches whose time barely
           depends on the batch, and the per-utterance arithmetic does not depend on the batch size - the tokens are
           bit-identical to the one-by-one decode.  `generate_many` decodes ANY number of utterances through a fixed number
           of decode slots with CONTINUOUS BATCHING: an utterance that has sampled its eos (text2semantic.py:803-818) frees
           its slot, and the sampling kernel of that very step hands the slot the next pending utterance (device-side
           queue, no host round trip) - real dialogues end at different steps, and a lock-step batch would run half empty.

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
MAX_BATCH = 64                 # decode slots per step (kernel limit)
WINDOW = 256                   # utterances queued on the device at a time (generate_many)
SR = 8                         # int32 per slot / dialogue record (include/covomix_hip.h, cvx_t2s_decoder)


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
                                 w2=_pad_cols(sd[p + ".2.4.weight"]), b2=sd[p + ".2.4.bias"]))
        self.dec_final = sd["target_transformer.final_norm.gamma"]
        pos = torch.arange(self.max_length, device=device, dtype=torch.float32)
        ang = pos[:, None] * sd["target_transformer.layers.0.0.rotary_emb.freqs"][None, :]
        self.rope = (ang.cos().contiguous(), ang.sin().contiguous())
        self.top_k = math.ceil(TOP_K_THRES * d["vocab"])
        self.buf: Dict[str, torch.Tensor] = {}
        self._slots = self._dialogues = self._steps = 0   # capacities of the decode buffers (_ensure)
        self._gen = 0                                      # bumped when they are re-allocated (captured graphs hold their addresses)
        self._layers = (_lib.T2SLayer * d["target_depth"])()
        for i, L in enumerate(self.dec):
            for name in ("gamma_s", "wqkv_s", "wo_s", "gamma_c", "wq_c", "wo_c", "gamma_f", "w1", "b1", "w2", "b2"):
                setattr(self._layers[i], name, L[name].data_ptr())
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self._cap = None                                   # capture stream of the decode graphs
        self._stage = None                                 # staging copies of the state record (two in flight), _decode_chunks
        self._ensure(8, 8, 0)

    @staticmethod
    def _slot_capacity(n: int) -> int:
        """per-slot buffers hold whole kernel groups: 1 / 2 / 4 slots, or a multiple of 8 (include/covomix_hip.h)"""
        return n if n in (1, 2, 4) else (n + 7) // 8 * 8

    def _ensure(self, slots: int, dialogues: int, steps: int) -> None:
        """Decode buffers for `slots` decode slots (x, q, att, h, logits, slot records, the self-attention caches), `dialogues`
        utterances in flight or queued (context k/v, token rows, dialogue records) and `steps` uniform draws per dialogue.  They only
        grow; growing re-allocates (and drops the captured graphs, which hold the old addresses)."""
        d, dev = self.d, self.device
        S, V, I = d["streams"], d["vocab"], d["inner"]
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        slots = max(8, self._slot_capacity(slots))
        grown = False
        if slots > self._slots:
            self.buf.update(x=f32(slots, d["dim_target"]), q=f32(slots, I), att=f32(slots, I), h=f32(slots, self.Fp), logits=f32(slots, S, V),
                            state=torch.zeros(slots, SR, dtype=torch.int32, device=dev))
            for i, L in enumerate(self.dec):
                L["k_cache"], L["v_cache"] = f32(slots, self.max_length, I), f32(slots, self.max_length, I)
                self._layers[i].k_cache, self._layers[i].v_cache = L["k_cache"].data_ptr(), L["v_cache"].data_ptr()
            self._slots, grown = slots, True
        if dialogues > self._dialogues:
            dialogues = max(dialogues, 8)
            for i, L in enumerate(self.dec):
                L["kv_c"] = f32(dialogues, self.max_source + 2, 2 * I)
                self._layers[i].kv_c = L["kv_c"].data_ptr()
            self.buf.update(tokens=torch.zeros(dialogues, S, self.max_length, dtype=torch.int64, device=dev),
                            dialogues=torch.zeros(dialogues, SR, dtype=torch.int32, device=dev),
                            queue=torch.zeros(2, dtype=torch.int32, device=dev))
            self._dialogues, grown = dialogues, True
        if steps > self._steps or "uniforms" not in self.buf or self.buf["uniforms"].numel() < self._dialogues * self._steps * S * V:
            self._steps = max(self._steps, steps, 1)
            self.buf["uniforms"] = f32(self._dialogues * self._steps * S * V)
            grown = True
        if grown:
            self._graphs.clear()
            self._stage = None
            self._gen += 1

    # ------------------------------------------------------------------ encoder (text2semantic.py:716-741)
    def _source_rows(self, source_ids: torch.Tensor) -> torch.Tensor:
        if source_ids.ndim == 2 and source_ids.shape[0] != 1:
            raise NotImplementedError("one utterance per entry (the generation scripts run batch 1)")
        ids = source_ids.reshape(-1).to(torch.int64)
        if bool((ids == 0).any()):
            raise NotImplementedError("padded text batches (id 0) are not supported: one un-padded utterance per call")
        if ids.numel() + 1 > self.max_source:
            raise ValueError(f"text of {ids.numel()} tokens exceeds max_source = {self.max_source}")
        return torch.cat((ids.cpu(), torch.tensor([self.d["text_eos"]])))                    # set_eos_id, no padding

    def encode_many(self, sources):
        """The source transformer over SEVERAL texts as one packed batch (rows of text i: [cu[i], cu[i + 1]); attention and rotary
        positions per text): the reference encodes one utterance per call, and a 65-token text alone is 19 GEMM launches of 30 us
        each on a handful of CUs - 64 dialogues encoded one by one cost a quarter of their decode.  Row results do not depend on the
        other rows of the batch.  -> (encoder output [M, dim], ops.Ragged)"""
        d = self.d
        rows = [self._source_rows(s_) for s_ in sources]
        rg = ops.Ragged([r.numel() for r in rows], self.device)
        src = ops.h2d(torch.cat(rows), self.device)
        M, H, I, D = rg.M, d["heads"], d["inner"], d["dim"]
        x = self.emb_text.index_select(0, src).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        normed, qkv, att = f(M, D), f(M, 3 * I), f(M, I)
        F_ = d["ff_src"]
        Fp = (F_ + 3) // 4 * 4
        h2, hg = f(M, 2 * F_), f(M, Fp)
        ang = rg.positions()[:, None] * self.freqs_src[None, :]
        rope = (ang.cos().contiguous(), ang.sin().contiguous())          # per-row tables (rope_T = M)
        for L in self.enc:
            ops.adarmsnorm(x, L["gamma_a"], None, normed)
            ops.gemm(normed, L["wqkv"], qkv, rope=rope, rope_cols=2 * I)
            ops.attention(qkv, att, 1, M, H, 64 ** -0.5, ragged=rg)
            ops.gemm(att, L["wo"], x, residual=x)
            ops.adarmsnorm(x, L["gamma_f"], None, normed)
            ops.gemm(normed, L["w1"], h2, bias=L["b1"])
            ops.geglu(h2, hg, F_)
            ops.gemm(hg, L["w2"], x, bias=L["b2"], residual=x)
        enc = f(M, D)
        ops.adarmsnorm(x, self.enc_final, None, enc)
        return enc, rg

    def encode(self, source_ids: torch.Tensor) -> torch.Tensor:
        return self.encode_many([source_ids])[0]

    # ------------------------------------------------------------------ decoder
    def _descriptor(self, temperature: float, batch: int = 1, cfg_scale: float = 1.0, queue: bool = False) -> "_lib.T2SDecoder":
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
        dec.uniform_steps = self._steps
        dec.group_loop = int(os.environ.get("CVX_T2S_GROUP_LOOP", "0"))          # dev A/B
        dec.pairs_per_wave = int(os.environ.get("CVX_T2S_PPW", "0"))              # dev A/B
        if queue:
            dec.queue, dec.dialogues, dec.start = b["queue"].data_ptr(), b["dialogues"].data_ptr(), self.start.data_ptr()
        return dec

    def _run_steps(self, temperature: float, batch: int, n: int, cfg_scale: float = 1.0, queue: bool = False) -> None:
        """n token steps on the current stream without a graph."""
        _lib.check(_lib.load().cvx_t2s_decode_steps(C.byref(self._descriptor(temperature, batch, cfg_scale, queue)), n,
                                                    ops._stream()), "cvx_t2s_decode_steps")

    def _uniform_view(self, n: int) -> torch.Tensor:
        """[n, steps, streams, vocab] view of the uniform draws of the first n dialogues"""
        S, V = self.d["streams"], self.d["vocab"]
        return self.buf["uniforms"][: n * self._steps * S * V].view(n, self._steps, S, V)

    def _slot_records(self, ctx, limit: int = 0, flags: int = 0) -> torch.Tensor:
        """slot records [slots, SR] (CPU): slot b decodes dialogue b from position 0; slots past len(ctx) idle at max_length"""
        rows = [[0, 0, 0, ctx[i], i, limit, flags, 0] if i < len(ctx) else [self.max_length, 1, 0, 1, 0, 0, 0, 0] for i in range(self._slots)]
        return torch.tensor(rows, dtype=torch.int32)


    def _read_state(self, nb: int) -> list:
        """slot records of the first nb slots (synchronises the current stream)."""
        return self.buf["state"].tolist()[:nb]

    # ---- the host looks at device-side records ONE CHUNK BEHIND the device
    def _mirror_setup(self) -> None:
        if self._stage is None:
            rows = max(self._slots, self._dialogues)
            self._stage = [torch.empty(rows, SR, dtype=torch.int32, device=self.device) for _ in range(2)]
            self._pin = [torch.empty(rows, SR, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._stage_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._pin_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._helper = torch.cuda.Stream(device=self.device)

    def _mirror_push(self, k: int, src: torch.Tensor, via_helper: bool) -> None:
        """enqueue a copy of the records `src` [rows, SR] into pinned buffer k behind everything the current stream holds.
        On an ordinary stream: a non_blocking copy into pinned memory on the decode stream itself.  On a CU-masked stream of
        ops.CUPartition that form is NOT used: torch's pinned-memory allocator remembers the stream of such a copy, and a masked stream
        destroyed at exit before the block is freed takes the process down (tools/archive/cu_mask_exit_probe.py) - there the record is copied
        device-to-device on the decode stream and a plain helper stream brings it to the host."""
        rows = src.shape[0]
        if via_helper:
            self._stage[k][:rows].copy_(src)
            self._stage_ev[k].record()
            with torch.cuda.stream(self._helper):
                self._helper.wait_event(self._stage_ev[k])
                self._pin[k][:rows].copy_(self._stage[k][:rows], non_blocking=True)
                self._pin_ev[k].record()
        else:
            self._pin[k][:rows].copy_(src, non_blocking=True)
            self._pin_ev[k].record()

    def _mirror_pull(self, k: int, rows: int) -> list:
        self._pin_ev[k].synchronize()
        return self._pin[k][:rows].tolist()

    def _decode_chunks(self, temperature: float, nb: int, max_len: int, cfg_scale: float, watch, ignore_eos: bool = False) -> list:
        """Graph-replayed chunks of CHUNK token steps until every utterance slot in `watch` has sampled its eos (or max_len steps).
        The host looks at the eos flags ONE CHUNK BEHIND the device: the slot records of chunk i are copied between the replays of
        chunks i and i + 1 and read while chunk i + 1 runs - the decode chain never waits for a host round trip (nor for a host
        thread that is waiting for the interpreter lock while another thread drives the acoustic solve, pipeline.py); the price is
        at most one chunk decoded past the last eos (masked afterwards like every token behind an eos).  Returns the final records."""
        self._mirror_setup()
        via_helper = ops.is_partition_stream()
        steps, i, pending = 0, 0, None
        while steps < max_len:
            self._run_chunk(temperature, nb, cfg_scale)
            steps += CHUNK
            k = i & 1
            self._mirror_push(k, self.buf["state"][:nb], via_helper)
            if pending is not None:
                st = self._mirror_pull(pending, nb)
                if all(st[r][1] for r in watch) and not ignore_eos:
                    break
            pending = k
            i += 1
        if pending is not None:
            self._pin_ev[pending].synchronize()      # (the helper stream's last copy: the buffers are reused by the next call)
        return self._read_state(nb)

    def _graph(self, temperature: float, batch: int, cfg_scale: float = 1.0, queue: bool = False):
        """The captured graph of CHUNK token steps for this (batch, stream CU count, mode); captured on first use.  Capturing runs the
        steps once outside the capture (module load, kernel attributes): callers get their graph BEFORE they set up the decode state -
        the warm-up runs on idle slot records (position max_length: the sampling kernel returns at once, every other kernel clamps)."""
        key = (temperature, batch, cfg_scale, ops.stream_cus(), queue, self._gen)   # (the kernels' shape follows the CUs the stream owns)
        g = self._graphs.get(key)
        if g is not None:
            return g

        def launch():
            _lib.check(_lib.load().cvx_t2s_decode_steps(C.byref(self._descriptor(temperature, batch, cfg_scale, queue)), CHUNK,
                                                        ops._stream()), "cvx_t2s_decode_steps")
        self.buf["state"].copy_(self._slot_records([]))
        launch()                                       # warm-up outside capture
        cur = torch.cuda.current_stream()
        if self._cap is None:
            self._cap = torch.cuda.Stream(device=self.device)
        ops.saturation_share(cur, self._cap)           # (the capture stream belongs to this call: flag and CU count of `cur`)
        with ops.CAPTURE_GATE.exclusive():             # (no other entry point of the package syncs / copies meanwhile)
            cur.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._cap, capture_error_mode="thread_local"):
                launch()
        if len(self._graphs) >= 8:
            self._graphs.clear()
        self._graphs[key] = g
        return g

    def _run_chunk(self, temperature: float, batch: int = 1, cfg_scale: float = 1.0, queue: bool = False) -> None:
        """CHUNK token steps on the current stream: a graph replay of the per-launch path (the graph must exist - `_graph` - unless
        CVX_GRAPH=0 asks for plain launches)."""
        if os.environ.get("CVX_GRAPH", "1") != "1":
            self._run_steps(temperature, batch, CHUNK, cfg_scale, queue)
            return
        self._graph(temperature, batch, cfg_scale, queue).replay()

    def _contexts(self, sources, rows=None) -> list:
        """encoder + the cross-attention k/v of the utterances into dialogue rows `rows` (default 0, 1, ...): [null | to_kv(enc)]
        -> context rows per utterance.  One packed encoder pass and one to_kv GEMM per decoder layer for all of them."""
        n = len(sources)
        rows = list(range(n)) if rows is None else list(rows)
        enc, rg = self.encode_many(sources)
        R = self.max_source + 2
        dst = torch.cat([torch.arange(t, dtype=torch.int64) + (rows[i] * R + 1) for i, t in enumerate(rg.lengths)])
        dst = ops.h2d(dst, self.device)
        first = ops.h2d(torch.tensor([r * R for r in rows], dtype=torch.int64), self.device)
        kv = torch.empty(rg.M, 2 * self.d["inner"], dtype=torch.float32, device=self.device)
        for L in self.dec:
            ops.gemm(enc, L["wkv_c"], kv)
            flat = L["kv_c"].view(-1, kv.shape[1])
            flat.index_copy_(0, dst, kv)
            flat.index_copy_(0, first, L["null"][None, :].expand(n, -1))
        return [t + 1 for t in rg.lengths]

    def _cut(self, j: int, length: int, logits=None):
        """(flat tokens, streams[, logits]) of dialogue row j after `length` steps: mask_after_eos (text2semantic.py:73-76)"""
        eos = self.d["vocab"] - 1
        streams = self.buf["tokens"][j, :, :length].clone()
        after = (streams == eos).cumsum(dim=-1) > 0
        after = torch.nn.functional.pad(after, (1, -1), value=False)
        flat = streams.masked_fill(after, PAD_ID).reshape(-1)
        item = (flat[flat != PAD_ID], streams)
        return item if logits is None else item + (logits[:length],)

    @ops.gated
    @torch.no_grad()          # (not inference_mode: tensors torch creates lazily during the first graph capture,
                              #  e.g. the generator's graph-safe state, would become inference tensors)
    def generate_batch(self, sources, uniforms=None, max_length: Optional[int] = None, temperature: float = 1.0,
                       generator: Optional[torch.Generator] = None, collect_logits: bool = False, cond_scale: float = 1.0,
                       ignore_eos: bool = False):
        """Decode up to MAX_BATCH utterances together IN LOCK STEP (all start at position 0; the batch runs until the last one has
        sampled its eos).  sources: list of [n] / [1, n] id tensors; uniforms: optional list of [steps, streams, vocab] tensors (one
        per utterance).  Returns a list of (flat tokens, streams[, logits]) tuples, each exactly what `generate` returns for that
        utterance alone.  (`generate_many`: any number of utterances through continuously refilled slots.)
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
        us = None
        if uniforms is not None:
            us = [u.to(self.device, torch.float32).reshape(u.shape[0], S, V) for u in uniforms]
            max_len = min([max_len] + [u.shape[0] for u in us])
        self._ensure(nb, nb, max_len)
        b = self.buf
        if not collect_logits and max_len > 0:
            self._graph(float(temperature), nb)
        ctx = self._contexts(sources)
        uview = self._uniform_view(nb)
        if us is None:            # (drawn step-major, as the [steps, batch, streams, vocab] buffer of earlier versions was: same seeds, same tokens)
            uview[:, :max_len].copy_(torch.rand(max_len, nb, S, V, device=self.device, generator=generator).permute(1, 0, 2, 3))
        else:
            for i, u in enumerate(us):
                uview[i, :max_len].copy_(u[:max_len])
        b["x"][:nb].copy_(self.start[None, :].expand(nb, -1))
        b["state"].copy_(self._slot_records(ctx))
        logits = []
        st = self._slot_records(ctx).tolist()[:nb]
        if collect_logits:                                              # (tests: one step at a time without a graph)
            for _ in range(max_len):
                self._run_steps(float(temperature), nb, 1)
                logits.append(b["logits"][:nb].clone())
                st = self._read_state(nb)
                if all(row[1] for row in st) and not ignore_eos:
                    break
        elif max_len > 0:
            st = self._decode_chunks(float(temperature), nb, max_len, 1.0, range(nb), ignore_eos)
        out = []
        for i in range(nb):
            length = min(st[i][2] if st[i][1] and st[i][2] <= max_len and not ignore_eos else max_len, max_len)
            out.append(self._cut(i, length, torch.stack([lg[i] for lg in logits]) if collect_logits and logits else None))
        return out

    def _generate_guided(self, sources, uniforms, max_length, temperature, generator, collect_logits, cond_scale):
        """generate_batch with cond_scale > 1: slots 2u (text context) / 2u + 1 (null context) per utterance u."""
        V, nu = self.d["vocab"], len(sources)
        nb = 2 * nu
        if not 1 <= nu <= MAX_BATCH // 2:
            raise ValueError(f"1..{MAX_BATCH // 2} utterances per guided decode batch, got {nu}")
        max_len = min(int(max_length or self.max_length), self.max_length)
        us = None
        if uniforms is not None:
            us = [u.to(self.device, torch.float32).reshape(u.shape[0], 1, V) for u in uniforms]
            max_len = min([max_len] + [u.shape[0] for u in us])
        self._ensure(nb, nb, max_len)
        b = self.buf
        if not collect_logits and max_len > 0:
            self._graph(float(temperature), nb, cond_scale)
        ctx = []
        for c in self._contexts(sources, range(0, nb, 2)):
            ctx += [c, 1]                                                # the null slot: row 0 (null k/v) only = every context key masked out
        for L in self.dec:
            L["kv_c"][1:nb:2, 0].copy_(L["null"][None, :].expand(nu, -1))
        uview = self._uniform_view(nb)
        if us is None:
            uview[0::2, :max_len].copy_(torch.rand(max_len, nu, 1, V, device=self.device, generator=generator).permute(1, 0, 2, 3))
        else:
            for u_, u in enumerate(us):
                uview[2 * u_, :max_len].copy_(u[:max_len])
        b["x"][:nb].copy_(self.start[None, :].expand(nb, -1))
        b["state"].copy_(self._slot_records(ctx))
        logits = []
        st = self._slot_records(ctx).tolist()[:nb]
        if collect_logits:
            for _ in range(max_len):
                self._run_steps(float(temperature), nb, 1, cond_scale)
                lg = b["logits"][:nb].clone()
                logits.append(lg[1::2] + (lg[0::2] - lg[1::2]) * cond_scale)
                st = self._read_state(nb)
                if all(st[2 * u_][1] for u_ in range(nu)):
                    break
        elif max_len > 0:
            st = self._decode_chunks(float(temperature), nb, max_len, cond_scale, [2 * u_ for u_ in range(nu)])
        out = []
        for u_ in range(nu):
            i = 2 * u_
            length = min(st[i][2] if st[i][1] and st[i][2] <= max_len else max_len, max_len)
            out.append(self._cut(i, length, torch.stack([lg[u_] for lg in logits]) if collect_logits and logits else None))
        return out

    @ops.gated
    @torch.no_grad()
    def generate_many(self, sources, uniforms=None, max_length: Optional[int] = None, temperature: float = 1.0,
                      generator: Optional[torch.Generator] = None, slots: int = 32, ignore_eos: bool = False, limits=None, on_done=None):
        """Decode ANY number of utterances through `slots` decode slots with continuous batching: every utterance runs the
        reference's loop (text2semantic.py:749-848) from position 0 to its first eos (:803-818) or its step limit, and the slot it
        ran in takes the next pending utterance in the sampling kernel of that very step (cvx_t2s_decoder.queue) - utterances end
        at different steps, and a lock-step batch would run half empty.  Every utterance gets exactly the tokens it gets alone.
        sources / uniforms as generate_batch; limits: optional per-utterance step limits (default max_length for all).
        on_done(j, (flat, streams)): called for utterance j as soon as the host has seen it finish (the host reads the dialogue
        records one chunk of CHUNK steps behind the device) - the next pipeline stage can start on the first results while the
        rest decodes.  Returns the list of (flat tokens, streams) in input order - int64 tensors ON THE HOST (they travel through pinned
        memory on a helper stream so that nothing makes the decode stream wait)."""
        d = self.d
        S, V = d["streams"], d["vocab"]
        n = len(sources)
        if n == 0:
            return []
        if n > WINDOW:            # (the context k/v, uniforms and token rows of every queued utterance are resident: bounded windows)
            out = []
            for w in range(0, n, WINDOW):
                out += self.generate_many(sources[w:w + WINDOW], None if uniforms is None else uniforms[w:w + WINDOW], max_length, temperature,
                                          generator, slots, ignore_eos, None if limits is None else limits[w:w + WINDOW],
                                          None if on_done is None else (lambda j, r, w=w: on_done(w + j, r)))
            return out
        nb = max(1, min(int(slots), MAX_BATCH, n))
        nb = nb if nb in (1, 2, 4) else min((nb + 7) // 8 * 8, MAX_BATCH)      # whole kernel groups (idle slots cost nothing)
        max_len = min(int(max_length or self.max_length), self.max_length)
        us = None
        if uniforms is not None:
            us = [u.to(self.device, torch.float32).reshape(u.shape[0], S, V) for u in uniforms]
            max_len = min([max_len] + [u.shape[0] for u in us])
        lim = [max_len] * n if limits is None else [max(1, min(int(x), max_len)) for x in limits]
        if max_len <= 0:
            raise ValueError("generate_many needs at least one step")
        self._ensure(nb, n, max_len)
        b = self.buf
        temperature = float(temperature)
        self._graph(temperature, nb, 1.0, True)
        ctx = self._contexts(sources)
        uview = self._uniform_view(n)
        if us is None:
            uview[:, :max_len].copy_(torch.rand(n, max_len, S, V, device=self.device, generator=generator))
        else:
            for j, u in enumerate(us):
                uview[j, :max_len].copy_(u[:max_len])
        flags = 1 if ignore_eos else 0
        first = min(nb, n)
        rec = torch.tensor([[ctx[j], lim[j], flags, 1 if j < first else 0, 0, j if j < first else 0, 0, 0] for j in range(n)], dtype=torch.int32)
        b["dialogues"][:n].copy_(rec)
        b["queue"].copy_(torch.tensor([first, n], dtype=torch.int32))
        slot = self._slot_records(ctx[:first]).clone()
        for j in range(first):
            slot[j, 5], slot[j, 6] = lim[j], flags
        b["state"].copy_(slot)
        b["x"][:first].copy_(self.start[None, :].expand(first, -1))
        self._mirror_setup()
        via_helper = ops.is_partition_stream()
        out: list = [None] * n
        seen = [False] * n
        # token rows of finished utterances reach the host through the helper stream into pinned memory (a dialogue's rows are final
        # once the host has SEEN it finished: any stream may read them) and are cut on the host - nothing here makes the decode
        # stream wait, so the chunks stay back to back (a device-side boolean index per utterance cost a stream sync each: 64
        # utterances on 64 slots ran 996 ms against 746 ms in lock step)
        tok_pin = torch.empty(n, S, max_len, dtype=torch.int64).pin_memory()
        copies: list = []                                  # (j, steps, event) in flight on the helper stream
        eos = V - 1

        def finish(block: bool):
            while copies and (block or copies[0][2].query()):
                j, length, ev = copies.pop(0)
                ev.synchronize()
                streams = tok_pin[j, :, :length].clone()
                after = (streams == eos).cumsum(dim=-1) > 0          # mask_after_eos (text2semantic.py:73-76)
                after = torch.nn.functional.pad(after, (1, -1), value=False)
                flat = streams.masked_fill(after, PAD_ID).reshape(-1)
                out[j] = (flat[flat != PAD_ID], streams)
                if on_done is not None:
                    on_done(j, out[j])

        def collect(records):
            self.last_records = records          # (tests / tools: status, steps and slot of every utterance)
            fresh = [j for j in range(n) if not seen[j] and records[j][3] >= 2]
            if fresh:
                with torch.cuda.stream(self._helper):
                    for j in fresh:
                        seen[j] = True
                        length = records[j][4]
                        tok_pin[j, :, :length].copy_(b["tokens"][j, :, :length], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                        copies.append((j, length, ev))
            finish(False)

        i, pending = 0, None
        cap = (sum(lim) + CHUNK - 1) // CHUNK + 4          # (one slot decoding everything: cannot be reached)
        while not all(seen) and i < cap:
            self._run_chunk(temperature, nb, 1.0, True)
            k = i & 1
            self._mirror_push(k, b["dialogues"][:n], via_helper)
            if pending is not None:
                collect(self._mirror_pull(pending, n))
            pending = k
            i += 1
        if not all(seen) and pending is not None:
            collect(self._mirror_pull(pending, n))
        if pending is not None:
            self._pin_ev[pending].synchronize()
        finish(True)
        if not all(seen):
            raise RuntimeError(f"text2semantic continuous decode: {seen.count(False)} of {n} utterances did not finish in {i} chunks")
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
