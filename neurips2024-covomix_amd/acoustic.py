"""VoMix / VoSingle vector field and ODE sampler on the MI355X kernels.

Host-side mirror of the reference classes on this path (paths relative to /root/reference):
  CoVoMix.forward / forward_with_cond_scale      covomix/covomix_model/acoustic.py:414-521
  Transformer / Attention / AdaptiveRMSNorm       acoustic.py:165-318
  ConditionalFlowMatcherWrapper.sample            acoustic.py:597-688   (torchdiffeq midpoint, :586-591)

MI355X-first restructuring (exact in real arithmetic; only fp32 rounding order changes):
  * both CFG branches run as ONE batch of 2B rows (the reference runs 2 sequential forwards);
  * the adaptive-norm projections depend only on t: all to_gamma/to_beta(time_emb) rows for the
    NFE time points are produced by two GEMMs per call instead of 32 GEMVs per forward (the
    reference re-reads 537 MB of weights per forward for them);
  * to_embed is linear in cat(x, phoneme_emb, cond): the 2208 step-invariant input columns are
    multiplied once per call, only the 80 x-columns every step;
  * RoPE is an epilogue of the to_qkv GEMM, attention never materialises T x T scores, GELU /
    bias / residual / skip-concat are GEMM epilogues or a K-split.
Python here only sequences kernel launches on torch's current stream and owns device buffers.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

from . import ops


def _dims_from_state(sd: Dict[str, torch.Tensor]) -> dict:
    dim, e_in = sd["to_embed.weight"].shape
    dim_cond = sd["null_cond"].shape[0]
    dim_emb = sd["to_phoneme_emb.weight"].shape[1]
    depth = 0
    while f"transformer.layers.{depth}.2.to_qkv.weight" in sd:
        depth += 1
    inner = sd["transformer.layers.0.2.to_qkv.weight"].shape[0] // 3
    dim_out = sd["to_pred.weight"].shape[0]
    streams = (e_in - dim_out - dim_cond) // dim_emb
    if dim_out + streams * dim_emb + dim_cond != e_in or inner % 64 != 0:
        raise ValueError("unsupported CoVoMix checkpoint geometry")
    return dict(dim=dim, e_in=e_in, dim_cond=dim_cond, dim_emb=dim_emb, depth=depth, heads=inner // 64,
                dim_head=64, dim_out=dim_out, streams=streams,
                null_id=sd["to_phoneme_emb.weight"].shape[0] - 1,
                time_hidden=sd["sinu_pos_emb.1.weight"].shape[0])


class VectorField:
    """Device-resident weights of one CoVoMix network + the launch sequence of one evaluation."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device: torch.device, precision: str = "f16x3"):
        """precision: 'f16x3' (default) runs the transformer GEMMs on the fp16 matrix pipe with every operand split
        into (hi, lo) fp16 halves and three MFMA products - fp32-class accuracy (kernel error 5.8e-7 vs 5.1e-7 for
        fp32 MFMA, tests/test_kernels_gpu.py) at 2x the rate; 'fp32' uses v_mfma_f32_32x32x2_f32 everywhere;
        'f16' (opt-in) keeps only the hi halves: plain fp16 GEMM / attention operands with fp32 accumulation,
        softmax, norms and residual stream - inside the 1e-3 rel-L2 budget of BASELINE.json, not fp32-class."""
        if precision not in ("f16x3", "fp32", "f16"):
            raise ValueError(f"precision must be 'f16x3', 'f16' or 'fp32', got {precision!r}")
        self.precision = precision
        if torch.device(device).type == "cuda":
            with torch.cuda.device(device):
                ops.saturation_reset()      # (allocates the device's saturation flag now, outside any stream capture)
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in state_dict.items()}
        self.device = device
        self.d = d = _dims_from_state(sd)
        if sd["conv_embed.dw_conv1d.0.weight"].shape[-1] != 31:
            raise ValueError("only conv_pos_embed_kernel_size == 31 is supported (reference default)")
        self.sd = sd
        # adaptive-norm projection weights packed [2*n_norms*dim, time_hidden]: (gamma, beta) per norm
        ws, bs = [], []
        for i in range(d["depth"]):
            for n in (1, 3):
                p = f"transformer.layers.{i}.{n}"
                ws += [sd[p + ".to_gamma.weight"], sd[p + ".to_beta.weight"]]
                bs += [sd[p + ".to_gamma.bias"], sd[p + ".to_beta.bias"]]
        self.ada_w = torch.cat(ws, dim=0).contiguous()
        self.ada_b = torch.cat(bs, dim=0).contiguous()
        for i in range(d["depth"]):          # the unpacked copies are never read again
            for n in (1, 3):
                for nm in ("to_gamma", "to_beta"):
                    for s in ("weight", "bias"):
                        del sd[f"transformer.layers.{i}.{n}.{nm}.{s}"]
        self.dw_w = sd["conv_embed.dw_conv1d.0.weight"].reshape(d["dim"], 31).contiguous()
        inv_freq = sd.get("transformer.rotary_emb.inv_freq")
        if inv_freq is None:
            inv_freq = 1.0 / (10000 ** (torch.arange(0, 64, 2, device=device).float() / 64))
        self.inv_freq = inv_freq
        self._ws: Dict[tuple, dict] = {}
        # split (fp16 hi, fp16 lo, 1/scale) copies of the big GEMM weights - load-time packing
        self.split: Dict[str, tuple] = {}
        if precision in ("f16x3", "f16"):
            for k, v in sd.items():
                if k.endswith(".weight") and v.ndim == 2 and v.shape[1] % 32 == 0 and (
                        ".2.to_qkv" in k or ".2.to_out" in k or ".4.0." in k or ".4.2." in k
                        or (k.startswith("transformer.layers.") and k.endswith(".0.weight")) or k == "to_pred.weight"):
                    self.split[k] = ops.split_f16(v, with_lo=(precision == "f16x3"))
            # the step-invariant columns of to_embed (phoneme embeddings | conditioning mel: one [2BT, 2208] x [2208, 1024] product
            # per solve) - 0.74 ms on the fp32 pipe at the bench shape
            # the adaptive-norm table GEMM - [n evaluation times, dim] x the packed [4 depth dim, dim] matrix: pure weight
            # streaming at 32 rows - 1.35 ms per solve on the fp32 kernel inside the model (cold weights), 0.76 ms here
            # (rocprofv3).  Neither is close to the 134 MB / HBM rate = 30 us: DESIGN 4.4
            # (solves of more than 32 evaluation times build self.split["ada"] on demand: _time_tables)
            w_rest = sd["to_embed.weight"][:, d["dim_out"]:]
            if precision == "f16x3" and w_rest.shape[1] % 32 == 0:
                self.w_rest = w_rest.contiguous()
                self.split["to_embed.rest"] = ops.split_f16(self.w_rest)
        self._init_gain_model()
        # interleaved copies ([hi 32 | lo 32] per K-step: whole cache lines for the DMA) for the large-problem kernel
        self.split_il: Dict[str, tuple] = {}
        self._dn_bufs: Dict[tuple, dict] = {}          # deferred norm: W diag(gamma(t)) pairs per (n, depth, solver grid)
        self._time_cache: Dict[tuple, dict] = {}       # solver grid (nfe, method) -> everything that depends on the evaluation times only
        if precision == "f16x3":
            for k, v in self.split.items():
                if v[0].shape[0] >= 512 or k == "to_pred.weight":
                    self.split_il[k] = ops.split_f16_interleaved(v)

    # ------------------------------------------------------------------ activation pre-scales (scale-free split pairs)
    # An (fp16 hi, fp16 lo) pair has full (22-bit) precision only while |x| is inside [2^-3, 2^16): below, `lo` falls into
    # fp16's subnormals (absolute floor 2^-25), above, `hi` saturates.  Every split ACTIVATION tensor is therefore written
    # times a power of two that puts its expected RMS at 2^4 (full precision for elements from 2^-7 to 2^12 times the RMS),
    # and the consuming kernel divides it out of its fp32 accumulators - exactly what load-time packing does for the
    # weights.  The powers come from a GAIN MODEL, not from the data: RMS of an AdaRMSNorm output = sqrt(mean(gamma^2) +
    # mean(beta^2)) of that evaluation time's table row (exact for RMS-normalised input), RMS after a Linear = input RMS
    # x ||W||_F / sqrt(N).  They are computed on the device by prepare() (no host round trip: the solve stays
    # graph-capturable), depend on the weights and the evaluation times only - never on the utterance - and sit in one
    # small tensor the kernels read through DEVICE pointers (cvx_gemm_split_io.a_scale_dev, ...).
    WORKSPACE_SHAPES = 4
    N_KINDS = 6          # per (evaluation, layer): normed->qkv, q|k, v, attention out, normed->ff1, ff hidden
    TARGET_RMS = 16.0

    def _init_gain_model(self) -> None:
        d, sd = self.d, self.sd
        inner = d["heads"] * 64

        def fro(w, rows=None):              # ||W||_F / sqrt(N): RMS gain of y = W x for x with unit per-element RMS
            w = w if rows is None else w[rows[0]:rows[1]]
            return float(w.double().square().sum().sqrt() / (w.shape[0] ** 0.5))
        g = dict(qk=[], v=[], o=[], f1=[], b1=[], f2=[], comb=[])
        for i in range(d["depth"]):
            p = f"transformer.layers.{i}"
            wq = sd[p + ".2.to_qkv.weight"]
            g["qk"].append(fro(wq, (0, 2 * inner))); g["v"].append(fro(wq, (2 * inner, 3 * inner)))
            g["o"].append(fro(sd[p + ".2.to_out.weight"]))
            g["f1"].append(fro(sd[p + ".4.0.weight"])); g["b1"].append(float(sd[p + ".4.0.bias"].double().square().mean()))
            g["f2"].append(fro(sd[p + ".4.2.weight"]))
            g["comb"].append(fro(sd[p + ".0.weight"]) if (p + ".0.weight") in sd else 0.0)
        dev = self.device
        self.gain = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in g.items()}
        # embedding output: x columns (x ~ N(0,1) at t = 0, O(1) later), phoneme-embedding columns (RMS of the table), cond
        # columns (log-mel prompt frames, |.| of a few units: taken as RMS 4), bias
        we = sd["to_embed.weight"].double()
        n_x, n_e = d["dim_out"], d["streams"] * d["dim_emb"]
        emb_ms = float(sd["to_phoneme_emb.weight"].double().square().mean())
        ms = lambda w: float(w.square().sum() / w.shape[0])
        self.embed_ms = (ms(we[:, :n_x]) + emb_ms * ms(we[:, n_x:n_x + n_e]) + 16.0 * ms(we[:, n_x + n_e:])
                         + float(sd["to_embed.bias"].double().square().mean()))
        self.has_comb = [c > 0.0 for c in g["comb"]]
        gf = float(sd["transformer.final_norm.gamma"].double().square().mean().sqrt())
        self.pred_scale = torch.tensor([self._pow2(self.TARGET_RMS / max(gf, 1e-30))], dtype=torch.float32, device=dev)

    @staticmethod
    def _pow2(x: float) -> float:
        import math
        return 2.0 ** max(-40, min(40, round(math.log2(x)))) if x > 0 and math.isfinite(x) else 1.0

    def _activation_scales(self, table: torch.Tensor) -> tuple:
        """(S [n_eval, depth, N_KINDS], H [1], HS [depth, 4]) power-of-two pre-scales from the gain model; device tensors, no host sync."""
        d, gn = self.d, self.gain
        n, L, dim = table.shape[0], d["depth"], d["dim"]
        tab = table.view(n, L, 4, dim)
        ms = tab.square().mean(dim=-1)                                  # [n, L, 4]: mean(gamma_a^2), mean(beta_a^2), gamma_f, beta_f
        rn_a = (ms[..., 0] + ms[..., 1]).sqrt()
        rn_f = (ms[..., 2] + ms[..., 3]).sqrt()
        qk, v = rn_a * gn["qk"], rn_a * gn["v"]
        att = v * 0.25                                                  # softmax averages values: between v_rms / sqrt(T) and v_rms
        ff = ((rn_f * gn["f1"]).square() + gn["b1"]).sqrt() * 0.5       # GELU(z) ~ z / 2 .. 0.6 z
        rms = torch.stack([rn_a, qk, v, att, rn_f, ff], dim=-1)         # [n, L, 6]
        tiny = torch.finfo(torch.float32).tiny
        S = torch.exp2(torch.round(torch.log2(self.TARGET_RMS / rms.clamp_min(tiny))).clamp(-40, 40)).contiguous()
        # residual stream (split twins of h feed the skip combiners; all of them share ONE scale because a combiner reads two
        # of them as the halves of its K range): embedding output, then every block adds its attention and FF output
        h2 = torch.full((), self.embed_ms * 2.25, dtype=torch.float32, device=table.device)   # x + GELU(conv(x)): up to 1.5 x
        hmax = h2
        att_o = (att * gn["o"]).square().amax(dim=0)                    # [L]: largest over the evaluation times
        ff_o = (ff * gn["f2"]).square().amax(dim=0)
        skips = []
        stages = []                          # mean squares of the stream per layer: at its input, behind its skip combiner, behind to_out, behind ff2
        for i in range(L):
            h_in = h2
            if self.has_comb[i]:
                h2 = (h2 + skips.pop()) * gn["comb"][i].square()
            else:
                skips.append(h2)
            h_c = h2
            h_a = h2 + att_o[i]
            h2 = h_a + ff_o[i]
            stages.append(torch.stack([h_in, h_c, h_a, h2]))
            hmax = torch.maximum(hmax, h2)
        H = torch.exp2(torch.round(torch.log2(self.TARGET_RMS / hmax.sqrt().clamp_min(tiny))).clamp(-40, 40)).reshape(1).contiguous()
        # the pair-only residual stream of the deferred-norm path (section 4.1d) carries ONE pre-scale PER STAGE: the stream IS those
        # pairs there, and a stage far below the largest one would sit under the full-precision window of a shared scale
        HS = torch.exp2(torch.round(torch.log2(self.TARGET_RMS / torch.stack(stages).sqrt().clamp_min(tiny))).clamp(-40, 40)).contiguous()
        return S, H, HS

    # ------------------------------------------------------------------ workspace
    RAGGED_ROW_QUANTUM = 1024       # ragged batches: workspaces are sized in steps of this many rows and shared by every
                                    # batch composition that fits (a directory never repeats a composition)

    def _workspace(self, Bt: int, T: int, ragged_rows: int = 0) -> dict:
        """Buffers of one batch shape.  Equal-length batch: Bt sequences of T frames.  Ragged batch (ragged_rows = M > 0):
        capacity rounded up to RAGGED_ROW_QUANTUM rows, ONE V^T row set per head over all packed rows, no RoPE table (it
        depends on the composition: prepare() builds it per call); callers use a row-cut view (_ws_cut)."""
        d, dev = self.d, self.device
        if ragged_rows:
            q = self.RAGGED_ROW_QUANTUM
            M = (ragged_rows + q - 1) // q * q
            key = ("ragged", M, ragged_rows >= ops.il_min_rows())  # (the interleaved-pair choice below depends on the real row count)
        else:
            M = Bt * T
            key = (Bt, T)
        ws = self._ws.pop(key, None)
        if ws is not None:
            self._ws[key] = ws                                   # most recently used last
            return ws
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        ws = dict(
            h=[f(M, d["dim"]) for _ in range(d["depth"] // 2 + 3)],
            normed=f(M, d["dim"]), qkv=f(M, 3 * d["heads"] * 64), att=f(M, d["heads"] * 64),
            ff=f(M, 4 * d["dim"]), base=f(M, d["dim"]), xin=f(M, d["dim_out"]), pred=f(M, d["dim_out"]),
            gathered=f(M, d["streams"] * d["dim_emb"] + d["dim_cond"]),
            rowsq=f(M, max(d["dim"] // 64, 1)), rs=f(M),           # deferred norm: row sums of squares per 64 columns, factor per row
        )
        if self.precision in ("f16x3", "f16"):      # activations that only feed GEMMs live as (fp16 hi, fp16 lo) pairs
            lo_too = self.precision == "f16x3"          # 'f16': (hi, None)
            h16 = lambda *s: (torch.empty(*s, dtype=torch.float16, device=dev),
                              torch.empty(*s, dtype=torch.float16, device=dev) if lo_too else None)
            # GEMM A operands: for the large-problem kernel (M >= 2048, interleaved weights available) as INTERLEAVED pairs
            # ([hi 32 | lo 32] per K-step: whole cache lines for the DMA), otherwise as two separate fp16 tensors
            a16 = h16
            if lo_too and (ragged_rows or M) >= ops.il_min_rows() and self.split_il and d["dim"] >= 512:
                a16 = lambda rows, cols: ops.SplitIL(rows, cols, dev)
            ws["normed16"], ws["att16"], ws["ff16"] = a16(M, d["dim"]), a16(M, d["heads"] * 64), a16(M, 4 * d["dim"])
            # final norm -> to_pred (N = 80): the medium-problem kernel takes it (interleaved pair; K slices below 2048 rows)
            ws["pred16"] = a16(M, d["dim"]) if (a16 is not h16 and d["dim_out"] % 16 == 0) else h16(M, d["dim"])
            ws["qk16"] = h16(M, 2 * d["heads"] * 64)
            ws["h16"] = [a16(M, d["dim"]) for _ in ws["h"]]      # split twins of the residual-stream buffers (skip GEMMs)
            # V^T rows, zero (always finite) beyond the last frame: read by the last key tile with weight 0
            vt_rows, Tp = (d["heads"] * 64, M) if ragged_rows else (Bt * d["heads"] * 64, ((T + 31) // 32) * 32)
            ws["vt16"] = (torch.zeros(vt_rows, Tp, dtype=torch.float16, device=dev),
                          torch.zeros(vt_rows, Tp, dtype=torch.float16, device=dev) if lo_too else None)
        if not ragged_rows:
            ws["rope"] = self._rope_tables(torch.arange(T, device=dev, dtype=torch.float32))
        while len(self._ws) >= self.WORKSPACE_SHAPES:          # keep the most recent shapes resident (a directory of
            self._ws.pop(next(iter(self._ws)))                  # utterances alternates between a few lengths; ~0.4 GB per 1000 rows)
        self._ws[key] = ws
        return ws

    def _rope_tables(self, pos: torch.Tensor) -> tuple:
        """(cos, sin) [len(pos), 32] of the half-split rotary embedding at the given positions (acoustic.py:120-137)."""
        ang = pos[:, None] * self.inv_freq[None, :]
        return ang.cos().contiguous(), ang.sin().contiguous()

    # ------------------------------------------------------------------ per-call setup
    def prepare(self, phoneme_ids: torch.Tensor, cond: torch.Tensor, times: torch.Tensor, use_null: bool,
                lengths: Optional[List[int]] = None, times_key=None) -> dict:
        """Everything that does not depend on the ODE state x:
        time MLP + adaptive-norm tables for every evaluation time (_time_tables: computed once per solver grid `times_key` and
        model), and the step-invariant part of to_embed.
        lengths: ragged batch - phoneme_ids [M1(, S)] and cond [M1, C] hold the utterances back to back (M1 = sum(lengths));
        the null-branch rows repeat the same sequence structure behind them."""
        d, sd = self.d, self.sd
        rg = None
        if lengths is not None:
            B, T, M1 = len(lengths), None, int(sum(lengths))
            assert cond.ndim == 2 and cond.shape[0] == M1 and phoneme_ids.shape[0] == M1
            Bt = 2 * B if use_null else B
            rg = ops.Ragged(lengths, self.device, repeat=2 if use_null else 1)
            full = self._workspace(0, 0, ragged_rows=rg.M)
            ws = self._ws_cut(full, 0, rg.M, vt=None)            # row views of the capacity-sized buffers; V^T stays whole
            ws["rope"] = self._rope_tables(rg.positions())       # per-ROW tables: the to_qkv epilogue then runs with rope_T = M
        else:
            B, T, _ = cond.shape
            Bt = 2 * B if use_null else B
            M1 = B * T
            ws = self._workspace(Bt, T)
        n = times.numel()
        tt = self._time_tables(times, times_key)
        table = tt["table"]
        # step-invariant to_embed columns: rows [0, M1) conditional, rows [M1, 2*M1) null branch
        g = ws["gathered"]
        ids = phoneme_ids.to(torch.int64).contiguous()
        ops.embed_gather(ids, d["streams"], sd["to_phoneme_emb.weight"], cond.contiguous(), None, d["dim_cond"],
                         d["null_id"], g[:M1], M1)
        if use_null:
            ops.embed_gather(None, d["streams"], sd["to_phoneme_emb.weight"], None, sd["null_cond"], d["dim_cond"],
                             d["null_id"], g[M1:], M1)
        if "to_embed.rest" in self.split:
            ops.gemm(g, self.w_rest, ws["base"], bias=sd["to_embed.bias"], w_split=self.split["to_embed.rest"])
        else:
            ops.gemm(g, sd["to_embed.weight"][:, d["dim_out"]:], ws["base"], bias=sd["to_embed.bias"])
        ctx = dict(ws=ws, table=table, B=B, T=T, Bt=Bt, M1=M1, use_null=use_null, M=(2 * M1 if use_null else M1), ragged=rg)
        if "S" in tt:
            ctx["scales"], ctx["h_scale"] = tt["S"], tt["H"]                   # keep the tensors alive as long as the pointers
            ctx["sp"], ctx["hp"] = tt["sp"], tt["H"].data_ptr()
            if self._defers(ctx["M"], ws):
                if tt["dn"] is None:
                    # built by split-pair kernels (split_f16_colscale_il): cached only when the build itself stayed inside the window
                    # (_tables_clean) - a clamped table must not outlive the call that is flagged for it
                    clean0 = self._flag_clear()
                    dn = self._deferred_norm_tables(table, tt["HS"], times_key)
                    if tt.get("cached") and clean0 and self._flag_clear():
                        tt["dn"], tt["dn_ready"] = dn, self._ready_event()
                    ctx["dn"] = dn
                else:
                    self._wait(tt["dn_ready"])                                  # (built on another stream, maybe)
                    ctx["dn"] = tt["dn"]
        return ctx

    @staticmethod
    def _flag_clear():
        """Is the current stream's sticky saturation flag clear right now (one stream synchronisation)?  None when nothing can be
        said: inside a stream capture, or with the checks switched off (CVX_SAT_CHECK=0)."""
        if torch.cuda.is_current_stream_capturing() or os.environ.get("CVX_SAT_CHECK", "1") != "1":
            return None
        return ops.saturation_query(reset=False) == 0

    @staticmethod
    def _wait(ev) -> None:
        """the current stream waits for tables another stream may have built (not inside a capture: a capture starts after the
        capturing call's own eager run and a stream synchronisation - everything cached is complete by then)"""
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(ev)

    @staticmethod
    def _ready_event():
        ev = torch.cuda.Event()
        ev.record()
        return ev

    # ------------------------------------------------------------------ everything that depends on the evaluation times only
    TIME_CACHE = 2          # evaluation-time grids whose tables stay resident (each holds 0.22 GB per evaluation time once a batch deferred its norms)

    def _time_tables(self, times: torch.Tensor, key) -> dict:
        """Time MLP + adaptive-norm table for every evaluation time, the activation pre-scales of the gain model and (filled in by
        prepare() when a batch defers its norms) the deferred-norm weight tables.  All of it is a function of the CHECKPOINT and the
        evaluation times - gamma / beta are functions of the time embedding alone (reference acoustic.py:198-204) - never of the
        utterance, so a solver grid `key` = (nfe, method) computes it once per VectorField (= per set of weights: CoVoMixModel builds a
        new field when its weights change) and every later call reuses it: 15 split_colscale_il + 34 weight-streaming launches and
        ~30 torch launches less per call.  key = None: not cached.  At most TIME_CACHE grids stay resident; evicting one also
        drops the captured graphs (they hold addresses of its tables)."""
        d, sd = self.d, self.sd
        if key is not None and key in self._time_cache:
            ent = self._time_cache.pop(key)
            self._time_cache[key] = ent                                        # most recently used last
            self._wait(ent["ready"])                                           # (the tables may have been built on another stream)
            return ent
        # The tables are built by split-precision kernels that can raise the sticky saturation flag (split_act_f16(temb), the f16x3
        # GEMM of the table above 32 rows, the activation scales).  A call that builds them and is flagged is re-run in fp32 - but a
        # CACHED clamped table would be reused by later calls under a clean flag: silently wrong.  So an entry is cached only when
        # its build provably stayed inside the window: flag clear before AND after (two stream synchronisations per (model, grid)).
        clean0 = self._flag_clear() if key is not None else None
        n = times.numel()
        four = torch.empty(n, d["dim"], dtype=torch.float32, device=self.device)
        ops.time_fourier(times, sd["sinu_pos_emb.0.weights"], four)
        temb = torch.empty(n, d["time_hidden"], dtype=torch.float32, device=self.device)
        # products with n <= 32 rows are weight streaming: cvx_gemm_skinny_f32 (the table, 537 MB of weights: 1.35 ms on the tiled fp32
        # kernel, 0.79 on the split-precision one, 0.25 here)
        skinny = lambda w: n <= 32 and w.shape[1] % 8 == 0
        if skinny(sd["sinu_pos_emb.1.weight"]):
            ops.gemm_skinny(four, sd["sinu_pos_emb.1.weight"], temb, bias=sd["sinu_pos_emb.1.bias"], act=ops.ACT_SILU)
        else:
            ops.gemm(four, sd["sinu_pos_emb.1.weight"], temb, bias=sd["sinu_pos_emb.1.bias"], act=ops.ACT_SILU)
        table = torch.empty(n, self.ada_w.shape[0], dtype=torch.float32, device=self.device)
        if skinny(self.ada_w):
            ops.gemm_skinny(temb, self.ada_w, table, bias=self.ada_b)
        elif "ada" in self.split or (self.precision == "f16x3" and self.ada_w.shape[1] % 32 == 0):
            if "ada" not in self.split:
                self.split["ada"] = ops.split_f16(self.ada_w)
            ops.gemm(temb, self.ada_w, table, bias=self.ada_b, w_split=self.split["ada"], a_split=ops.split_act_f16(temb))
        else:
            ops.gemm(temb, self.ada_w, table, bias=self.ada_b)
        ent = dict(table=table, dn=None)
        if self.precision in ("f16x3", "f16"):
            S, H, HS = self._activation_scales(table)
            p0, K = S.data_ptr(), self.N_KINDS
            ent.update(S=S, H=H, HS=HS, sp=[[[p0 + 4 * ((e * d["depth"] + i) * K + k) for k in range(K)] for i in range(d["depth"])] for e in range(n)])
        ent["ready"] = self._ready_event()
        if key is not None and clean0 and self._flag_clear():     # (None inside a capture: tables made there belong to that graph's pool)
            ent["cached"] = True
            while len(self._time_cache) >= self.TIME_CACHE:
                old = next(iter(self._time_cache))
                del self._time_cache[old]
                self._dn_bufs = {k: v for k, v in self._dn_bufs.items() if k[2] != old}
                self.__dict__.pop("_graphs", None)
            self._time_cache[key] = ent
        return ent

    # ------------------------------------------------------------------ deferred AdaptiveRMSNorm (large batches)
    DEFER_MIN_ROWS = int(os.environ.get("CVX_DEFER_NORM_ROWS", "8192"))
    DEFER_RULE = False       # (round 4 kept batches whose 256-row rounds came out part-empty - 18,000 / 20,480 rows - off this path; with the
                             #  large-problem kernel's 192-row tiles, round 5, the path wins at every size from 8192 rows: tools/archive/defer_rows_bench.py,
                             #  20,480 rows 448.7 vs 470.1 ms, 18,000 rows 404.2 vs 417.5, 9,300 rows 232.8 vs 242.1)

    def _defers(self, M: int, ws: dict) -> bool:
        """Large batches run WITHOUT the norm kernel and with the residual stream as split pairs only.  An AdaptiveRMSNorm is one
        gamma / beta row per evaluation for every frame (acoustic.py:198-204), so for the product that follows it (:306-318)
            norm(x) W^T = (sqrt(D) / ||x_row||) * (x (W diag(gamma))^T) + beta W^T :
        * the producing GEMM (to_out, ff2, skip combiner) writes its output ONLY as the split pair the skip combiners already
          needed (no fp32 store: it moves the bytes the fp32 form moved), reads its residual from that pair, and leaves the rows'
          sums of squares per 64 columns (cvx_gemm_split_io.R_hi / c_rowsq); cvx_rownorm_scale_f32 makes one factor per row;
        * the consuming GEMM (to_qkv, ff1) multiplies its accumulator rows by that factor, carries beta W^T in its bias and runs on
          W diag(gamma(t)) - split copies of the weights for every evaluation time of the solver grid, built ONCE per (model, grid)
          (_time_tables; cvx_split_f16_colscale_il: 7 GB for 32 evaluation times - capacity HBM3E has - written in ~2 ms).
        The norm kernel (131 MB of traffic per launch at the bench shape, 16 of 17 launches per evaluation) does not run and ff2 no
        longer writes its output twice.  Only the first layer's attention norm (input from the embedding, fp32) and the final norm
        stay.  Large-problem kernel only: batches of DEFER_MIN_ROWS rows and more (CVX_DEFER_NORM=0: off)."""
        if not (self.precision == "f16x3" and M >= self.DEFER_MIN_ROWS and isinstance(ws.get("normed16"), ops.SplitIL)
                and self.d["dim"] % 64 == 0 and 512 <= self.d["dim"] <= 4096 and os.environ.get("CVX_DEFER_NORM", "1") == "1"):
            return False
        return True

    def _deferred_norm_tables(self, table: torch.Tensor, HS: torch.Tensor, times_key) -> dict:
        """Per (evaluation time, layer): W diag(gamma) for to_qkv (layers 1..) and ff1 as interleaved split pairs, beta W^T as their
        bias (ff1: b1 + beta_ff W1^T), and the power of two that keeps |gamma| <= 1 inside the weight pair (divided out on the
        accumulators through the consumer's a_scale).  A function of the weights and the evaluation times: built once per solver
        grid (_time_tables)."""
        d, sd = self.d, self.sd
        n, L, dim = table.shape[0], d["depth"], d["dim"]
        tab = table.view(n, L, 4, dim)
        tiny = torch.finfo(torch.float32).tiny
        gmax = tab[:, :, 0::2, :].abs().amax(dim=-1)                                     # [n, L, 2]: max |gamma_attn|, max |gamma_ff|
        gs = torch.exp2(-torch.ceil(torch.log2(gmax.clamp_min(tiny))).clamp(-40, 40)).contiguous()
        # HS [L, 4]: stream pre-scale at a layer's input / behind its combiner / to_out / ff2
        AS = (gs * HS[None, :, 1:3]).contiguous()                                        # pre-scale of the consumers' A pairs (stage scale) times the weights' gs

        def beta_w(beta_rows: torch.Tensor, w: torch.Tensor, bias) -> torch.Tensor:
            out = torch.empty(n, w.shape[0], dtype=torch.float32, device=self.device)
            for r0 in range(0, n, 32):                                                   # weight streaming: 32 rows at a time on the skinny kernel
                r1 = min(n, r0 + 32)
                ops.gemm_skinny(beta_rows[r0:r1], w, out[r0:r1], bias=bias)
            return out
        key = (n, L, times_key)
        bufs = self._dn_bufs.get(key)
        if bufs is None:                     # (0.22 GB per evaluation time; released with the grid's cache entry, _time_tables)
            mk = lambda N: torch.empty(n, N, 2 * dim, dtype=torch.float16, device=self.device)
            bufs = dict(wq=[None if i == 0 else mk(3 * d["heads"] * 64) for i in range(L)], w1=[mk(4 * dim) for _ in range(L)])
            self._dn_bufs[key] = bufs
        b1p, bq, wq, w1 = [], [], [], []
        for i in range(L):
            p = f"transformer.layers.{i}"
            nq, n1 = p + ".2.to_qkv.weight", p + ".4.0.weight"
            b1p.append(beta_w(tab[:, i, 3, :], sd[n1], sd[p + ".4.0.bias"]))
            inv1 = self.split_il[n1][1]
            ops.split_f16_colscale_il(sd[n1], tab[:, i, 2, :], gs[:, i, 1], 1.0 / inv1, bufs["w1"][i])
            w1.append([(bufs["w1"][i][e], inv1) for e in range(n)])
            if i == 0:
                bq.append(None); wq.append(None)
                continue
            bq.append(beta_w(tab[:, i, 1, :], sd[nq], None))
            invq = self.split_il[nq][1]
            ops.split_f16_colscale_il(sd[nq], tab[:, i, 0, :], gs[:, i, 0], 1.0 / invq, bufs["wq"][i])
            wq.append([(bufs["wq"][i][e], invq) for e in range(n)])
        a0, h0 = AS.data_ptr(), HS.data_ptr()
        return dict(b1p=b1p, bq=bq, wq=wq, w1=w1, gs=gs, AS=AS, HS=HS, hsp=[[h0 + 4 * (4 * i + k) for k in range(4)] for i in range(L)],
                         asp=[[(a0 + 4 * ((e * L + i) * 2), a0 + 4 * ((e * L + i) * 2 + 1)) for i in range(L)] for e in range(n)])

    def _layers_pair_stream(self, ctx: dict, step: int, ws: dict, h: torch.Tensor, twin: dict, free: list, Bt: int, T: int, M: int, rg):
        """The transformer layers of one evaluation with the residual stream as split pairs and every AdaptiveRMSNorm deferred into
        the GEMMs on either side of it (see _defers).  h: the embedding output (fp32, its pair already in twin[id(h)])."""
        d, sd, dn = self.d, self.sd, ctx["dn"]
        dim, L = d["dim"], d["depth"]
        tab = ctx["table"][step]
        sp, il = self.split.get, self.split_il.get
        sp_step, pp, hsp = ctx["sp"][step], self.pred_scale.data_ptr(), dn["hsp"]
        cur = hsp[0][0]                      # device pointer of the pre-scale the current h's pair carries (one per stage, _activation_scales)
        take = free.pop
        parts64, rt_dim = dim // 64, float(dim) ** 0.5
        n16, a16, f16 = ws["normed16"], ws["att16"], ws["ff16"]
        rowsq = ws["rowsq"]
        rs = ws["rs"]
        def factor():                        # sqrt(D) / ||row|| from the producer's partial sums (computing it in the consumer's epilogue instead
            ops.rownorm_scale(rowsq, M, parts64, rs, rt_dim)       # was built and measured: + 5 ms per step against these 480 launches of 5 us)
        fkw = dict(a_row_scale=rs)
        skips: List[tuple] = []
        have_rs = False                      # rowsq holds the sums of squares of the current h's rows (per 64 columns)
        for i in range(L):
            p = f"transformer.layers.{i}"
            s_na, s_qk, s_v, s_at, s_nf, s_ff = sp_step[i]
            as_a, as_f = dn["asp"][step][i]
            last = i + 1 == L
            if self.has_comb[i]:
                s, s_scale = skips.pop()
                comb = take()
                ops.gemm(h, sd[p + ".0.weight"], comb, bias=sd[p + ".0.bias"], a2=s, w_split=sp(p + ".0.weight"), w_il=il(p + ".0.weight"),
                         a_split=twin[id(h)], a2_split=twin[id(s)], a_scale=cur, a2_scale=s_scale, out_split=twin[id(comb)], c_scale=hsp[i][1],
                         c_rowsq=rowsq, write_f32=False)
                cur = hsp[i][1]
                factor()
                have_rs = True
                free += [h, s]
                h, keep_input = comb, False
            else:
                skips.append((h, cur))
                keep_input = True
            nq = p + ".2.to_qkv.weight"
            if have_rs:
                ops.gemm(ws["normed"], sd[nq], ws["qkv"], rope=ws["rope"], rope_cols=2 * d["heads"] * 64, w_split=sp(nq), w_il=dn["wq"][i][step],
                         a_split=twin[id(h)], out_split=ws["qk16"], vt_split=ws["vt16"], write_f32=False, a_scale=as_a, c_scale=s_qk, vt_scale=s_v,
                         bias=dn["bq"][i][step], **fkw)
            else:                            # first layer: the embedding output exists in fp32
                ops.adarmsnorm(h, tab[(4 * i) * dim:(4 * i + 1) * dim], tab[(4 * i + 1) * dim:(4 * i + 2) * dim], None, out_split=n16, split_scale=s_na)
                ops.gemm(ws["normed"], sd[nq], ws["qkv"], rope=ws["rope"], rope_cols=2 * d["heads"] * 64, w_split=sp(nq), w_il=il(nq),
                         a_split=n16, out_split=ws["qk16"], vt_split=ws["vt16"], write_f32=False, a_scale=s_na, c_scale=s_qk, vt_scale=s_v)
            ops.attention_f16x3(ws["qk16"], ws["vt16"], None, Bt, T, d["heads"], 64 ** -0.5, out_split=a16,
                                qk_scale=s_qk, v_scale=s_v, out_scale=s_at, ragged=rg)
            h_att = take() if keep_input else h
            no = p + ".2.to_out.weight"
            ops.gemm(ws["att"], sd[no], h_att, w_split=sp(no), w_il=il(no), a_split=a16, a_scale=s_at, res_split=twin[id(h)], res_scale=cur,
                     out_split=twin[id(h_att)], c_scale=hsp[i][2], c_rowsq=rowsq, write_f32=False)
            cur = hsp[i][2]
            factor()
            h = h_att
            n1, n2 = p + ".4.0.weight", p + ".4.2.weight"
            ops.gemm(ws["normed"], sd[n1], ws["ff"], bias=dn["b1p"][i][step], act=ops.ACT_GELU, w_split=sp(n1), w_il=dn["w1"][i][step],
                     a_split=twin[id(h)], out_split=f16, write_f32=False, a_scale=as_f, c_scale=s_ff, **fkw)
            next_defers = (not last) and not self.has_comb[i + 1]          # the next layer's attention norm reads THIS output's rows
            ops.gemm(ws["ff"], sd[n2], h, bias=sd[p + ".4.2.bias"], w_split=sp(n2), w_il=il(n2), a_split=f16, a_scale=s_ff,
                     res_split=twin[id(h)], res_scale=cur, out_split=twin[id(h)], c_scale=hsp[i][3], c_rowsq=rowsq if next_defers else None,
                     write_f32=last)                                      # (the final norm reads fp32)
            cur = hsp[i][3]
            if next_defers:
                factor()
            have_rs = next_defers
        ops.adarmsnorm(h, sd["transformer.final_norm.gamma"], None, None, out_split=ws["pred16"], split_scale=pp)
        ops.gemm(ws["normed"], sd["to_pred.weight"], ws["pred"], w_split=sp("to_pred.weight"), a_split=ws["pred16"], a_scale=pp,
                 w_il=il("to_pred.weight") if isinstance(ws["pred16"], ops.SplitIL) else None)
        return ws["pred"]

    # ------------------------------------------------------------------ one evaluation (both CFG branches)
    @staticmethod
    def _ws_cut(ws: dict, r0: int, r1: int, vt) -> dict:
        """Views of every workspace buffer restricted to rows [r0, r1); vt = (lo, hi): the V^T rows that go with them
        (None: V^T is shared whole - ragged batches)."""

        def cut(v, lo=r0, hi=r1):
            if isinstance(v, ops.SplitIL):
                return v.rows_view(lo, hi)
            if isinstance(v, tuple):
                return tuple(None if t is None else t[lo:hi] for t in v)
            if isinstance(v, list):
                return [cut(t, lo, hi) for t in v]
            return v[lo:hi]
        out = {}
        for k, v in ws.items():
            if k == "rope":
                out[k] = v
            elif k == "vt16":
                out[k] = v if vt is None else cut(v, vt[0], vt[1])
            else:
                out[k] = cut(v)
        return out

    def evaluate(self, ctx: dict, step: int) -> torch.Tensor:
        """Run the network on ws['xin'] (rows: cond branch then null branch) at evaluation time #step.
        Returns ws['pred'] [Bt*T, dim_out]."""
        d, sd, ws = self.d, self.sd, ctx["ws"]
        Bt, T, M, rg = ctx["Bt"], ctx["T"], ctx["M"], ctx.get("ragged")
        dim = d["dim"]
        tab = ctx["table"][step]
        free: List[torch.Tensor] = list(ws["h"])
        take = free.pop
        # split activations need the f16x3 kernel on every consumer GEMM (K % 32 == 0 and more than 64 rows; the
        # single-term mode consumes 64 k per stage).  'f16' without split I/O falls back to the fp32 kernels.
        il = self.split_il.get
        if self.precision == "f16":
            split_io = M > 64 and dim % 64 == 0
            sp = self.split.get if split_io else (lambda k: None)
        else:
            split_io = self.precision == "f16x3" and M > 64 and dim % 32 == 0
            sp = self.split.get

        h = take()
        h0 = take()
        ops.gemm(ws["xin"], sd["to_embed.weight"][:, : d["dim_out"]], h0, residual=ws["base"])
        ops.dwconv31_gelu_res(h0, self.dw_w, sd["conv_embed.dw_conv1d.0.bias"], h, Bt, T, ragged=rg)
        free.append(h0)
        # residual-stream tensors that later feed a skip combiner (as x or as the popped skip) also get a split
        # twin, so that GEMM takes both operands pre-split (all-DMA kernel) instead of splitting on the fly
        twin = {id(b): pr for b, pr in zip(ws["h"], ws["h16"])} if split_io else None
        sp_step = ctx["sp"][step] if (split_io and "sp" in ctx) else None
        hp = ctx["hp"] if sp_step is not None else None
        pp = self.pred_scale.data_ptr() if sp_step is not None else None
        use_dn = split_io and M >= 2048 and ctx.get("dn") is not None
        if split_io:
            tw = twin[id(h)]
            h0s = ctx["dn"]["hsp"][0][0] if use_dn else hp      # (deferred-norm path: one pre-scale per stage of the stream)
            ops.split_act_f16(h, tw, scale=h0s) if isinstance(tw, ops.SplitIL) else ops.split_act_f16(h, *tw, scale=h0s)

        # every to_out / ff2 / skip-combiner product is followed by a norm of its output: one call (ops.gemm(norm=...)), so that
        # problems on the split-K path (one utterance) normalise inside the reduction
        # (2048 rows and more never split K: the call would run the same two kernels, so it stays two calls there)
        fuse_norm = split_io and M < 2048
        normed_ahead = False                 # the attention norm of the layer about to start was produced by the previous GEMM
        if use_dn:                           # (decided per call in prepare())
            return self._layers_pair_stream(ctx, step, ws, h, twin, free, Bt, T, M, rg)
        def tab_rows(i_, k_):
            return tab[(4 * i_ + k_) * dim:(4 * i_ + k_ + 1) * dim]
        skips: List[torch.Tensor] = []
        for i in range(d["depth"]):
            p = f"transformer.layers.{i}"
            g_attn = tab[(4 * i + 0) * dim:(4 * i + 1) * dim]
            b_attn = tab[(4 * i + 1) * dim:(4 * i + 2) * dim]
            g_ff = tab[(4 * i + 2) * dim:(4 * i + 3) * dim]
            b_ff = tab[(4 * i + 3) * dim:(4 * i + 4) * dim]
            if (p + ".0.weight") in sd:
                s = skips.pop()
                comb = take()
                if split_io:
                    nm = dict(gamma=g_attn, beta=b_attn, out_split=ws["normed16"],
                              scale=sp_step[i][0] if sp_step is not None else None) if fuse_norm else None
                    ops.gemm(h, sd[p + ".0.weight"], comb, bias=sd[p + ".0.bias"], a2=s, w_split=sp(p + ".0.weight"), w_il=il(p + ".0.weight"),
                             a_split=twin[id(h)], a2_split=twin[id(s)], a_scale=hp, norm=nm)
                    normed_ahead = nm is not None
                else:
                    ops.gemm(h, sd[p + ".0.weight"], comb, bias=sd[p + ".0.bias"], a2=s, w_split=sp(p + ".0.weight"))
                free += [h, s]
                h, keep_input = comb, False
            else:
                skips.append(h)
                keep_input = True
            if split_io:
                n16, a16, f16 = ws["normed16"], ws["att16"], ws["ff16"]
                # device pointers of this (evaluation, layer)'s activation pre-scales: normed, q|k, v, attention out, normed, ff
                s_na, s_qk, s_v, s_at, s_nf, s_ff = sp_step[i] if sp_step is not None else (None,) * 6
                if not normed_ahead:
                    ops.adarmsnorm(h, g_attn, b_attn, None, out_split=n16, split_scale=s_na)
                # q | k split row-major, v split + transposed, straight into the f16x3 attention (any T: sequences whose
                # length is not a multiple of 4 store their v columns 2 bytes at a time)
                ops.gemm(ws["normed"], sd[p + ".2.to_qkv.weight"], ws["qkv"], rope=ws["rope"], rope_cols=2 * d["heads"] * 64,
                         w_split=sp(p + ".2.to_qkv.weight"), w_il=il(p + ".2.to_qkv.weight"), a_split=n16, out_split=ws["qk16"], vt_split=ws["vt16"],
                         write_f32=False, a_scale=s_na, c_scale=s_qk, vt_scale=s_v)
                normed_ahead = False
                ops.attention_f16x3(ws["qk16"], ws["vt16"], None, Bt, T, d["heads"], 64 ** -0.5, out_split=a16,
                                    qk_scale=s_qk, v_scale=s_v, out_scale=s_at, ragged=rg)
                h_att = take() if keep_input else h
                ops.gemm(ws["att"], sd[p + ".2.to_out.weight"], h_att, residual=h, w_split=sp(p + ".2.to_out.weight"), w_il=il(p + ".2.to_out.weight"),
                         a_split=a16, a_scale=s_at, norm=dict(gamma=g_ff, beta=b_ff, out_split=n16, scale=s_nf) if fuse_norm else None)
                h = h_att
                if not fuse_norm:
                    ops.adarmsnorm(h, g_ff, b_ff, None, out_split=n16, split_scale=s_nf)
                ops.gemm(ws["normed"], sd[p + ".4.0.weight"], ws["ff"], bias=sd[p + ".4.0.bias"], act=ops.ACT_GELU,
                         w_split=sp(p + ".4.0.weight"), w_il=il(p + ".4.0.weight"), a_split=n16, out_split=f16, write_f32=False,
                         a_scale=s_nf, c_scale=s_ff)
                last = i + 1 == d["depth"]
                nm = None
                if fuse_norm and last:                   # the final RMSNorm in front of to_pred
                    nm = dict(gamma=sd["transformer.final_norm.gamma"], beta=None, out_split=ws["pred16"], scale=pp)
                elif fuse_norm and (f"transformer.layers.{i + 1}.0.weight") not in sd:      # next layer starts with its attention norm
                    nm = dict(gamma=tab_rows(i + 1, 0), beta=tab_rows(i + 1, 1), out_split=n16,
                              scale=sp_step[i + 1][0] if sp_step is not None else None)
                ops.gemm(ws["ff"], sd[p + ".4.2.weight"], h, bias=sd[p + ".4.2.bias"], residual=h,
                         w_split=sp(p + ".4.2.weight"), w_il=il(p + ".4.2.weight"), a_split=f16,
                         out_split=None if last else twin[id(h)], a_scale=s_ff, c_scale=None if last else hp, norm=nm)
                normed_ahead = nm is not None
                continue
            ops.adarmsnorm(h, g_attn, b_attn, ws["normed"])
            ops.gemm(ws["normed"], sd[p + ".2.to_qkv.weight"], ws["qkv"], rope=ws["rope"], rope_cols=2 * d["heads"] * 64,
                     w_split=sp(p + ".2.to_qkv.weight"))
            ops.attention(ws["qkv"], ws["att"], Bt, T, d["heads"], 64 ** -0.5, ragged=rg)
            h_att = take() if keep_input else h
            ops.gemm(ws["att"], sd[p + ".2.to_out.weight"], h_att, residual=h, w_split=sp(p + ".2.to_out.weight"))
            h = h_att
            ops.adarmsnorm(h, g_ff, b_ff, ws["normed"])
            ops.gemm(ws["normed"], sd[p + ".4.0.weight"], ws["ff"], bias=sd[p + ".4.0.bias"], act=ops.ACT_GELU,
                     w_split=sp(p + ".4.0.weight"))
            ops.gemm(ws["ff"], sd[p + ".4.2.weight"], h, bias=sd[p + ".4.2.bias"], residual=h, w_split=sp(p + ".4.2.weight"))
        if split_io:
            if not normed_ahead:
                ops.adarmsnorm(h, sd["transformer.final_norm.gamma"], None, None, out_split=ws["pred16"], split_scale=pp)
            ops.gemm(ws["normed"], sd["to_pred.weight"], ws["pred"], w_split=sp("to_pred.weight"), a_split=ws["pred16"], a_scale=pp,
                     w_il=il("to_pred.weight") if isinstance(ws["pred16"], ops.SplitIL) else None)
        else:
            ops.adarmsnorm(h, sd["transformer.final_norm.gamma"], None, ws["normed"])
            ops.gemm(ws["normed"], sd["to_pred.weight"], ws["pred"], w_split=sp("to_pred.weight"))
        return ws["pred"]


def evaluation_times(nfe: int, method: str):
    """Fixed grid of the reference solver (torchdiffeq fixed-grid, options.step_size) as python floats
    computed in fp32 exactly like the solver does: (t0, dt) per step, plus every time the field is evaluated at."""
    steps = nfe // 2 if method == "midpoint" else nfe
    if steps < 1 or (method == "midpoint" and nfe % 2):
        raise ValueError(f"nfe={nfe} is not valid for method={method}")
    h = torch.tensor(1.0 / steps, dtype=torch.float32)
    grid = torch.arange(0, steps + 1, dtype=torch.float32) * h
    grid[-1] = 1.0
    times, dts = [], []
    for a, b in zip(grid[:-1], grid[1:]):
        dt = b - a
        dts.append(float(dt))
        times.append(float(a))
        if method == "midpoint":
            times.append(float(a + 0.5 * dt))
    return torch.tensor(times, dtype=torch.float32), dts


class FlowMatchingSampler:
    """ConditionalFlowMatcherWrapper.sample (acoustic.py:597-688) on a VectorField.

    `nfe` counts CFG-combined field evaluations: the reference's ode_step_size=0.0625 midpoint
    setting is nfe=32 (16 steps x 2).  method='euler' (nfe steps) is an extra the reference lacks."""

    def __init__(self, field: VectorField, nfe: int = 32, method: str = "midpoint"):
        self.field, self.nfe, self.method = field, nfe, method

    # launch-bound regime (short / single utterances): the whole solve - ~2400 kernel launches for 32 NFE - is captured
    # once per input shape into a HIP graph and replayed (env CVX_GRAPH=0 disables, CVX_GRAPH_MAX_ROWS bounds the
    # batch size it applies to; large batches are GPU-bound and stay eager).
    GRAPH_CACHE = 4

    def _integrate(self, phoneme_ids, cond, y, times_dev, dts, s: float, use_null: bool, lengths=None) -> None:
        """prepare() + the fixed-grid loop; advances `y` in place (ragged batch: y, cond, ids packed, see prepare)."""
        f = self.field
        ctx = f.prepare(phoneme_ids, cond, times_dev, use_null, lengths=lengths, times_key=(int(self.nfe), str(self.method)))
        ws, M1 = ctx["ws"], ctx["M1"]
        xin = ws["xin"]
        x_c = xin[:M1]
        x_n = xin[M1:] if use_null else None
        x_c.copy_(y.reshape(M1, -1))
        if use_null:
            x_n.copy_(x_c)
        evaluate = f.evaluate
        e = 0
        for dt in dts:
            if self.method == "midpoint":
                pred = evaluate(ctx, e); e += 1
                ops.cfg_combine_axpy(pred[:M1], pred[M1:] if use_null else None, y, s, 0.5 * dt, x_c, x_n)
                pred = evaluate(ctx, e); e += 1
                ops.cfg_combine_axpy(pred[:M1], pred[M1:] if use_null else None, y, s, dt, y, x_c, x_n)
            else:
                pred = evaluate(ctx, e); e += 1
                ops.cfg_combine_axpy(pred[:M1], pred[M1:] if use_null else None, y, s, dt, y, x_c, x_n)
        return ctx

    @ops.gated
    @torch.no_grad()
    def sample(self, *, phoneme_ids: torch.Tensor, cond: torch.Tensor, mask: Optional[torch.Tensor] = None,
               cond_scale: float = 1.0, y0: Optional[torch.Tensor] = None) -> torch.Tensor:
        import os
        f, d = self.field, self.field.d
        dev = f.device
        if cond.ndim != 3 or cond.shape[-1] != d["dim_cond"]:
            raise AssertionError(f"cond must be [B,T,{d['dim_cond']}], got {tuple(cond.shape)}")
        expect_ids = cond.shape[:2] + ((d["streams"],) if d["streams"] > 1 else ())
        if tuple(phoneme_ids.shape) != tuple(expect_ids):
            raise AssertionError(f"phoneme_ids must be {tuple(expect_ids)}, got {tuple(phoneme_ids.shape)}")
        cond = cond.to(device=dev, dtype=torch.float32).contiguous()
        phoneme_ids = phoneme_ids.to(device=dev, dtype=torch.int64).contiguous()
        B, T, _ = cond.shape
        if y0 is None:                       # acoustic.py:647-650 (VoMix draws 80 channels)
            y0 = torch.randn(B, T, d["dim_out"], device=dev, dtype=torch.float32)
        if tuple(y0.shape) != (B, T, d["dim_out"]):
            raise AssertionError(f"y0 must be [B,T,{d['dim_out']}] = {(B, T, d['dim_out'])}, got {tuple(y0.shape)}")
        y = y0.to(device=dev, dtype=torch.float32).contiguous().clone()
        use_null = float(cond_scale) != 1.0          # acoustic.py:423
        s = float(cond_scale)
        times, dts = evaluation_times(self.nfe, self.method)
        rows = (2 * B if use_null else B) * T
        if os.environ.get("CVX_GRAPH", "1") == "0" or rows > int(os.environ.get("CVX_GRAPH_MAX_ROWS", "8192")):
            self._integrate(phoneme_ids, cond, y, ops.h2d(times, dev), dts, s, use_null)
            return y
        main = torch.cuda.current_stream()
        # (the saturation flag of the launching stream is a kernel argument of the captured launches: one graph per stream)
        key = (B, T, use_null, s, self.nfe, self.method, main.cuda_stream)
        cache = f.__dict__.setdefault("_graphs", {})
        ent = cache.get(key)
        if ent is None:
            st = dict(ids=phoneme_ids.clone(), cond=cond.clone(), y=y.clone(), times=ops.h2d(times, dev))
            ctx = self._integrate(st["ids"], st["cond"], st["y"], st["times"], dts, s, use_null)   # eager warm-up
            st["ws"] = ctx["ws"]             # keep this shape's workspace alive for as long as the graph is
            # (thread-local capture mode + the package's capture gate: another host thread driving its own model on this device
            #  must not break this capture, nor be broken by it - HIP rejects its synchronisations while a capture is open even
            #  in thread-local mode, so the capture waits until the package's other entry points have left; ops._CaptureGate)
            with ops.CAPTURE_GATE.exclusive():
                main.synchronize()
                g = torch.cuda.CUDAGraph()
                if getattr(self, "_cap", None) is None:
                    self._cap = torch.cuda.Stream(device=dev)
                ops.saturation_share(main, self._cap)    # the captured launches report into the calling stream's flag
                with torch.cuda.graph(g, stream=self._cap, capture_error_mode="thread_local"):
                    self._integrate(st["ids"], st["cond"], st["y"], st["times"], dts, s, use_null)
            if len(cache) >= self.GRAPH_CACHE:
                cache.pop(next(iter(cache)))
            ent = cache[key] = (g, st)
        g, st = ent
        st["ids"].copy_(phoneme_ids)
        st["cond"].copy_(cond)
        st["y"].copy_(y)
        g.replay()
        return st["y"].clone()

    @ops.gated
    @torch.no_grad()
    def sample_ragged(self, *, phoneme_ids: List[torch.Tensor], cond: List[torch.Tensor], cond_scale: float = 1.0,
                      y0: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
        """The same solve for several utterances of DIFFERENT length in one launch sequence: utterance i is
        phoneme_ids[i] [T_i(, streams)], cond[i] [T_i, dim_cond] (y0[i] [T_i, dim_out]); returns the list of [T_i, dim_out]
        results.  The reference runs such utterances one at a time (monologue_generation.py:259-304) and its network has
        no padding mask (acoustic.py:313), so they are PACKED back to back - M = sum T_i rows - and the three operators
        that look along time (attention, rotary positions, ConvPositionEmbed) work per sequence (include/covomix_hip.h,
        RAGGED BATCHES); every utterance gets the result of its own B = 1 run up to fp32 summation order."""
        f, d = self.field, self.field.d
        dev = f.device
        n = len(cond)
        if n == 0:
            return []
        lengths = [int(c.shape[0]) for c in cond]
        for i in range(n):
            if cond[i].ndim != 2 or cond[i].shape[1] != d["dim_cond"] or lengths[i] < 1:
                raise AssertionError(f"cond[{i}] must be [T,{d['dim_cond']}] with T >= 1, got {tuple(cond[i].shape)}")
            expect = (lengths[i],) + ((d["streams"],) if d["streams"] > 1 else ())
            if tuple(phoneme_ids[i].shape) != expect:
                raise AssertionError(f"phoneme_ids[{i}] must be {expect}, got {tuple(phoneme_ids[i].shape)}")
            if y0 is not None and tuple(y0[i].shape) != (lengths[i], d["dim_out"]):
                raise AssertionError(f"y0[{i}] must be {(lengths[i], d['dim_out'])}, got {tuple(y0[i].shape)}")
        cond_p = torch.cat([c.to(device=dev, dtype=torch.float32) for c in cond]).contiguous()
        ids_p = torch.cat([p.to(device=dev, dtype=torch.int64) for p in phoneme_ids]).contiguous()
        if y0 is None:                       # acoustic.py:647-650, one draw per utterance
            y = torch.randn(cond_p.shape[0], d["dim_out"], device=dev, dtype=torch.float32)
        else:
            y = torch.cat([t.to(device=dev, dtype=torch.float32) for t in y0]).contiguous().clone()
        use_null = float(cond_scale) != 1.0          # acoustic.py:423
        times, dts = evaluation_times(self.nfe, self.method)
        self._integrate(ids_p, cond_p, y, ops.h2d(times, dev), dts, float(cond_scale), use_null, lengths=lengths)
        return list(torch.split(y, lengths))

