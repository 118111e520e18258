"""BASELINE config 5 as a synthetic workload: CoMix text2semantic AR decode + VoMix 64-NFE + HiFi-GAN for batches of dialogues on
one GPU (recipe weights, seeded inputs - there are no trained weights upstream).  Used by tools/bench_pipeline.py, bench.py's `c5`
key and tests/test_pipeline_gpu.py; the generation scripts run the same two stages on real files (generation.py).

Stage 1 (text2semantic, reference dialogue_generation.py:297-320 comix_pred): `tokens` decoded steps per dialogue with the eos ignored so
that the work is fixed; up to 64 dialogues decode together (bit-identical to one by one, t2s.py) whatever the acoustic batch size is.
Stage 2 (reference :306-329 covomix + mel_decode_to_wav): prompt + predicted token assembly (assembly.py), the 64-NFE solve on the
[B, tokens + prompt, .] batch, HiFi-GAN on the generated frames, int16 cast."""
from __future__ import annotations

import contextlib
import sys

import torch

from . import assembly, ops, pipeline
from . import synthetic as syn


class Config5:
    def __init__(self, device, tokens: int = 608, prompt: int = 400, nfe: int = 64):
        from .conditional_model import CoVoMixModel
        from .t2s import TextToSemanticDecoder
        from .vocoder import AttrDict, Generator
        self.device = dev = torch.device(device)
        self.tokens, self.prompt, self.T = int(tokens), int(prompt), int(tokens) + int(prompt)
        with contextlib.redirect_stdout(sys.stderr):
            self.t2s = TextToSemanticDecoder({k: torch.from_numpy(v) for k, v in syn.t2s_state_dict(
                syn.t2s_param_shapes(two_output=True, dim=512, dim_target=1024), seed=0).items()}, dev)
            sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
            sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
            self.model = CoVoMixModel.from_state_dict(sd, nfe=nfe).eval().to(dev)
            self.gen = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)).to(dev)
            self.gen.load_state_dict({k: torch.from_numpy(v) for k, v in
                                      syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()})
            self.gen.eval()
            self.gen.remove_weight_norm()

    # ---- inputs of dialogue number j (seeded: the same whatever the schedule and the batching)
    def dialogue(self, j: int) -> dict:
        g = torch.Generator().manual_seed(5000 + j)
        text = torch.randint(1, 30000, (1, 64), generator=g)
        sem = torch.randint(0, 500, (2, self.prompt), generator=g)
        mel = (torch.randn(2, self.prompt, 80, generator=g) * 2.0 - 6.0).clamp(-11.52, 2.0)
        return dict(j=j, text=text, sem=sem, mel=mel)

    def stage1(self, group: list) -> list:
        """text2semantic of up to 64 dialogues together on the current stream -> their records + `streams` (int64 CPU [2, tokens])."""
        d = self.t2s.d
        unis = []
        for rec in group:
            gen = torch.Generator(device=self.device).manual_seed(7000 + rec["j"])
            unis.append(torch.rand(self.tokens, d["streams"], d["vocab"], device=self.device, generator=gen))
        res = self.t2s.generate_batch([rec["text"] for rec in group], unis, max_length=self.tokens, ignore_eos=True)
        streams = torch.stack([r[1] for r in res]).cpu()
        return [dict(rec, streams=streams[i]) for i, rec in enumerate(group)]

    def decode_ragged(self, n: int = 128, slots: int = 16, seed: int = 3) -> dict:
        """Stage 1 alone on dialogues that END at different steps - real dialogues sample their eos anywhere (text2semantic.py:803-818);
        here: step limits spread over [100, tokens], the eos itself ignored so that the work is a function of the seed.  Useful
        tokens/s of the continuously refilled slots (t2s.generate_many) against the same slots in lock step (generate_batch: a batch
        runs until its longest dialogue ends) and against the fixed-length case (every dialogue `tokens` steps)."""
        import time
        g = torch.Generator().manual_seed(seed)
        lims = torch.randint(100, self.tokens + 1, (n,), generator=g).tolist()
        srcs = [torch.randint(1, 30000, (1, 64), generator=g) for _ in range(n)]
        t2s = self.t2s

        def timed(fn):
            torch.cuda.synchronize(self.device); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        t2s.generate_many(srcs, max_length=self.tokens, slots=slots, ignore_eos=True, limits=[20] * n)       # graph + buffers of the timed shape
        t_many = timed(lambda: t2s.generate_many(srcs, max_length=self.tokens, slots=slots, ignore_eos=True, limits=lims))
        t_lock = timed(lambda: [t2s.generate_batch(srcs[w:w + slots], max_length=max(lims[w:w + slots]), ignore_eos=True)
                                for w in range(0, n, slots)])
        t_fix = timed(lambda: [t2s.generate_batch(srcs[w:w + slots], max_length=self.tokens, ignore_eos=True) for w in range(0, n, slots)])
        useful = sum(lims)
        return {"dialogues": n, "slots": slots, "steps_per_dialogue": f"100..{self.tokens}, {useful} in total",
                "continuous_tokens_per_s": round(useful / t_many, 1), "lock_step_tokens_per_s": round(useful / t_lock, 1),
                "fixed_length_tokens_per_s": round(n * self.tokens / t_fix, 1),
                "continuous_vs_lock_step": round(t_lock / t_many, 3), "continuous_vs_fixed_length": round((useful / t_many) / (n * self.tokens / t_fix), 3)}

    def _solve(self, batch: list) -> torch.Tensor:
        """assembly, 64-NFE solve, vocoder, int16 cast of one batch on the current stream -> PCM on the device"""
        ids, cond, mask, y0 = [], [], [], []
        dev = self.device
        for rec in batch:
            sem, mel = rec["sem"], rec["mel"]
            a, b, c = assembly.build_dialogue_inputs(sem[0], sem[1], rec["streams"][0], rec["streams"][1], mel[0], mel[1])
            ids.append(a); cond.append(b); mask.append(c)
            y0.append(torch.randn(self.T, 80, device=dev, generator=torch.Generator(device=dev).manual_seed(9000 + rec["j"])))
        sampled = self.model.synthesis_sample(ops.h2d(torch.stack(ids), dev), ops.h2d(torch.stack(cond), dev), torch.stack(mask), 0.7,
                                              y0=torch.stack(y0))
        mel = sampled[:, self.prompt:, :].permute(0, 2, 1).contiguous()            # the generated frames (monologue_generation.py:299-300)
        return ops.wav_to_int16(self.gen(mel).squeeze(1).contiguous())

    def stage2_launch(self, batch: list):
        """acoustic solve + vocoder of one batch of decoded dialogues ENQUEUED on the current stream -> handle for stage2_finish: the PCM
        and the saturation flag travel into pinned memory behind an event, the host does not wait."""
        with ops.saturation_deferred(read=False):
            pcm_dev = self._solve(batch)
        flag = ops.saturation_snapshot()
        pcm = torch.empty(pcm_dev.shape, dtype=pcm_dev.dtype, pin_memory=True)
        pcm.copy_(pcm_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return batch, pcm, flag, ev

    def stage2_finish(self, handle) -> list:
        batch, pcm, flag, ev = handle
        ev.synchronize()
        if flag is not None and int(flag[0]) != 0:          # saturated: repeat with the per-call checks (fp32 re-run or raise)
            pcm = self._solve(batch).cpu()
        return [dict(j=rec["j"], streams=rec["streams"], pcm=pcm[i].clone()) for i, rec in enumerate(batch)]

    def stage2(self, batch: list) -> list:
        """acoustic solve + vocoder of one batch of decoded dialogues on the current stream -> records dict(j, streams, pcm int16 CPU)."""
        pcm = self._solve(batch).cpu()
        return [dict(j=rec["j"], streams=rec["streams"], pcm=pcm[i]) for i, rec in enumerate(batch)]

    def run(self, n_dialogues: int, B: int, overlap: bool, partitioned: bool = True, first: int = 0, B1: int = 8, recs=None) -> list:
        """n_dialogues dialogues: text2semantic in groups of B1, the solve + vocoder in batches of B -> one record per dialogue, in
        order.  partitioned: on the CU partition's two streams (overlap: pipelined / alternately); otherwise everything on the
        current stream, group by group (the round-4 schedule)."""
        if recs is None:                     # (recs: the dialogues' inputs made by the caller - outside its timed region)
            recs = [self.dialogue(first + j) for j in range(n_dialogues)]
        groups = [recs[i:i + B1] for i in range(0, n_dialogues, B1)]
        if partitioned and overlap:
            # (the host runs one acoustic batch ahead of the device: stage2_finish(k) after stage2_launch(k + 1))
            res = pipeline.run_two_stage(groups, self.stage1, self.stage2_launch, self.device, overlap=True, collate=pipeline.regroup(B),
                                         finish=self.stage2_finish)
        elif partitioned:
            res = pipeline.run_two_stage(groups, self.stage1, self.stage2, self.device, overlap=False, collate=pipeline.regroup(B))
        else:
            # one stream, the reference's order of stages; the host still runs one acoustic batch ahead of the device
            res, pend = [], None
            for b in pipeline.regroup(B)(self.stage1(g) for g in groups):
                h = self.stage2_launch(b)
                if pend is not None:
                    res.append(self.stage2_finish(pend))
                pend = h
            if pend is not None:
                res.append(self.stage2_finish(pend))
            torch.cuda.synchronize(self.device)
        return [r for batch in res for r in batch]
