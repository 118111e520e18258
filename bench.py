#!/usr/bin/env python3
"""mel-frames/sec for the CoVoMix hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch per GPU:
    VoMix 2-speaker acoustic model, 32 NFE (16 midpoint steps x 2, CFG scale 0.7 => 64 network
    forwards) on B=8 utterances of T=1000 frames (dual-channel 160-dim prompt of 400 frames)
    followed by the HiFi-GAN generator on all 8 x 1000 generated frames and the int16 cast.
Weights: reference-shaped random checkpoints from covomix_amd.synthetic (no pretrained weights
exist); inputs resident in HBM before the timed region.  N > 1: one process per GPU - under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, started from a plain shell, self-launched
(`python bench.py --gpus 8` re-executes itself under torch.distributed.run with a free rendezvous port);
utterances sharded (8 per rank, weak scaling), weights broadcast once over RCCL, no data-path collective.

    python bench.py [--gpus N] [--steps K] [--warmup W]
Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B, T, PROMPT, NFE, COND_SCALE = 8, 1000, 400, 32, 0.7
_JSON_FD = None                                         # the process's original stdout once main() has pointed fd 1 at stderr
FLOP_PER_FRAME = 64 * 255_784_960 + 281_398_000        # 16.65 GFLOP per mel frame
PEAK_F32_MFMA = 157.3e12                                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA = 2.5e15                                  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (not the 2:1 sparse figure)


class GemmTimer:
    """HIP-event bracket around every GEMM launch of the dominant kernel class (events are recorded on torch's
    current stream = the stream the kernel is launched on).  Sums algorithmic FLOPs (2*M*N*K) and durations.
    `dominant(a, w, kw)` says whether a launch goes to the dominant kernel (f16x3: the pre-split all-DMA 256x256
    kernel, i.e. every per-layer transformer GEMM except the skip combiners; fp32: every GEMM)."""

    def __init__(self, split: bool, every: int = 1):
        self.split = split
        # bracket every n-th dominant launch: the event packets themselves cost queue time (all launches bracketed: -1.3 %
        # on `value`, same average).  7 is coprime with the 36-GEMM cycle of an evaluation, so every shape is sampled alike.
        self.every = max(1, every)
        self.seen = 0
        self.pairs = []
        self.flops = 0.0
        self.launches = 0
        self.other_launches = 0

    def dominant(self, a, w, kw):
        if not self.split:
            return True
        return kw.get("a_split") is not None and a.shape[0] >= 2048 and w.shape[0] >= 512

    def install(self, ops):
        inner = ops.gemm

        def timed(a, w, out, **kw):
            if not self.dominant(a, w, kw):
                self.other_launches += 1
                return inner(a, w, out, **kw)
            self.seen += 1
            if self.seen % self.every:
                return inner(a, w, out, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = inner(a, w, out, **kw)
            e.record()
            k = a.shape[1] + (kw["a2"].shape[1] if kw.get("a2") is not None else 0)
            self.pairs.append((s, e))
            self.flops += 2.0 * a.shape[0] * w.shape[0] * k
            self.launches += 1
            return r
        ops.gemm = timed
        self._ops, self._inner = ops, inner

    def remove(self):
        self._ops.gemm = self._inner

    def result(self):
        secs = sum(s.elapsed_time(e) for s, e in self.pairs) * 1e-3
        return self.flops, secs, self.launches, self.seen


class ClassTimer:
    """HIP-event brackets around every n-th call of one Python-level op (a kernel class of the step: attention, the stand-alone
    AdaptiveRMSNorm launches, the whole vocoder call) -> ms per step next to the dominant GEMM's, so that a slower box or a
    slower kernel class can be told apart from the bench line alone."""

    def __init__(self, owner, name, every):
        self.owner, self.name, self.every = owner, name, max(1, every)
        self.inner = getattr(owner, name)
        self.seen, self.pairs = 0, []

        def timed(*a, **kw):
            self.seen += 1
            if self.seen % self.every:
                return self.inner(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.inner(*a, **kw)
            e.record()
            self.pairs.append((s, e))
            return r
        setattr(owner, name, timed)

    def remove(self):
        setattr(self.owner, self.name, self.inner)

    def result(self, steps):
        if not self.pairs:
            return None
        avg_ms = sum(s.elapsed_time(e) for s, e in self.pairs) / len(self.pairs)
        return {"launches_per_step": round(self.seen / steps, 1), "timed": len(self.pairs), "avg_launch_ms": round(avg_ms, 4),
                "ms_per_step": round(avg_ms * self.seen / steps, 3)}


class PowerSampler:
    """Package power and shader clock of the GPU while the timed region runs (a host thread reading the amdgpu hwmon files, or
    `rocm-smi --showpower --showclocks` when there are none): the launch times of the MFMA-bound kernels are joules / power cap on
    this chip, and boxes of the pool differ - without these two numbers a slower bench line cannot be told
    from a slower box.  The node exposes the hwmon directories of ALL its GPUs, also of those this process cannot see: every card
    is sampled and the one that drew the most power during the timed region is this process's GPU (the others idle at ~250 W)."""

    def __init__(self, period=0.1, device_index=0):
        import glob
        import threading
        self.period, self.stop_ev = period, threading.Event()
        self.cards = []
        self.mine = None                       # PCI address of this process's GPU, when torch tells it
        try:
            pr = torch.cuda.get_device_properties(device_index)
            self.mine = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pci = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(h))))
            if self.mine and pci != self.mine and any(os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(x)))) == self.mine
                                                      for x in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                continue
            for pw in ("power1_average", "power1_input"):
                if os.path.isfile(os.path.join(h, pw)):
                    self.cards.append((h, pw))
                    break
        self.samples = {h: [] for h, _ in self.cards}
        self.smi = "/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None
        self.smi_samples = []
        self.thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _num(path, scale):
        try:
            return int(open(path).read()) / scale
        except Exception:
            return None

    def _read_smi(self):
        import re
        import subprocess
        out = subprocess.run([self.smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        p = re.search(r"Power \(W\): ([\d.]+)", out)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        return (float(p.group(1)) if p else None), (float(c.group(1)) if c else None)

    def _run(self):
        while not self.stop_ev.is_set():
            t = time.perf_counter()
            try:
                if self.cards:
                    for h, pw in self.cards:
                        p = self._num(os.path.join(h, pw), 1e6)
                        if p is not None:
                            self.samples[h].append((t, p, self._num(os.path.join(h, "freq1_input"), 1e6)))
                elif self.smi:
                    p, f = self._read_smi()
                    if p is not None:
                        self.smi_samples.append((t, p, f))
                else:
                    return
            except Exception:
                pass
            self.stop_ev.wait(self.period)

    def start(self):
        self.thread.start()

    def stop(self, t0, t1):
        self.stop_ev.set()
        self.thread.join(timeout=15)
        src, cap, sel = None, None, []
        if self.cards:
            best = None
            for h, pw in self.cards:
                w = [x for x in self.samples[h] if t0 <= x[0] <= t1]
                if w and (best is None or sum(x[1] for x in w) / len(w) > best[0]):
                    best = (sum(x[1] for x in w) / len(w), h, pw, w)
            if best:
                _, h, pw, sel = best
                src = f"hwmon {pw} of {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(h))))} ({'PCI ' + self.mine if self.mine and len(self.cards) == 1 else 'busiest of %d cards' % len(self.cards)})"
                cap = self._num(os.path.join(h, "power1_cap"), 1e6)
        elif self.smi_samples:
            sel = [x for x in self.smi_samples if t0 <= x[0] <= t1] or self.smi_samples[-1:]
            src = "rocm-smi"
        if not sel:
            return {"power_w": None, "sclk_mhz": None, "power_cap_w": None, "power_samples": 0, "power_source": src}
        pw_ = [x[1] for x in sel]
        ck = [x[2] for x in sel if x[2]]
        return {"power_w": round(sum(pw_) / len(pw_), 1), "power_w_max": round(max(pw_), 1),
                "sclk_mhz": round(sum(ck) / len(ck), 1) if ck else None, "power_cap_w": cap,
                "power_samples": len(sel), "power_source": src}


def config2(dev, calls: int = 20):
    """BASELINE config 2 next to the headline: VoSingle, 32 NFE, ONE 500-frame utterance (200-frame prompt) per call - the
    reference's own calling pattern (monologue_generation.py:259-304).  ms per utterance, frames/s and the fraction of the dense
    fp16 MFMA peak its algorithmic FLOPs (15.18 GFLOP per frame) reach."""
    import covomix_amd.synthetic as syn
    from covomix_amd.conditional_model import CoVoMixModel
    shapes = syn.acoustic_param_shapes(dim_cond=80, streams=1)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    model = CoVoMixModel.from_state_dict(sd, nfe=NFE).eval().to(dev)
    inp = syn.synthetic_inputs("vosingle", 1, 500, 200, seed=1234)
    ids, cond, mask = inp["phoneme_ids"].to(dev), inp["cond"].to(dev), inp["mask"].to(dev)
    for _ in range(2):
        model.synthesis_sample(ids, cond, mask, COND_SCALE)

    def loop():
        for _ in range(calls):
            model.synthesis_sample(ids, cond, mask, COND_SCALE)
    _, dt, pw = timed_region(loop, dev.index or 0)
    dt /= calls
    flop_per_frame = 64 * 237_139_968
    # config 2 is ~2,400 dependent launches of 5-40 us per call: besides the shader clock of a lightly loaded chip (sclk_mhz above), the
    # box's cost of one dependent launch inside a graph decides it - measured here on a chain of 2,000 one-element kernels
    x = torch.zeros(1, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        x.add_(1.0)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(2000):
                x.add_(1.0)
        gr.replay(); side.synchronize()
        t0 = time.perf_counter(); gr.replay(); side.synchronize()
        floor_us = (time.perf_counter() - t0) / 2000 * 1e6
    pw["dependent_launch_us"] = round(floor_us, 3)
    return {"workload": "VoSingle 32-NFE, B=1, T=500 (200-frame prompt), acoustic model only", "calls": calls,
            "ms_per_utterance": round(dt * 1e3, 3), "frames_per_s": round(500 / dt, 1),
            "frac": round(500 / dt * flop_per_frame / PEAK_F16_MFMA, 4), "vs_f32_mfma_peak": round(500 / dt * flop_per_frame / PEAK_F32_MFMA, 4), **pw}


def timed_region(fn, dev_index=0):
    """fn() between two device synchronisations -> (result, seconds, {power_w, sclk_mhz, ...} of THAT region): boxes of the pool
    differ by several per cent, and every figure of the line must say which box state it was measured in."""
    from covomix_amd import ops
    power = PowerSampler(device_index=dev_index)
    power.start()
    torch.cuda.synchronize(); c0 = ops.clock_stamps(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); t1 = time.perf_counter()
    c1 = ops.clock_stamps(); torch.cuda.synchronize()
    info = power.stop(t0, t1)
    out = {k: info.get(k) for k in ("power_w", "sclk_mhz", "power_samples")}
    out.update(effective_clock(c0, c1))
    return r, t1 - t0, out


def effective_clock(c0, c1) -> dict:
    """Shader clock of a region from two ops.clock_stamps(): cycles counted / real time on every CU, per XCD (`sclk_mhz` is the hwmon
    file: ONE XCD's momentary value, sampled every 0.1 s).  Lightly loaded configurations run at whatever clock a box's power management
    grants each XCD: this is the figure that tells a slow box from a slow kernel there."""
    from covomix_amd import ops
    r = ops.clock_from_stamps(c0, c1)
    if not r:
        return {"sclk_eff_mhz": None}
    return {"sclk_eff_mhz": round(r["mhz"], 1), "sclk_eff_mhz_xcd_min_max": [round(min(r["xcd_mhz"]), 1), round(max(r["xcd_mhz"]), 1)],
            "sclk_eff_cus": r["cus"]}


def config5(dev, dialogues: int = 56):
    """BASELINE config 5 next to the headline (extra key `c5`, N = 1 only): CoMix text2semantic AR decode (608 steps per dialogue, eos
    ignored so the work is fixed) + VoMix 64-NFE + HiFi-GAN on 1008-frame dialogues (covomix_amd/config5.py), dialogues/s of
      serial_64 - THE SCHEDULE THE CLI RUNS (`--pipeline auto`): 64 dialogues per text2semantic pass on the whole chip (the decode is a
                  latency chain whose step time barely depends on the batch: 4.7 ms per dialogue at 64 slots, 20 at 8), then 8 dialogues
                  per acoustic batch, everything on one stream in the reference's order of stages (dialogue_generation.py:272-329);
      serial_8  - the same with 8 dialogues per decode pass (the round-5 serial schedule; tokens and PCM must be IDENTICAL);
      pipelined - the decode of the next 8 dialogues on a CU-masked side stream UNDER the solve of the current batch (7 per acoustic
                  batch: whole GEMM rounds on the 224 CUs the solve keeps; covomix_amd/pipeline.py) - the round-5 schedule,
    and `ragged_eos`: the decode stage alone on dialogues that END at different steps (100 ... 608) through continuously refilled decode
    slots (t2s.generate_many) against the same slots in lock step and against the fixed-length case."""
    from covomix_amd.config5 import Config5
    c5 = Config5(dev)
    c5.run(8, 8, overlap=False, partitioned=False)            # warm-up of every shape on every stream
    c5.run(64, 8, overlap=False, partitioned=False, B1=64)
    c5.run(14, 7, overlap=True)
    idx = dev.index or 0
    recs = [c5.dialogue(j) for j in range(dialogues)]          # synthetic inputs: made outside the timed regions
    big, tb, pb = timed_region(lambda: c5.run(dialogues, 8, overlap=False, partitioned=False, recs=recs, B1=64), idx)
    ser, ts, ps = timed_region(lambda: c5.run(dialogues, 8, overlap=False, partitioned=False, recs=recs, B1=8), idx)
    n_pip = min(dialogues, 28)
    pip, tp, pp = timed_region(lambda: c5.run(n_pip, 7, overlap=True, recs=recs[:n_pip]), idx)
    # the acoustic stage alone (8 decoded dialogues: 64-NFE solve + vocoder + int16): what a free decode would leave
    decoded = [dict(recs[i], streams=big[i]["streams"]) for i in range(8)]
    _, t_solve, _ = timed_region(lambda: c5.stage2(decoded), idx)
    same = all(torch.equal(a["streams"], b["streams"]) and torch.equal(a["pcm"], b["pcm"]) for a, b in zip(ser, big))
    tok = all(torch.equal(a["streams"], b["streams"]) for a, b in zip(ser, pip))
    from covomix_amd import ops
    part = ops.cu_partition(dev)
    return {"workload": f"CoMix text2semantic (608 steps) + VoMix 64-NFE + HiFi-GAN, T = {c5.T} frames per dialogue, recipe weights",
            "best_schedule": "serial_64: 64 dialogues per text2semantic pass on the whole chip, 8 per acoustic batch, one stream (= --pipeline auto)",
            "dialogues_per_s": round(dialogues / tb, 3), "dialogues": dialogues, "mel_frames_per_s": round(dialogues * c5.T / tb, 1),
            "solve_only_bound_dialogues_per_s": round(8 / t_solve, 3), "frac_of_solve_only_bound": round((dialogues / tb) / (8 / t_solve), 4),
            "serial_64": dict(dialogues_per_s=round(dialogues / tb, 3), dialogues=dialogues, **pb),
            "serial_8": dict(dialogues_per_s=round(dialogues / ts, 3), dialogues=dialogues, **ps),
            "pipelined_8": dict(dialogues_per_s=round(n_pip / tp, 3), dialogues=n_pip, cu_partition={"main": part.n_main, "side": part.n_side}, **pp),
            "serial_64_tokens_and_pcm_equal_serial_8": bool(same), "pipelined_tokens_equal_serial_8": bool(tok),
            "ragged_eos": c5.decode_ragged()}


def config1(dev, calls: int = 150):
    """BASELINE config 1's shape on the GPU (extra key `c1`, N = 1 only): HiFi-GAN config_covomix on ONE unbatched [80, 1000] mel - what
    the reference's scripts call per utterance (monologue_generation.py:300 -> mel_decode_to_wav :52-59), int16 cast included.
    ms per call, mel frames/s and the fraction of the dense fp16 MFMA peak its 281.4 MFLOP per frame reach."""
    import covomix_amd.synthetic as syn
    from covomix_amd import ops
    from covomix_amd.vocoder import AttrDict, Generator
    gen = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)).to(dev)
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()})
    gen.eval(); gen.remove_weight_norm()
    g = torch.Generator().manual_seed(1234)
    mel = (torch.randn(80, 1000, generator=g) * 2.0 - 6.0).clamp(-11.52, 2.0).to(dev)          # C1 input statistics

    def call():
        return ops.wav_to_int16(gen(mel).squeeze(0).contiguous())
    for _ in range(3):
        pcm = call()
    assert pcm.shape == (160 * 1000 + 32,)

    def loop():
        for _ in range(calls):
            call()
    _, dt, pw = timed_region(loop, dev.index or 0)
    dt /= calls
    return {"workload": "HiFi-GAN config_covomix, ONE unbatched [80, 1000] mel -> int16 PCM (monologue_generation.py:300)", "calls": calls,
            "ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(1000 / dt, 1), "frac": round(1000 / dt * 281_398_000 / PEAK_F16_MFMA, 5),
            "vs_f32_mfma_peak": round(1000 / dt * 281_398_000 / PEAK_F32_MFMA, 4), **pw}


def make_models(dev, rank, world, precision=None):
    import covomix_amd.synthetic as syn
    from covomix_amd import dp
    from covomix_amd.conditional_model import CoVoMixModel
    from covomix_amd.vocoder import AttrDict, Generator
    shapes = syn.acoustic_param_shapes()                       # VoMix: dim 1024, depth 8, 16 heads, E_in 2288
    vshapes = syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG)
    if rank == 0:
        sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
        vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(vshapes, seed=0).items()}
    else:
        sd = {k: torch.empty(v, dtype=torch.float32) for k, v in shapes.items()}
        vsd = {k: torch.empty(v, dtype=torch.float32) for k, v in vshapes.items()}
    cpu_sd = (sd, vsd) if rank == 0 else None
    if world > 1:                                              # ONE start-up broadcast over RCCL/xGMI
        sd = dp.broadcast_state_dict(sd, dev, src=0)
        vsd = dp.broadcast_state_dict(vsd, dev, src=0)
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    model = CoVoMixModel.from_state_dict(sd, nfe=NFE, precision=precision).eval().to(dev)
    gen = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG)).to(dev)
    gen.load_state_dict({k: v.cpu() for k, v in vsd.items()})
    gen.eval()
    gen.remove_weight_norm()
    return model, gen, cpu_sd


def host_cores() -> int:
    """CPU cores this process may actually use: min(affinity mask, cgroup-v2 cpu.max quota).
    (The GPU box exposes 256 hardware threads but a 16-CPU quota; 256 torch threads ran 80x slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cpu_sd):
    """The oracle (== the reference's PyTorch path, pinned <= 1e-5 in tests/golden) on this box's host cores AT THE METRIC
    CONFIGURATION (B=8 utterances x T=1000 frames): bounded sample = 2 timed CFG evaluations (4 network forwards on
    8 x 1000 frames, after a B=1 warm-up), each standing for one of the 32 NFE, + one HiFi-GAN call on the 8 x 1000
    frames.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import covomix_oracle as orc
    import covomix_amd.synthetic as syn
    sd, vsd = cpu_sd
    sd = dict(sd)
    sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    cores = host_cores()
    torch.set_num_threads(cores)
    inp = syn.synthetic_inputs("vomix", B, T, PROMPT, seed=1234)
    tm = torch.tensor(0.25)
    with torch.inference_mode():
        orc.forward_with_cond_scale(sd, inp["y0"][:1], tm, inp["phoneme_ids"][:1], inp["cond"][:1], COND_SCALE)   # warm-up (B=1)
        t0 = time.perf_counter()
        n_eval = 2
        for _ in range(n_eval):
            orc.forward_with_cond_scale(sd, inp["y0"], tm, inp["phoneme_ids"], inp["cond"], COND_SCALE)
        t_eval = (time.perf_counter() - t0) / n_eval
        folded = orc.fold_weight_norm(vsd)
        mel = inp["cond"][:, :, :80].transpose(1, 2).contiguous()
        t0 = time.perf_counter()
        orc.hifigan_forward(folded, syn.HIFIGAN_COVOMIX_CONFIG, mel)
        t_voc = time.perf_counter() - t0
    fps = B * T / (NFE * t_eval + t_voc)
    return {"value": round(fps, 2), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU restatement of the reference PyTorch path) at the metric config B={B} x T={T}: {n_eval} timed "
                      f"CFG evals ({t_eval:.3f} s each, x{NFE} = the 32-NFE solve) + 1 HiFi-GAN call on {B} x {T} frames "
                      f"({t_voc:.3f} s), torch {torch.get_num_threads()} threads"}


def fp32_exact(dev, rank, world, steps: int = 2):
    """The same step with precision='fp32' (v_mfma_f32_32x32x2_f32 everywhere): the exact-fp32 number that backs the
    precision statement of `dtype` (extra key of the JSON line, N=1 only, a few steps)."""
    import covomix_amd.synthetic as syn
    from covomix_amd import ops
    from covomix_amd.vocoder import AttrDict, Generator
    model, _, _ = make_models(dev, rank, world, "fp32")
    gen = Generator(AttrDict(syn.HIFIGAN_COVOMIX_CONFIG), precision="fp32").to(dev)
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(syn.HIFIGAN_COVOMIX_CONFIG), seed=0).items()}
    gen.load_state_dict(vsd); gen.eval(); gen.remove_weight_norm()
    inp = syn.synthetic_inputs("vomix", B, T, PROMPT, seed=1234 + rank)
    ids, cond, mask = inp["phoneme_ids"].to(dev), inp["cond"].to(dev), inp["mask"].to(dev)

    def step():
        mel = model.synthesis_sample(ids, cond, mask, COND_SCALE, y0=torch.randn(B, T, 80, device=dev))
        return ops.wav_to_int16(gen(mel.permute(0, 2, 1).contiguous()).squeeze(1).contiguous())
    step()

    def loop():
        for _ in range(steps):
            step()
    _, dt, pw = timed_region(loop, dev.index or 0)
    return {"value": round(B * T * steps / dt, 2), "unit": "mel-frames/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
            "dtype": "f32 (v_mfma_f32_32x32x2_f32 GEMM / attention / vocoder)", **pw}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-exact", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="skip the BASELINE config-2 figure (extra key `c2`, N = 1 only)")
    ap.add_argument("--no-c1", action="store_true", help="skip the BASELINE config-1 figure (extra key `c1`, N = 1 only)")
    ap.add_argument("--no-c5", action="store_true", help="skip the BASELINE config-5 figure (extra key `c5`, N = 1 only)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): 8 utterances per GPU per step.  strong: BASELINE config 4 literally - 64 utterances per step "
                         "dealt over the N ranks (N must divide 8), i.e. 64/N per GPU in batches of 8")
    ap.add_argument("--precision", choices=["f16x3", "f16", "fp32"], default=None,
                    help="default f16x3 (fp32-class); f16 = opt-in single-term fp16 operands (<= 1e-3 rel-L2 budget)")
    args = ap.parse_args()

    from covomix_amd import dp, ops
    import covomix_amd.synthetic as syn
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False): covomix_amd has no CPU path")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started from a plain shell: become the launcher of N ranks (one process per GPU, free rendezvous port) - the
        # reference's own multi-GPU entry point spawns its ranks itself too (hifi-gan/train.py:268-278)
        sys.exit(dp.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    # stdout carries ONE JSON line and nothing else: C++ libraries write there too (gloo prints "[Gloo] Rank r is connected to n peer
    # ranks" on every rank when its mesh comes up) - file descriptor 1 points at stderr from here on and the line goes to the saved one
    sys.stdout.flush()
    global _JSON_FD
    json_fd = _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    rank, world, local = dp.init_from_env("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import contextlib
    t_load = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):          # keep stdout to the single JSON line
        model, gen, cpu_sd = make_models(dev, rank, world, args.precision)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t_load                 # weight generation (rank 0) + ONE bucketed RCCL broadcast + packing

    inp = syn.synthetic_inputs("vomix", B, T, PROMPT, seed=1234 + rank)
    ids, cond, mask = inp["phoneme_ids"].to(dev), inp["cond"].to(dev), inp["mask"].to(dev)
    noise = torch.Generator(device=dev).manual_seed(1234 + rank)

    voc_events = []

    def one_batch():
        y0 = torch.randn(B, T, 80, device=dev, generator=noise)
        mel = model.synthesis_sample(ids, cond, mask, COND_SCALE, y0=y0)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        wav = gen(mel.permute(0, 2, 1).contiguous())
        ev[1].record()
        voc_events.append(ev)
        return ops.wav_to_int16(wav.squeeze(1).contiguous())

    strong = args.scaling == "strong"
    if strong and (64 // B) % world != 0:
        raise SystemExit(f"--scaling strong deals {64 // B} batches of {B} utterances over the ranks: --gpus must divide {64 // B}")
    batches_per_step = (64 // B) // world if strong else 1

    def step():
        for _ in range(batches_per_step):
            pcm_ = one_batch()
        return pcm_

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = GemmTimer(split=(model.precision in ("f16x3", "f16")), every=int(os.environ.get("CVX_BENCH_TIMER_EVERY", "7")))
    timer.install(ops)
    every = int(os.environ.get("CVX_BENCH_TIMER_EVERY", "7"))
    classes = {"attention": ClassTimer(ops, "attention_f16x3" if model.precision != "fp32" else "attention", every),
               "adarmsnorm": ClassTimer(ops, "adarmsnorm", every)}
    voc_events.clear()
    power = PowerSampler(device_index=local) if rank == 0 else None
    if power:
        power.start()
    barrier()
    clk0 = ops.clock_stamps()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu0 = time.process_time()                           # CPU seconds of this process, all its threads
    for _ in range(args.steps):
        pcm = step()
    barrier()
    t1 = time.perf_counter()
    clk1 = ops.clock_stamps()
    torch.cuda.synchronize()
    host_cpu = time.process_time() - cpu0
    elapsed = t1 - t0
    timer.remove()
    for c in classes.values():
        c.remove()
    power_info = power.stop(t0, t1) if power else {}
    my_elapsed = elapsed
    assert pcm.shape == (B, 160 * T + 32) and pcm.dtype == torch.int16
    frames, elapsed = dp.reduce_metric(float(B * T * args.steps * batches_per_step), elapsed, dev)
    per_rank = [my_elapsed]
    loads = [t_load]
    cpus = [host_cpu]
    if world > 1:                                        # diagnosis of the first hardware scaling runs: who was slow, and where
        allb = dp.gather_floats([my_elapsed, t_load, host_cpu], dev)
        per_rank = [b[0] for b in allb]
        loads = [b[1] for b in allb]
        cpus = [b[2] for b in allb]

    if rank == 0:
        value = frames / elapsed
        flops, gemm_s, launches, all_launches = timer.result()
        achieved = flops / gemm_s / 1e12
        split = model.precision in ("f16x3", "f16")
        terms = {"f16x3": 3, "f16": 1, "fp32": 1}[model.precision]
        p8 = model.precision == "f16x3" and ops._GEMM_FLAGS == 0
        kname = ("gemm_f16x3_p8s_kernel" if p8 else "gemm_f16x3_dma_kernel") if split else "gemm_f32_glds_kernel"
        mfma = "v_mfma_f32_16x16x32_f16" if p8 else "v_mfma_f32_32x32x16_f16"
        peak = PEAK_F16_MFMA if split else PEAK_F32_MFMA
        traffic = None
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.isfile(pmc) and model.precision == "f16x3":      # the committed PMC passes profile the default precision
            try:
                traffic = json.load(open(pmc)).get(kname, {}).get("hbm_bytes_per_launch")
                # NOT measured by this run: PMC counters need their own rocprofv3 --pmc passes (tools/archive/collect_profiles.sh)
                traffic_source = "profiles/pmc_summary.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run)"
            except Exception:
                traffic = None
        out = {
            "metric": "mel-frames/sec (VoMix 32-step + HiFi-GAN, Bx1000x80)",
            "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f16x3": "f16x3->f32 (fp32 operands split into fp16 hi/lo pairs, 3 MFMA products, fp32 accumulate)",
                      "f16": "f16->f32 (OPT-IN reduced precision: fp16 GEMM/attention operands, fp32 accumulate/softmax/norm/residual)",
                      "fp32": "f32"}[model.precision],
            "data": "synthetic",
            "config": {"workload": f"VoMix 32-NFE (16 midpoint steps, CFG 0.7) + HiFi-GAN config_covomix, "
                                   f"B={B} utterances x T={T} frames per GPU, prompt {PROMPT}",
                       "per_gpu_batch": B, "frames": T, "nfe": NFE, "parallelism": f"dp{world} (utterance-sharded)",
                       "utterances_per_step": B * batches_per_step * world},
            "path_flop_per_frame": FLOP_PER_FRAME,
            "path_frac_of_f32_mfma_peak": round(value / world * FLOP_PER_FRAME / PEAK_F32_MFMA, 4),
            # dominant kernel.  achieved = ALGORITHMIC flops (2*M*N*K per launch) / HIP-event time of the launches;
            # the split kernel executes 3 MFMA products per algorithmic product, so its matrix-pipe work is 3x that.
            "roofline": {"bound": "mfma",
                         "kernel": kname + ((" (%s x%d)" % (mfma, terms)) if split else " (v_mfma_f32_32x32x2_f32)"),
                         "achieved": round(achieved, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved / (peak / 1e12), 4), "traffic": traffic, "traffic_source": traffic_source,
                         "executed_mfma_frac": round(achieved * terms / (peak / 1e12), 4),
                         "vs_f32_mfma_peak": round(achieved / (PEAK_F32_MFMA / 1e12), 4),
                         "launches": all_launches, "timed_launches": launches,
                         "avg_launch_ms": round(gemm_s / launches * 1e3, 4),
                         "time_share_of_step": round(gemm_s / launches * all_launches / elapsed, 4),
                         "ms_per_step": round(gemm_s / launches * all_launches / args.steps * 1e3, 3)},
        }
        out["roofline"].update(power_info)          # power_w, sclk_mhz, power_cap_w of the timed region (rank 0's GPU)
        out["roofline"].update(effective_clock(clk0, clk1))
        # host budget: a rank drives ~10k launches/s from Python; `busy_cores` = CPU seconds of the process (all threads, the power
        # sampler included) per second of the timed region.  A rank whose share of the cgroup quota is below what it needs is
        # host-bound however fast the GPU is - the first thing to rule out when an N > 1 line comes in low.
        busy = [c / max(e, 1e-9) for c, e in zip(cpus, per_rank)]
        need = max(1.5, -(-busy[0] // 0.5) * 0.5)     # cores a rank needs: what rank 0 used here (launching thread + event brackets + sampler), rounded up
        share = dp.INFO.get("host_threads_per_rank") or host_cores()
        out["host"] = {"cpu_s_per_step": round(host_cpu / args.steps, 4), "busy_cores": [round(b, 3) for b in busy],
                       "cores_needed_per_rank": need, "host_cores_quota": host_cores(), "host_threads_per_rank": share,
                       "warning": (f"only {share} host thread(s) per rank for {need} needed: the ranks are host-bound" if share < need else None)}
        out["kernel_classes_ms_per_step"] = {k: c.result(args.steps) for k, c in classes.items()}
        voc_ms = sum(a.elapsed_time(b) for a, b in voc_events) / max(len(voc_events), 1)
        out["kernel_classes_ms_per_step"]["vocoder"] = {"launches_per_step": 1, "timed": len(voc_events), "avg_launch_ms": round(voc_ms, 4),
                                                        "ms_per_step": round(voc_ms, 3), "note": "whole Generator call incl. its saturation-flag read"}
        if world > 1:
            out["ranks"] = {"elapsed_s": [round(x, 4) for x in per_rank], "skew_max_over_min": round(max(per_rank) / max(min(per_rank), 1e-9), 4),
                            "load_and_broadcast_s": [round(x, 3) for x in loads]}
            out["ranks"].update({k: v for k, v in dp.INFO.items()})       # transports, RCCL version, IPC mode, a fallback's reason
        model_precision = model.precision
        if world == 1 and not args.no_fp32_exact and model.precision == "f16x3":
            del model, gen
            torch.cuda.empty_cache()
            with contextlib.redirect_stdout(sys.stderr):
                out["fp32_exact"] = fp32_exact(dev, rank, world)
        if world == 1 and not args.no_c2 and model_precision == "f16x3":
            with contextlib.redirect_stdout(sys.stderr):
                out["c2"] = config2(dev)
        if world == 1 and not args.no_c1 and model_precision == "f16x3":
            with contextlib.redirect_stdout(sys.stderr):
                out["c1"] = config1(dev)
        if world == 1 and not args.no_c5 and model_precision == "f16x3":
            torch.cuda.empty_cache()
            with contextlib.redirect_stdout(sys.stderr):
                out["c5"] = config5(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cpu_sd)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        torch.distributed.destroy_process_group()


def _guarded_main():
    """main(), and at N > 1 a line that says WHICH rank failed and how when one does (an RCCL abort takes the process down without a
    Python exception: the launcher's per-rank stderr tails cover that, covomix_amd/dp.launch_ranks; a Python-level failure is reported
    here) - rank 0 still prints ONE JSON line (value null, the error, what dp.INFO knows about the start-up) so that a failed scaling
    run is diagnosable from the record alone."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:          # noqa: BLE001
        import traceback
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        tb = traceback.format_exc()
        print(f"[bench.py rank {rank}/{world}] FAILED: {type(e).__name__}: {e}\n{tb[-2000:]}", file=sys.stderr, flush=True)
        if world > 1 and rank == 0:
            from covomix_amd import dp
            line = json.dumps({"metric": "mel-frames/sec (VoMix 32-step + HiFi-GAN, Bx1000x80)", "value": None, "unit": "mel-frames/s",
                              "n_gpus": world, "error": f"rank 0: {type(e).__name__}: {str(e)[:500]}",
                              "ranks": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in dp.INFO.items()}})
            os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())
        raise


if __name__ == "__main__":
    _guarded_main()
