#!/usr/bin/env python3
"""Dev: BASELINE config 5 in one process - CoMix text2semantic AR decode + VoMix 64-NFE + HiFi-GAN (covomix_amd/config5.py; recipe
weights, synthetic inputs, TOKENS decoded steps per dialogue with the eos ignored so the work is fixed, T = TOKENS + 400 prompt frames).
Three schedules over the same DIALOGUES dialogues (default 56; text2semantic always decodes 8 at a time), dialogues/s each:
  serial    the round-4 schedule: both stages on one plain stream, 8 dialogues per acoustic batch (whole GEMM rounds on 256 CUs);
  alternate both stages on the CU partition's streams (224 + 32 CUs), one after the other, 7 per acoustic batch (whole rounds on 224 CUs);
  pipelined the same calls with the decode of the next dialogues UNDER the solve of the current batch (covomix_amd/pipeline.py)
and the bit-identity of `alternate` and `pipelined` (tokens and PCM), tokens identical / PCM within 1 LSB against `serial`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd.config5 import Config5
from covomix_amd import ops

dev = torch.device("cuda:0")
TOK = int(os.environ.get("TOKENS", "608"))
ND = int(os.environ.get("DIALOGUES", "56"))
B_SERIAL, B_PIPE = int(os.environ.get("B_SERIAL", "8")), int(os.environ.get("B_PIPE", "7"))
c5 = Config5(dev, tokens=TOK)
part = ops.cu_partition(dev)
print(f"config 5, {ND} dialogues, T = {c5.T} frames ({TOK} decoded steps x 2 streams, 64 NFE); CU partition main {part.n_main} / side {part.n_side}", flush=True)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t0


def stage_times(B, s1=None, s2=None):
    """text2semantic of 8 dialogues / acoustic + vocoder of B, each alone on its stream (ms)."""
    cur = torch.cuda.current_stream()
    with torch.cuda.stream(s1 or cur):
        x, a = timed(lambda: c5.stage1([c5.dialogue(1000 + j) for j in range(8)]))
    with torch.cuda.stream(s2 or cur):
        _, b = timed(lambda: c5.stage2(x[:B]))
    return a * 1e3, b * 1e3


# warm-up of every shape on every stream (graph captures, workspaces, deferred-norm tables)
c5.run(8, B_SERIAL, overlap=False, partitioned=False)
c5.run(16, B_PIPE, overlap=False)
c5.run(16, B_PIPE, overlap=True)

t0 = time.perf_counter()
RECS = [c5.dialogue(j) for j in range(ND)]          # synthetic inputs: not part of the workload
print(f"(inputs of {ND} dialogues made on the host in {(time.perf_counter() - t0) * 1e3:.0f} ms, outside the timed runs)", flush=True)
_w = {"stage1": [], "stage2": []}
for _n in _w:
    def _wrap(fn, _n=_n):
        def f(x):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(x); torch.cuda.synchronize(); _w[_n].append(time.perf_counter() - t0)
            return r
        return f
    setattr(c5, _n, _wrap(getattr(c5, _n)))
ser, t = timed(lambda: c5.run(ND, B_SERIAL, overlap=False, partitioned=False, recs=RECS))
for _n in _w:
    setattr(c5, _n, getattr(Config5, _n).__get__(c5))
print("serial, per call [ms]: text2semantic " + "/".join(f"{x * 1e3:.0f}" for x in _w["stage1"]) + "; solve + vocoder " + "/".join(f"{x * 1e3:.0f}" for x in _w["stage2"]), flush=True)
a, b = stage_times(B_SERIAL)
print(f"serial    (one plain stream, {B_SERIAL} per solve): {ND / t:6.2f} dialogues/s = {ND * c5.T / t:7.0f} mel-frames/s  "
      f"[text2semantic of 8: {a:.0f} ms; solve + vocoder of {B_SERIAL}: {b:.0f} ms]", flush=True)
alt, t = timed(lambda: c5.run(ND, B_PIPE, overlap=False, recs=RECS))
a, b = stage_times(B_PIPE, part.side, part.main)
print(f"alternate (CU partition, {B_PIPE} per solve):       {ND / t:6.2f} dialogues/s = {ND * c5.T / t:7.0f} mel-frames/s  "
      f"[text2semantic of 8 on {part.n_side} CUs: {a:.0f} ms; solve + vocoder of {B_PIPE} on {part.n_main} CUs: {b:.0f} ms]", flush=True)
walls = {"stage1": []}
spans = {"stage1": [], "stage2": []}                  # (start, end) HIP events of every stage call ON ITS OWN STREAM: the device's view
base = torch.cuda.Event(enable_timing=True)
def wrap1(fn):                                        # wall time of every text2semantic call inside the pipelined run (host view)
    def f(x):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a_.record(); r = fn(x); b_.record(); torch.cuda.current_stream().synchronize()
        walls["stage1"].append(time.perf_counter() - t0); spans["stage1"].append((a_, b_))
        return r
    return f
def wrap2(fn):                                        # the solve is only ENQUEUED by its call (the host runs a batch ahead): events, no wait
    def f(x):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record(); r = fn(x); b_.record(); spans["stage2"].append((a_, b_))
        return r
    return f
c5.stage1, c5.stage2_launch = wrap1(c5.stage1), wrap2(c5.stage2_launch)
torch.cuda.synchronize(); base.record(); torch.cuda.synchronize()
pip, t = timed(lambda: c5.run(ND, B_PIPE, overlap=True, recs=RECS))
for name in ("stage1", "stage2_launch"):
    setattr(c5, name, getattr(Config5, name).__get__(c5))
iv = {k: [(base.elapsed_time(a_), base.elapsed_time(b_)) for a_, b_ in v] for k, v in spans.items()}
walls["stage2"] = [(e - s_) * 1e-3 for s_, e in iv["stage2"]]
ms = lambda v: "/".join(f"{x * 1e3:.0f}" for x in v)
print(f"pipelined (CU partition, {B_PIPE} per solve):       {ND / t:6.2f} dialogues/s = {ND * c5.T / t:7.0f} mel-frames/s  "
      f"[ms per call: text2semantic (wall) {ms(walls['stage1'])}; solve + vocoder (on its stream) {ms(walls['stage2'])}]", flush=True)
steady = sum(walls["stage2"][1:-1]) / max(len(walls["stage2"]) - 2, 1)
print(f"          steady state (a full solve + vocoder call per {B_PIPE} dialogues, the decode hidden): {B_PIPE / steady:.2f} dialogues/s; "
      f"this run pays one un-overlapped decode ({walls['stage1'][0] * 1e3:.0f} ms) to fill the pipeline")
# the two queues on the device's clock: when was a decode group running, when a solve batch, when both
both = sum(max(0.0, min(e1, e2) - max(s1, s2)) for s1, e1 in iv["stage1"] for s2, e2 in iv["stage2"])
dec, sol = sum(e - s_ for s_, e in iv["stage1"]), sum(e - s_ for s_, e in iv["stage2"])
end = max(e for v in iv.values() for _, e in v)
print(f"          device timeline (HIP events on each stage's own stream): decode stream busy {dec:.0f} ms, solve stream busy {sol:.0f} ms of {end:.0f} ms; "
      f"BOTH busy {both:.0f} ms = {100 * both / dec:.0f} % of the decode hidden under a solve")
print("          decode groups [ms]: " + " ".join(f"{s_:.0f}-{e:.0f}" for s_, e in iv["stage1"]))
print("          solve batches [ms]: " + " ".join(f"{s_:.0f}-{e:.0f}" for s_, e in iv["stage2"]))
same = all(torch.equal(x["streams"], y["streams"]) and torch.equal(x["pcm"], y["pcm"]) for x, y in zip(alt, pip))
print("tokens and PCM of the pipelined schedule bit-identical to the alternate one:", same)
tok_same = all(torch.equal(x["streams"], y["streams"]) for x, y in zip(ser, pip))
lsb = max(int((x["pcm"].int() - y["pcm"].int()).abs().max()) for x, y in zip(ser, pip))
print(f"against the serial single-stream schedule: tokens identical: {tok_same}; PCM max |difference| {lsb} LSB")

# round 6: the decode batch is no longer capped at 8 - the same schedules with 32 / 64 dialogues per decode pass
if os.environ.get("BIG", "1") == "1":
    for b1 in (32, 64):
        c5.run(min(ND, b1), B_SERIAL, overlap=False, partitioned=False, B1=b1)          # graphs / buffers of this decode batch
        big, t = timed(lambda: c5.run(ND, B_SERIAL, overlap=False, partitioned=False, recs=RECS, B1=b1))
        tok = all(torch.equal(x["streams"], y["streams"]) for x, y in zip(ser, big))
        pcm = all(torch.equal(x["pcm"], y["pcm"]) for x, y in zip(ser, big))
        print(f"serial, {b1} dialogues per decode pass (one plain stream, {B_SERIAL} per solve): {ND / t:6.2f} dialogues/s; tokens identical to the "
              f"8-per-pass serial schedule: {tok}; PCM identical: {pcm}", flush=True)
    for b1 in (16, 32):
        c5.run(min(ND, 2 * b1), B_PIPE, overlap=True, B1=b1)
        big, t = timed(lambda: c5.run(ND, B_PIPE, overlap=True, recs=RECS, B1=b1))
        same = all(torch.equal(x["streams"], y["streams"]) and torch.equal(x["pcm"], y["pcm"]) for x, y in zip(pip, big))
        print(f"pipelined, {b1} dialogues per decode pass (CU partition, {B_PIPE} per solve): {ND / t:6.2f} dialogues/s; bits equal to the 8-per-pass "
              f"pipelined schedule: {same}", flush=True)
