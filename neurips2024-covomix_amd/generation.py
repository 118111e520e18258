"""Driver behind monologue_generation.py / dialogue_generation.py at the repo root.

Keeps the reference CLI (monologue_generation.py:324-333: --t2s_ckpt --acous_ckpt --hifigan_ckpt --text_dir
--prompt_dir --saved_dir --seed --mode) and the per-utterance flow of covosingle()/covosinx()/covomix()
(monologue_generation.py:146-304, dialogue_generation.py:145-329) for the stages this build covers:
token/prompt assembly -> synthesis_sample (cond_scale 0.7) -> frame selection -> HiFi-GAN -> int16 wav.

Upstream stages:
  * text2semantic (N1, built: t2s.py): with --t2s_ckpt, `<text_dir>/<name>.text_ids.npy` (the BERT token ids of the
      cleaned text, what tokenizer(...).input_ids holds in cosingle_pred / comix_pred, :179-186, :307-319) or
      `<name>.txt` (needs the bert-base-uncased vocabulary in the local transformers cache - a third-party asset
      this repo does not ship) is decoded on the GPU; otherwise `<text_dir>/<name>.semantic.npy` holds already
      predicted tokens (covosingle / covosinx: int array [n]; covomix: [2, n] or flat [2n], split at half).
  * prompt mel extraction (N3, built: mel.py): `<prompt_dir>/<name>.mel.npy` ([80, T] log-mel) if present, else the
      log-mel of `<prompt_dir>/<name>.wav` (8 kHz) computed on the GPU
  * prompt tokens (N4, built: hubert.py): `<prompt_dir>/<name>.hubert_code.npy` if present (what
      fairseq-hubert/get_fisher_semantic_tokens.py wrote offline), else, with --hubert_ckpt and --km_path, the HuBERT
      layer-12 + k-means codes of `<prompt_dir>/<name>.wav` computed on the GPU (resampled to the checkpoint's rate);
      dialogue mode uses `<name>_1.*` and `<name>_2.*` (dialogue_generation.py:285-286)
Dialogue covosingle / covosinx decode the text turn by turn (split at "[spkchange]") with alternating prompts / streams
exactly as dialogue_generation.py:145-193, :205-272 do; pre-tokenised turns are `<name>.turn<k>.semantic.npy` /
`<name>.turn<k>.text_ids.npy`.
New relative to the reference: utterances are sharded over ranks (torchrun; the plan is built from file names and sizes
only, so every rank computes the same one and decodes only its own utterances), batched by equal length through the
acoustic model and by equal generated length through the vocoder; the sampled tokens and the ODE noise of an utterance
are seeded from (--seed, name, turn) and therefore do not depend on the number of ranks.
"""
from __future__ import annotations

import glob
import json
import os
import random
from argparse import ArgumentParser

import numpy as np
import torch

from . import assembly, dp
from .conditional_model import CoVoMixModel
from .vocoder import AttrDict, Generator

COND_SCALE = 0.7    # every shipped caller (monologue_generation.py:171,238,298)


def build_parser() -> ArgumentParser:
    p = ArgumentParser()
    p.add_argument("--t2s_ckpt", type=str, default=None, help="text2semantic checkpoint (CoSingle / CoMix); without it "
                   "<name>.semantic.npy files are read from --text_dir")
    p.add_argument("--acous_ckpt", type=str, default="/pretrained_models/comix.ckpt", help="acoustic model checkpoint")
    p.add_argument("--hifigan_ckpt", type=str, default="/pretrained_models/vocoder.ckpt", help="vocoder checkpoint")
    p.add_argument("--text_dir", type=str, default="test/test_dir", help="directory with <name>.semantic.npy")
    p.add_argument("--prompt_dir", type=str, default="test/monologue_prompt_dir", help="directory with acoustic prompts")
    p.add_argument("--saved_dir", type=str, default=".saved_dir", help="target directory")
    p.add_argument("--seed", type=int, default=30, help="random seed")
    p.add_argument("--mode", type=str, choices=["covosingle", "covosinx", "covomix"], default="covosingle")
    p.add_argument("--max_batch", type=int, default=32, help="utterances per launch at most (any lengths: they are packed back to back)")
    p.add_argument("--max_frames", type=int, default=None, help="frames per launch (sum over its utterances); default: chosen from the "
                   "directory's lengths among 8192 / 12288 / 16384 / 24576 by the cost of the packing in rounds of GEMM tiles "
                   "(dp.choose_max_frames); 14336 under --pipeline on (two rounds on the 224 CUs the acoustic stage owns)")
    p.add_argument("--pipeline", type=str, choices=["auto", "batch", "on", "serial", "off"], default="auto",
                   help="extension: how text2semantic (--t2s_ckpt) and the acoustic stage share the GPU.  batch (= auto when there is text "
                        "to decode): ALL turns first, through 64 continuously refilled decode slots on the whole chip, then one global packing "
                        "of the acoustic work - the fastest schedule measured (bench.py `c5`: the decode costs 4.7 ms per dialogue at 64 "
                        "slots against 89 ms of solve; hiding it on a CU partition costs the solve 12.5 %% of the chip).  on: decode the "
                        "NEXT utterances on a CU-masked side stream while the acoustic model and the vocoder work on the current batch "
                        "(covomix_amd/pipeline.py: first audio sooner); serial = the batches of `on` on the same two streams one after "
                        "the other (bit-identical output, for comparison); off = batch with the decode outside the timed region")
    p.add_argument("--hubert_ckpt", type=str, default=None, help="HuBERT checkpoint (fairseq layout): tokenise <name>.wav prompts "
                   "that have no <name>.hubert_code.npy (fairseq-hubert/get_fisher_semantic_tokens.py:23-24)")
    p.add_argument("--km_path", type=str, default=None, help="k-means model (joblib) for --hubert_ckpt")
    p.add_argument("--nfe", type=int, default=32, help="extension: CFG-combined field evaluations of the midpoint solver (32 = the "
                   "reference's ode_step_size 0.0625, acoustic.py:586-591; 64 = BASELINE config 5's 64-step setting)")
    p.add_argument("--gpus", type=int, default=1, help="extension: from a plain shell, start this many ranks (one per GPU, "
                   "utterances sharded; under torch.distributed.run the launcher's WORLD_SIZE is used instead)")
    return p


HEAD_START = 16     # utterances read before the first launch of a large directory (generation.run, --pipeline off)
_HUBERT = None      # HubertTokenizer, built by main() when --hubert_ckpt / --km_path are given


def _load_prompt(prompt_dir: str, name: str):
    code = os.path.join(prompt_dir, name + ".hubert_code.npy")
    if os.path.isfile(code) or _HUBERT is None:
        tok = torch.from_numpy(np.load(code).astype(np.int64))
    else:                                                  # encoder.wav2code(file, 1) (get_fisher_semantic_tokens.py:35-37, row N4)
        tok = torch.tensor([int(c) for c in _HUBERT.wav2code(os.path.join(prompt_dir, name + ".wav"), 1).split(" ")], dtype=torch.int64)
    npy = os.path.join(prompt_dir, name + ".mel.npy")
    if os.path.isfile(npy):
        mel = torch.from_numpy(np.load(npy).astype(np.float32))
    else:                                                  # extract_mel(prompt wav), monologue_generation.py:62-74 (row N3)
        from .mel import extract_mel
        mel = extract_mel(os.path.join(prompt_dir, name + ".wav"))
    return assembly.truncate_prompt(tok, mel)            # -> tokens [Tp], mel [Tp, 80]


def remove_punctuation(text: str) -> str:
    """monologue_generation.py:108-114."""
    punctuation = '''!()-{};:'"\\,<>./?@#$%^&*_~'''
    text = text.lower()
    for x in text:
        if x in punctuation:
            text = text.replace(x, "")
    return text


_TOKENIZER = None


def _stable_seed(seed: int, name: str, turn: int, salt: int) -> int:
    """Per-utterance RNG seed that depends only on (--seed, utterance name, turn): the sampled tokens and the ODE noise of an
    utterance are the same whatever the number of ranks, the sharding or the batching."""
    import zlib
    return (zlib.crc32(f"{name}|{turn}|{salt}".encode()) ^ (int(seed) * 0x9E3779B1)) & 0x7FFFFFFF


def _turn_sources(text_dir: str, name: str, multi_turn: bool, have_t2s: bool):
    """The text turns of one utterance as a list of ("sem", path) / ("ids", path) / ("txt", string) sources.
    Multi-turn (dialogue covosingle / covosinx: the reference splits the text at "[spkchange]" and decodes every turn on its
    own, dialogue_generation.py:160-165, :243-247): `<name>.turn<k>.semantic.npy` or `<name>.turn<k>.text_ids.npy`, k = 0, 1,
    ..., else `<name>.txt` split at "[spkchange]".  Otherwise (or when no turn files exist) the whole utterance is one turn:
    `<name>.semantic.npy`, `<name>.text_ids.npy` or `<name>.txt`."""
    j = lambda suffix: os.path.join(text_dir, name + suffix)
    if multi_turn:
        turns, k = [], 0
        while True:
            if os.path.isfile(j(f".turn{k}.semantic.npy")):
                turns.append(("sem", j(f".turn{k}.semantic.npy")))
            elif have_t2s and os.path.isfile(j(f".turn{k}.text_ids.npy")):
                turns.append(("ids", j(f".turn{k}.text_ids.npy")))
            else:
                break
            k += 1
        if turns:
            return turns
        if have_t2s and os.path.isfile(j(".txt")) and not os.path.isfile(j(".semantic.npy")) and not os.path.isfile(j(".text_ids.npy")):
            with open(j(".txt"), "r", encoding="utf-8") as f:
                return [("txt", t) for t in f.read().split("[spkchange]")]
    if os.path.isfile(j(".semantic.npy")):
        return [("sem", j(".semantic.npy"))]
    if have_t2s and os.path.isfile(j(".text_ids.npy")):
        return [("ids", j(".text_ids.npy"))]
    if have_t2s and os.path.isfile(j(".txt")):
        with open(j(".txt"), "r", encoding="utf-8") as f:
            return [("txt", f.read())]
    raise FileNotFoundError(f"{name}: no .semantic.npy" + (" / .text_ids.npy / .txt" if have_t2s else " (and no --t2s_ckpt)") + f" in {text_dir}")


def _source_cost(src) -> int:
    """Rank-independent size of a turn (bytes of its file or characters of its text): what the sharding plan is built on."""
    kind, v = src
    return len(v) if kind == "txt" else os.path.getsize(v)


def _tokenize(txt: str) -> torch.Tensor:
    global _TOKENIZER
    if _TOKENIZER is None:
        from transformers import BertTokenizer
        tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
        for t in ("[laughter]", "[spkchange]", "[spka]", "[spkb]", "[partialoverlap]", "[backchannel]"):
            tok.add_tokens([t])
        _TOKENIZER = tok
    return _TOKENIZER([remove_punctuation(txt).lower()], padding=True, truncation=True, return_tensors="pt").input_ids


def _predict_turns(work, t2s, device, seed: int, slots: int = 64) -> dict:
    """(name, turn) -> predicted semantic tokens (int64 numpy).  work: list of (name, turn, source).  Sources that are
    already tokens are read; the others are decoded by text2semantic on the GPU through `slots` decode slots, every turn
    with its OWN stream of uniforms (seeded from (--seed, name, turn)): the tokens do not depend on batching or ranks."""
    out, todo = {}, []
    for name, k, (kind, v) in work:
        if kind == "sem":
            out[(name, k)] = np.load(v).astype(np.int64)
        else:
            ids = torch.from_numpy(np.load(v).astype(np.int64)).reshape(1, -1) if kind == "ids" else _tokenize(v)
            todo.append((name, k, ids))
    if todo and t2s is None:
        raise RuntimeError("text sources need --t2s_ckpt")
    if todo:
        # ALL turns through t2s.generate_many: 64 continuously refilled decode slots - a turn that has sampled its eos frees its slot for
        # the next one on the device (the reference decodes turn by turn, dialogue_generation.py:297-304).  In windows of 256 turns: the
        # uniform draws of a window (8 MB per turn at 2048 steps) and its context k/v are resident while it decodes.
        from .t2s import WINDOW
        dec = t2s._get_t2s()
        S, V, L = dec.d["streams"], dec.d["vocab"], dec.max_length
        for w in range(0, len(todo), WINDOW):
            part = todo[w:w + WINDOW]
            uniforms = []
            for name, k, _ in part:
                g = torch.Generator(device=device).manual_seed(_stable_seed(seed, name, k, 1))
                uniforms.append(torch.rand(L, S, V, device=device, generator=g))
            toks = t2s.synthesis_sample_text2semantic([ids.to(device) for _, _, ids in part], uniforms=uniforms, slots=slots)
            for (name, k, _), t in zip(part, toks):
                out[(name, k)] = t.cpu().numpy().astype(np.int64)
    return out


def _build_items(mode: str, dialogue: bool, prompt_dir: str, name: str, preds):
    """One utterance -> list of (phoneme_ids, cond, mask) network inputs (one per OUTPUT SEGMENT; the segments' audio is
    concatenated).  preds: the predicted tokens of its turns, in order."""
    if mode == "covosingle":
        if not dialogue:                                                # monologue_generation.py:146-176
            sem, mel = _load_prompt(prompt_dir, name)
            return [assembly.build_monologue_inputs(sem, torch.from_numpy(preds[0].reshape(-1)), mel)]
        prompts = [_load_prompt(prompt_dir, name + "_1"), _load_prompt(prompt_dir, name + "_2")]
        return [assembly.build_monologue_inputs(prompts[k % 2][0], torch.from_numpy(p.reshape(-1)), prompts[k % 2][1])
                for k, p in enumerate(preds)]                           # dialogue_generation.py:145-193: turn k speaks with prompt k % 2
    if dialogue:
        sa, ma = _load_prompt(prompt_dir, name + "_1")
        sb, mb = _load_prompt(prompt_dir, name + "_2")
    else:
        sa, ma = _load_prompt(prompt_dir, name)
        sb, mb = sa, ma
    if mode == "covosinx":                # every turn on one stream, the other filled with 157; streams alternate per turn
        pa, pb = [], []                   # (dialogue_generation.py:243-259; one turn, stream A: monologue_generation.py:222-226)
        for k, p in enumerate(preds):
            t = torch.from_numpy(p.reshape(-1))
            sil = torch.ones_like(t) * assembly.SILENT_TOKEN
            pa.append(t if k % 2 == 0 else sil)
            pb.append(sil if k % 2 == 0 else t)
        pa, pb = torch.cat(pa), torch.cat(pb)
    else:                                 # covomix: one decode yields both streams, split at half (comix_pred, :316-320)
        flat = preds[0].reshape(-1)
        half = flat.shape[0] // 2
        pa, pb = torch.from_numpy(flat[:half].copy()), torch.from_numpy(flat[half:].copy())
    return [assembly.build_dialogue_inputs(sa, sb, pa, pb, ma, mb)]


def utterance_plan(text_dir: str, dialogue: bool, mode: str, have_t2s: bool, world: int):
    """(names, sources, plan): the sorted utterance names, their text turns, and plan[r] = the names rank r generates.
    RANK-INVARIANT by construction: built from file names and sizes only, before anything is sampled, so every rank
    computes the same plan, every utterance is generated exactly once, and a rank decodes only its own text."""
    stems = set()
    multi_turn = dialogue and mode in ("covosingle", "covosinx")
    for ext in (".semantic.npy",) + ((".text_ids.npy", ".txt") if have_t2s else ()):
        for p in glob.glob(os.path.join(text_dir, "*" + ext)):
            stem = os.path.basename(p)[: -len(ext)]
            if multi_turn and ".turn" in stem and stem.rsplit(".turn", 1)[1].isdigit():
                stem = stem.rsplit(".turn", 1)[0]
            stems.add(stem)
    names = sorted(stems)
    if not names:
        raise FileNotFoundError(f"no utterances in --text_dir {text_dir} (looked for *.semantic.npy"
                                + (", *.text_ids.npy, *.txt" if have_t2s else "; text sources need --t2s_ckpt") + ")")
    sources = {n: _turn_sources(text_dir, n, multi_turn, have_t2s) for n in names}
    cost = [sum(_source_cost(s) for s in sources[n]) for n in names]
    plan = [[names[i] for i in sorted(idx)] for idx in dp.shard_utterances(cost, world)]
    return names, sources, plan


def run(dialogue: bool, argv=None) -> int:
    import time
    from scipy.io.wavfile import write
    from . import ops
    import sys
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # spawn our own ranks (hifi-gan/train.py:268-278 does the same)
        rc = dp.launch_ranks(os.path.abspath(sys.argv[0]), list(sys.argv[1:] if argv is None else argv), args.gpus)
        if rc:
            raise SystemExit(rc)
        return 0
    print(args)
    os.makedirs(args.saved_dir, exist_ok=True)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    rank, world, local = dp.init_from_env()
    if not torch.cuda.is_available():
        from ._lib import CovomixHipError
        raise CovomixHipError("generation needs an MI355X: covomix_amd has no CPU path")
    torch.cuda.set_device(local)
    torch.cuda.manual_seed(args.seed)
    device = torch.device("cuda", local)

    config_file = os.path.join(os.path.split(args.hifigan_ckpt)[0], "vocoder_config.json")   # :368
    with open(config_file) as f:
        h = AttrDict(json.loads(f.read()))
    generator = Generator(h).to(device)
    assert os.path.isfile(args.hifigan_ckpt)
    state_dict_g = torch.load(args.hifigan_ckpt, map_location="cpu", weights_only=False)
    generator.load_state_dict(state_dict_g["generator"])
    generator.eval()
    generator.remove_weight_norm()
    generator.pack()                           # (part of loading the model, not of the first batch)
    model = CoVoMixModel.load_from_checkpoint(args.acous_ckpt, base_dir="", batch_size=16, num_workers=0)
    model.eval()
    model = model.to(device)
    model.nfe = int(args.nfe)
    if rank == 0:
        with open(os.path.join(args.saved_dir, "config.txt"), "w") as f:
            f.write("Vocoder: " + str(dict(h)) + "\n")
            f.write("t2s_ckpt: " + str(args.t2s_ckpt) + "\n")
            f.write("acoustic model: " + args.acous_ckpt + "\n")

    t2s = None
    if args.t2s_ckpt:
        assert os.path.isfile(args.t2s_ckpt), f"--t2s_ckpt {args.t2s_ckpt}: no such file (monologue_generation.py:46)"
        t2s = CoVoMixModel.load_from_checkpoint(args.t2s_ckpt, base_dir="", batch_size=16, num_workers=0)   # :93-96
        t2s.eval()
        t2s = t2s.to(device)
    global _HUBERT
    _HUBERT = None
    if args.hubert_ckpt or args.km_path:
        assert args.hubert_ckpt and args.km_path and os.path.isfile(args.hubert_ckpt) and os.path.isfile(args.km_path), \
            "--hubert_ckpt and --km_path must both name existing files"
        from .hubert import HubertTokenizer
        _HUBERT = HubertTokenizer(hubert_path=args.hubert_ckpt, hubert_layer=12, km_path=args.km_path)   # get_fisher_semantic_tokens.py:30-32

    names, sources, plan = utterance_plan(args.text_dir, dialogue, args.mode, t2s is not None, world)
    mine = plan[rank]
    work = [(n, k, s) for n in mine for k, s in enumerate(sources[n])]
    to_decode = sum(1 for _, _, (kind, _) in work if kind != "sem")
    mode = args.pipeline
    if mode == "auto":
        mode = "batch" if (t2s is not None and to_decode > 0) else "off"
    if mode != "off" and t2s is None:
        mode = "off"                                              # nothing to decode: tokens come from files
    two_stage = mode in ("on", "serial")
    from . import pipeline as pl
    # frames per launch: pipelined - two rounds of the N = 1024 products' tiles on the acoustic stage's CUs (the batches form while the
    # text is still being decoded: the lengths are not known up front); otherwise chosen from the directory's lengths below
    max_frames = args.max_frames or (2 * pl.frames_per_launch(device) if two_stage else None)
    segments = {n: {} for n in mine}
    n_out = model._get_field().d["dim_out"]       # acoustic.py:647-650: 80 channels (twocondition_oneoutput) or as wide as cond

    # The reference generates one utterance at a time (monologue_generation.py:259-304); here up to --max_batch utterances
    # of ANY lengths share a launch sequence: packed back to back (no padding), every utterance attending to itself only
    # (sample_ragged), so each gets the result of its own B = 1 run.  Batches are FILLED to --max_frames (first-fit decreasing,
    # dp.pack_by_frames): the frames of a launch decide how many whole rounds of GEMM tiles it runs.
    def one_batch(batch, y0, defer=False):
        """acoustic solve + vocoder of one packed batch of (inputs, (name, segment)) -> parts, generated frames.  parts: one
        (int16 PCM [items, samples] on the HOST, [(name, segment, samples of that item)]) per vocoder call; defer: the PCM travels by
        an asynchronous copy into pinned memory - valid once the stream has got there (_finish)."""
        its = [it for it, _ in batch]
        lens = [int(it[0].shape[0]) for it in its]
        up = lambda t: ops.h2d(t, device)          # (through pinned memory: a pageable copy would make the host wait for the stream)
        if len(set(lens)) == 1:                                                          # equal lengths: the plain [B, T, .] call
            sampled = list(model.synthesis_sample(phoneme_ids=up(torch.stack([it[0] for it in its])),
                                                  cond=up(torch.stack([it[1] for it in its])),
                                                  mask=torch.stack([it[2] for it in its]),
                                                  cond_scale=COND_SCALE, y0=torch.stack(y0)))
        else:
            sampled = model.synthesis_sample(phoneme_ids=[up(it[0]) for it in its], cond=[up(it[1]) for it in its],
                                             mask=[it[2] for it in its], cond_scale=COND_SCALE, y0=y0)
        # vocoder: the generated frames of every utterance of the batch (:299-300: mask is a suffix) go through HiFi-GAN in
        # ONE ragged call (zero-padded to the longest, per-item lengths: every item gets its B = 1 waveform); one int16 cast
        # and one device-to-host copy per batch, sliced per utterance on the host
        n_prompt = [int((~it[2]).sum()) for it in its]
        js = [j for j in range(len(its)) if lens[j] - n_prompt[j] > 0]
        # (items of similar length only: the ragged vocoder call pads to its longest item and skips no work behind a short
        #  one, dp.group_by_padding keeps that padding below 25 % of the real frames)
        tg_all = [lens[j] - n_prompt[j] for j in js]
        parts, nfr = [], 0
        for grp in dp.group_by_padding(tg_all):
            gj = [js[k] for k in grp]
            tgen = [tg_all[k] for k in grp]
            mel = torch.zeros(len(gj), n_out, max(tgen), dtype=torch.float32, device=device)
            for r, j in enumerate(gj):
                mel[r, :, : tgen[r]] = sampled[j][n_prompt[j]:, :].T
            wav = generator(mel, lengths=tgen) if len(set(tgen)) > 1 else generator(mel)
            pcm_dev = ops.wav_to_int16(wav.squeeze(1).contiguous())                        # mel_decode_to_wav (:52-59), batched
            if defer:
                pcm = torch.empty(pcm_dev.shape, dtype=pcm_dev.dtype, pin_memory=True)
                pcm.copy_(pcm_dev, non_blocking=True)
            else:
                pcm = pcm_dev.cpu()
            nfr += sum(tgen)
            parts.append((pcm, [(batch[j][1][0], batch[j][1][1], generator.output_length(tgen[r])) for r, j in enumerate(gj)]))
        return parts, nfr

    def _store(parts):
        for pcm, owners in parts:
            pcm = pcm.numpy()
            for r, (n, seg, ns) in enumerate(owners):
                segments[n][seg] = pcm[r, :ns].copy()

    # The host runs ONE BATCH AHEAD of the device: solve() enqueues a batch - inputs through pinned memory, the solve, the vocoder, the
    # PCM and the saturation flag into pinned memory, an event - and only then waits for the batch BEFORE it.  File-to-tensor work,
    # packing, the per-utterance copies around the vocoder and the Python between the launches (7 % of the device's time on a directory
    # of 16 utterances when every batch ended in a blocking copy) then happen under the previous batch's kernels.
    pending: list = []

    def _finish(entry):
        batch, y0, parts, flag, ev, nfr, st = entry
        ev.synchronize()
        if flag is not None and int(flag[0]) != 0:
            # a flagged batch is repeated with the per-call checks (the stage that saturated warns and re-runs in fp32, or raises
            # under CVX_ON_SATURATION=raise) ON THE STREAM IT WAS ENQUEUED ON: the last batch of a two-stage run is collected after
            # the runner has left its main stream - another stream owns other CUs, picks other GEMM kernels and does not own y0
            with torch.cuda.stream(st):
                parts, nfr = one_batch(batch, y0)
        _store(parts)
        now = time.perf_counter()
        batch_log.append((len(batch), sum(int(it[0].shape[0]) for it, _ in batch), nfr, now - batch_log_t[0]))
        batch_log_t[0] = now

    def solve(batch):
        """one packed batch on the current stream: noise, solve, vocoder, saturation flag -> generated frames (its PCM is collected when
        the NEXT batch has been enqueued, or by drain())"""
        y0 = [torch.randn(int(it[0].shape[0]), n_out, device=device, generator=torch.Generator(device=device).manual_seed(
            _stable_seed(args.seed, own[0], own[1], 2))) for it, own in batch]             # acoustic.py:647-650, per utterance
        with ops.saturation_deferred(read=False):          # one flag read per batch instead of one blocking read per call
            parts, nfr = one_batch(batch, y0, defer=True)
        flag = ops.saturation_snapshot()
        ev = torch.cuda.Event()
        ev.record()
        pending.append((batch, y0, parts, flag, ev, nfr, torch.cuda.current_stream()))
        while len(pending) > 1:
            _finish(pending.pop(0))
        return nfr

    def drain():
        while pending:
            _finish(pending.pop(0))

    def items_of(n, pred):
        return [(it, (n, seg)) for seg, it in enumerate(_build_items(args.mode, dialogue, args.prompt_dir, n,
                                                                     [pred[(n, k)] for k in range(len(sources[n]))]))]

    done, frames = 0, 0
    batch_log_t = [0.0]
    batch_log: list = []                     # (utterances, frames in the launch, generated frames, seconds) per batch -> last_stats
    load_s = 0.0
    if mode == "off":       # ---- text2semantic for this rank's utterances first (not timed, as in round 4), then ONE global packing
        pred = _predict_turns(work, t2s, device, args.seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch_log_t[0] = t0
    if mode == "batch":     # ---- every turn of this rank through the continuously batched decode (timed), then ONE global packing
        pred = _predict_turns(work, t2s, device, args.seed)
    head_start = not two_stage and len(mine) >= 2 * HEAD_START
    if head_start:
        # ---- a large directory: the model inputs of the first HEAD_START utterances are read, their fullest first-fit-decreasing bin
        # is ENQUEUED, and the rest of the directory is read while the device works on it (reading is Python + numpy, ~0.7 ms per
        # utterance: 2-3 % of the run when it all happens before the first launch; a reader THREAD was tried first and fought the
        # enqueueing thread for the interpreter lock - 40 utterances in 224-295 ms instead of 40, the first launch enqueued in
        # 200-390 ms instead of 60); then one global packing of everything that is left.  Every batch is a function of the
        # utterance list alone.
        if max_frames is None:
            max_frames = max(256, 32 * ops.stream_cus())          # two rounds of the N = 1024 products' tiles (8192 on 256 CUs)
        head = [x for n in mine[:HEAD_START] for x in items_of(n, pred)]
        load_s = time.perf_counter() - t0
        lengths = [int(it[0].shape[0]) for it, _ in head]
        bins = dp.pack_by_frames(list(range(len(head))), lengths, max_frames, args.max_batch)
        first = max(range(len(bins)), key=lambda b: (sum(lengths[i] for i in bins[b]), -b))
        frames += solve([head[i] for i in bins[first]])
        taken = set(bins[first])
        pool = [x for i, x in enumerate(head) if i not in taken] + [x for n in mine[HEAD_START:] for x in items_of(n, pred)]
        lengths = [int(it[0].shape[0]) for it, _ in pool]
        for b in dp.pack_by_frames(list(range(len(pool))), lengths, max_frames, args.max_batch):
            frames += solve([pool[i] for i in b])
    elif not two_stage:
        pool = [x for n in mine for x in items_of(n, pred)]
        load_s = time.perf_counter() - t0                         # prompt files -> model inputs (host only)
        lengths = [int(it[0].shape[0]) for it, _ in pool]
        if max_frames is None:               # the cap whose packing costs least in rounds of GEMM tiles (dp.choose_max_frames)
            max_frames = dp.choose_max_frames(lengths, args.max_batch, ops.stream_cus())
        for b in dp.pack_by_frames(list(range(len(pool))), lengths, max_frames, args.max_batch):
            frames += solve([pool[i] for i in b])
    else:
        # ---- two-stage pipeline: stage 1 decodes the turns 8 at a time (side stream, worker thread), the collate step assembles
        # the utterances whose turns are all there and packs them; stage 2 = solve() on the main stream.  A bin is launched when it
        # is (nearly) full; the rest waits for more utterances - every batch is a function of the utterance list alone, so
        # `--pipeline serial` runs exactly the same batches (bit-identical PCM).
        groups = [work[i:i + 8] for i in range(0, len(work), 8)]

        def collate(results):
            pred, pool, pending = {}, [], list(mine)
            for part_pred in results:
                pred.update(part_pred)
                while pending and all((pending[0], k) in pred for k in range(len(sources[pending[0]]))):
                    pool.extend(items_of(pending.pop(0), pred))
                last = not pending
                lengths = [int(it[0].shape[0]) for it, _ in pool]
                bins = dp.pack_by_frames(list(range(len(pool))), lengths, max_frames, args.max_batch)
                keep = []
                for b in bins:
                    full = sum(lengths[i] for i in b) >= 0.9 * max_frames or len(b) >= args.max_batch
                    if full or last:
                        yield [pool[i] for i in b]
                    else:
                        keep += b
                pool = [pool[i] for i in sorted(keep)]
        nfr = pl.run_two_stage(groups, lambda g: _predict_turns(g, t2s, device, args.seed), solve, device,
                               overlap=(mode == "on"), collate=collate)
        frames = sum(nfr)
    drain()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for n in mine:
        if not segments[n]:
            continue
        audio = np.concatenate([segments[n][k] for k in sorted(segments[n])])           # dialogue covosingle: turns in order (:190)
        out = os.path.join(args.saved_dir, n + ".wav")
        write(out, 8000, audio)
        print("Saved wavfile", out)
        done += 1
    print(f"rank {rank}: {done} utterances, {frames} generated frames in {elapsed:.3f} s ({frames / max(elapsed, 1e-9):.1f} frames/s, "
          f"{'text2semantic + ' if mode != 'off' else ''}sampling + vocoder, excluding model load; --pipeline {mode})")
    run.last_stats = dict(utterances=done, frames=frames, seconds=elapsed, pipeline=mode, load_seconds=load_s, batches=batch_log,
                          max_frames=max_frames, head_start=bool(head_start))
    return done
