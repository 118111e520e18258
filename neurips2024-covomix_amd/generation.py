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
New relative to the reference: utterances are sharded over ranks (torchrun) and batched by equal length.
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
from .vocoder import AttrDict, Generator, mel_decode_to_wav

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
    p.add_argument("--max_batch", type=int, default=8, help="equal-length utterances per launch")
    p.add_argument("--hubert_ckpt", type=str, default=None, help="HuBERT checkpoint (fairseq layout): tokenise <name>.wav prompts "
                   "that have no <name>.hubert_code.npy (fairseq-hubert/get_fisher_semantic_tokens.py:23-24)")
    p.add_argument("--km_path", type=str, default=None, help="k-means model (joblib) for --hubert_ckpt")
    return p


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


def _text_ids(text_dir: str, name: str) -> torch.Tensor:
    """[1, n] BERT ids of the utterance text (load_text2semantic_model, monologue_generation.py:92-104)."""
    npy = os.path.join(text_dir, name + ".text_ids.npy")
    if os.path.isfile(npy):
        return torch.from_numpy(np.load(npy).astype(np.int64)).reshape(1, -1)
    global _TOKENIZER
    if _TOKENIZER is None:
        from transformers import BertTokenizer
        tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
        for t in ("[laughter]", "[spkchange]", "[spka]", "[spkb]", "[partialoverlap]", "[backchannel]"):
            tok.add_tokens([t])
        _TOKENIZER = tok
    with open(os.path.join(text_dir, name + ".txt"), "r", encoding="utf-8") as f:
        txt = remove_punctuation(f.read()).lower()
    return _TOKENIZER([txt], padding=True, truncation=True, return_tensors="pt").input_ids


def _predicted_tokens(text_dir: str, names, t2s, device, batch: int = 8) -> dict:
    """name -> predicted semantic tokens: read from <name>.semantic.npy when present, otherwise text2semantic on the GPU,
    `batch` utterances per decode batch (the tokens do not depend on the batching)."""
    out, todo = {}, []
    for n in names:
        sem = os.path.join(text_dir, n + ".semantic.npy")
        if t2s is None or os.path.isfile(sem):
            out[n] = np.load(sem).astype(np.int64)
        else:
            todo.append(n)
    for i in range(0, len(todo), batch):
        group = todo[i:i + batch]
        toks = t2s.synthesis_sample_text2semantic([_text_ids(text_dir, n).to(device) for n in group])
        for n, t in zip(group, toks):
            out[n] = t.cpu().numpy().astype(np.int64)
    return out


def _utterance_inputs(mode: str, dialogue: bool, text_dir: str, prompt_dir: str, name: str, pred: np.ndarray):
    if mode == "covosingle":
        sem, mel = _load_prompt(prompt_dir, name)
        return assembly.build_monologue_inputs(sem, torch.from_numpy(pred.reshape(-1)), mel)
    if dialogue:
        sa, ma = _load_prompt(prompt_dir, name + "_1")
        sb, mb = _load_prompt(prompt_dir, name + "_2")
    else:
        sa, ma = _load_prompt(prompt_dir, name)
        sb, mb = sa, ma
    if mode == "covosinx":                                # second stream silent (:223-224)
        pa = torch.from_numpy(pred.reshape(-1))
        pb = torch.ones_like(pa) * assembly.SILENT_TOKEN
    else:
        flat = pred.reshape(-1)
        half = flat.shape[0] // 2
        pa, pb = torch.from_numpy(flat[:half].copy()), torch.from_numpy(flat[half:].copy())
    return assembly.build_dialogue_inputs(sa, sb, pa, pb, ma, mb)


def run(dialogue: bool, argv=None) -> int:
    args = build_parser().parse_args(argv)
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
    torch.cuda.manual_seed(args.seed + rank)
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
    model = CoVoMixModel.load_from_checkpoint(args.acous_ckpt, base_dir="", batch_size=16, num_workers=0)
    model.eval()
    model = model.to(device)
    if rank == 0:
        with open(os.path.join(args.saved_dir, "config.txt"), "w") as f:
            f.write("Vocoder: " + str(dict(h)) + "\n")
            f.write("t2s_ckpt: " + str(args.t2s_ckpt) + "\n")
            f.write("acoustic model: " + args.acous_ckpt + "\n")

    t2s = None
    if args.t2s_ckpt and os.path.isfile(args.t2s_ckpt):
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
    stems = set()
    for ext in (".semantic.npy",) + ((".text_ids.npy", ".txt") if t2s is not None else ()):
        stems |= {os.path.basename(p)[: -len(ext)] for p in glob.glob(os.path.join(args.text_dir, "*" + ext))}
    names = sorted(stems)
    pred = _predicted_tokens(args.text_dir, names, t2s, device)
    items = [_utterance_inputs(args.mode, dialogue, args.text_dir, args.prompt_dir, n, pred[n]) for n in names]
    lengths = [int(it[0].shape[0]) for it in items]
    mine = dp.shard_utterances(lengths, world)[rank]
    done = 0
    for batch in dp.batch_equal_length(mine, lengths, args.max_batch):
        ids = torch.stack([items[i][0] for i in batch]).to(device)
        cond = torch.stack([items[i][1] for i in batch]).to(device)
        mask = torch.stack([items[i][2] for i in batch]).to(device)
        sampled = model.synthesis_sample(phoneme_ids=ids, cond=cond, mask=mask, cond_scale=COND_SCALE)
        for j, i in enumerate(batch):
            valid = assembly.select_generated_frames(sampled[j:j + 1], mask[j])
            if valid.shape[1] == 0:
                continue
            audio = mel_decode_to_wav(generator, valid.contiguous())
            from scipy.io.wavfile import write
            out = os.path.join(args.saved_dir, names[i] + ".wav")
            write(out, 8000, audio)
            print("Saved wavfile", out)
            done += 1
    print(f"rank {rank}: {done} utterances")
    return done
