"""Integer / index path around the acoustic model: prompt + predicted token assembly and the
selection of generated frames.  Bit-exact mirror of the reference script logic (host-side,
CPU tensors, exactly where the reference runs it):
  covomix()    monologue_generation.py:263-295, dialogue_generation.py:287-320
  covosingle() monologue_generation.py:161-166
  output selection monologue_generation.py:299-300
"""
from __future__ import annotations

import torch

SILENT_TOKEN = 157     # filler for the shorter stream (monologue_generation.py:287-288)
MAX_TOKEN = 501        # clamp applied after concatenation (:290)
MAX_PROMPT_FRAMES = 400  # prepare_oracle_hubert truncation, 8 s (:87-89)


def build_dialogue_inputs(sem_a, sem_b, pred_a, pred_b, mel_a, mel_b):
    """Two-stream VoMix inputs.  sem_*: prompt tokens [Tp*]; pred_*: predicted tokens; mel_*: [Tp*,80].
    Returns (phoneme_ids i64[T,2], cond f32[T,160], mask bool[T])."""
    n_prompt = min(int(mel_a.shape[0]), int(mel_b.shape[0]))
    streams = []
    for sem, pred in ((sem_a, pred_a), (sem_b, pred_b)):
        streams.append(torch.cat((sem[:n_prompt].long(), pred.long())))
    total = max(int(s.shape[0]) for s in streams)
    ids = torch.full((total, 2), SILENT_TOKEN, dtype=torch.long)
    for j, s in enumerate(streams):
        ids[: s.shape[0], j] = s
    ids.clamp_(max=MAX_TOKEN)
    cond = torch.zeros(total, 160, dtype=torch.float32)
    cond[:n_prompt, :80] = mel_a[:n_prompt]
    cond[:n_prompt, 80:] = mel_b[:n_prompt]
    mask = torch.arange(total) >= n_prompt
    return ids, cond, mask


def build_monologue_inputs(sem, pred, mel_prompt):
    """Single-stream VoSingle inputs: (phoneme_ids i64[T], cond f32[T,80], mask bool[T])."""
    ids = torch.cat((sem.long(), pred.long())).clamp(max=MAX_TOKEN)
    total, n_prompt = int(ids.shape[0]), int(mel_prompt.shape[0])
    cond = torch.zeros(total, 80, dtype=torch.float32)
    cond[:n_prompt] = mel_prompt
    mask = torch.arange(total) >= n_prompt
    return ids, cond, mask


def truncate_prompt(tokens, mel):
    """equal_len + 8 s cap of prepare_oracle_hubert (monologue_generation.py:76-90); mel is [80,T]."""
    n = min(int(tokens.shape[0]), int(mel.shape[1]), MAX_PROMPT_FRAMES)
    return tokens[:n], mel[:, :n].permute(1, 0)


def select_generated_frames(sampled, mask):
    """sampled [1,T,80], mask bool[T] -> [80,Tgen] for the vocoder (monologue_generation.py:299-300)."""
    return sampled[:, mask.to(sampled.device), :].permute(0, 2, 1).squeeze(0)
