"""Utterance-sharded data parallelism: one process per GPU, no data-path
collective.  The reference has no inference-time multi-GPU path (single process, B=1,
monologue_generation.py:259-304); each utterance is an independent ODE solve + vocoder call, so
ranks only need identical weights: ONE broadcast from rank 0 over RCCL/xGMI at start-up
(`backend="nccl"` is RCCL on ROCm), then nothing until the optional metric reduction.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun env; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deal utterance indices to ranks: longest first, each to the currently lightest rank
    (greedy by total frames); ties broken by rank so every rank computes the same plan."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(i)
        load[r] += int(lengths[i])
    return plan


def batch_equal_length(indices: Sequence[int], lengths: Sequence[int], max_batch: int) -> List[List[int]]:
    """Group a rank's utterances into batches of identical T (the network has no key-padding mask,
    acoustic.py:313, so only equal-length batching preserves B=1 results)."""
    by_len: Dict[int, List[int]] = {}
    for i in indices:
        by_len.setdefault(int(lengths[i]), []).append(i)
    out = []
    for T in sorted(by_len, reverse=True):
        g = by_len[T]
        out += [g[k:k + max_batch] for k in range(0, len(g), max_batch)]
    return out


def broadcast_state_dict(sd: Dict[str, torch.Tensor], device: torch.device, src: int = 0,
                         bucket_bytes: int = 256 << 20) -> Dict[str, torch.Tensor]:
    """Make every rank hold rank `src`'s tensors.  All ranks must pass dicts with identical
    keys/shapes/dtypes (non-src contents are overwritten).  Tensors are packed into flat fp32
    buckets (default 256 MB) so a 1 GB VoMix checkpoint is a handful of large broadcasts - the
    per-link-bound regime xGMI rings want - instead of 131 small ones."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: v.to(device) for k, v in sd.items()}
    keys = list(sd.keys())
    out: Dict[str, torch.Tensor] = {}
    bucket: List[str] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([sd[k].to(device=device, dtype=torch.float32).reshape(-1) for k in bucket])
        dist.broadcast(flat, src=src)
        off = 0
        for k in bucket:
            n = sd[k].numel()
            out[k] = flat[off:off + n].reshape(sd[k].shape).to(sd[k].dtype)
            off += n
        bucket, size = [], 0

    for k in keys:
        nbytes = sd[k].numel() * 4
        if size and size + nbytes > bucket_bytes:
            flush()
        bucket.append(k)
        size += nbytes
    flush()
    return out


def reduce_metric(frames: float, seconds: float, device: torch.device) -> tuple[float, float]:
    """(sum of frames over ranks, max of elapsed over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return frames, seconds
    t = torch.tensor([frames], dtype=torch.float64, device=device)
    m = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(t.item()), float(m.item())
