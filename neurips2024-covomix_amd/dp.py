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


def free_port() -> int:
    """A TCP port that is free on 127.0.0.1 right now (chosen by the kernel)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def single_device_test_mode() -> bool:
    """TEST-ONLY switch (env CVX_DP_SINGLE_DEVICE=1): every rank uses cuda:0 and the process group runs on gloo, so the
    N > 1 code path (sharding, weight broadcast, metric reduction, one output per utterance) can be exercised on a box with a
    single GPU.  RCCL cannot put two ranks on one device; production launches never set this."""
    return os.environ.get("CVX_DP_SINGLE_DEVICE", "0") == "1"


def launch_ranks(script: str, argv: Sequence[str], n: int) -> int:
    """Run `script argv...` as n ranks of ONE node under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 at a free port) and return its exit code.  What `bench.py --gpus N` / the generation scripts do when they are
    started from a plain shell - the reference's multi-GPU entry point spawns its own ranks too (hifi-gan/train.py:268-278)."""
    import glob
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory(prefix="covomix_ranks_") as logs:
        # --tee 2: every rank's stderr still streams through (stdout is untouched: rank 0's JSON line stays ONE bare line) and a copy
        # per rank lands under `logs`, so that a rank that dies (an RCCL abort is not a Python exception: nothing else would say which
        # rank and why) can be reported below
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
        rc = 1
        for attempt in range(2):
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                   "--master-port", str(free_port()), "--tee", "2", "--log-dir", os.path.join(logs, f"try{attempt}"), script, *argv]
            # (the launcher's own stderr is kept apart from the ranks': a rendezvous port that someone took between free_port() and
            #  the store's bind is the one failure worth a second attempt)
            proc = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")
            seen = []
            for line in proc.stderr:                                # forwarded as it comes, remembered for the check below
                sys.stderr.write(line)
                seen.append(line)
                del seen[:-200]
            rc = proc.wait()
            if rc == 0 or not any(k in ln for ln in seen for k in ("Address already in use", "EADDRINUSE", "address already in use")):
                break
            print("[covomix_amd.dp] the rendezvous port was taken: trying once more on another port", file=sys.stderr)
        if rc != 0:
            print(f"[covomix_amd.dp] {n} ranks of {os.path.basename(script)} ended with exit code {rc}; stderr tail per rank:", file=sys.stderr)
            for path in sorted(glob.glob(os.path.join(logs, "**", "stderr.log"), recursive=True)):
                try:
                    tail = open(path, errors="replace").read().strip().splitlines()[-12:]
                except OSError:
                    continue
                rank = os.path.basename(os.path.dirname(path))
                print(f"--- rank {rank} ---\n" + "\n".join(tail), file=sys.stderr)
        return rc


# what happened at start-up, for the `ranks` block of the bench line / the CLI log: which transport carried the weight broadcast,
# library versions and the environment RCCL depends on, and - if RCCL could not be used - why
INFO: Dict[str, object] = {}


def pin_host_threads(local: int, local_world: int) -> int:
    """Give this rank its share of the host: torch intra-op threads = cores / ranks, and (when the affinity mask is wide enough)
    a disjoint slice of the allowed CPUs - eight ranks each spawning a full-width OpenMP team oversubscribe the cgroup quota."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = list(range(os.cpu_count() or 1))
    n = len(cpus)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    share = max(1, n // max(1, local_world))
    torch.set_num_threads(share)
    if local_world > 1 and len(cpus) >= 2 * local_world:
        per = len(cpus) // local_world
        slot = int(os.environ.get("LOCAL_RANK", local)) % local_world      # (the REAL local rank: the one-device test modes pass local = 0 for every rank)
        try:
            os.sched_setaffinity(0, cpus[slot * per:(slot + 1) * per])
        except Exception:
            pass
    INFO["host_threads_per_rank"] = share
    return share


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1.
    MASTER_ADDR defaults to 127.0.0.1; MASTER_PORT must come from the launcher (torch.distributed.run sets it; launch_ranks
    picks a free one) - ranks cannot agree on a port by themselves."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if single_device_test_mode():
        backend, local = "gloo", 0
    elif os.environ.get("CVX_DP_RCCL_ON_ONE_DEVICE", "0") == "1":
        local = 0           # TEST ONLY: every rank on cuda:0 but RCCL still requested - it must refuse (duplicate GPU) and the start-up
                            # path must fall back to gloo with the reason in INFO (tests/test_multirank_gpu.py)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 but no MASTER_PORT: start the ranks with `python -m torch.distributed.run "
                               "--master-addr 127.0.0.1 --master-port <free port> ...` or covomix_amd.dp.launch_ranks")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL says why when it fails
        # The DEFAULT group runs on gloo (TCP on 127.0.0.1: rendezvous, the scalar metric reduction, diagnostics) - it works
        # wherever torch.distributed does.  The one collective that carries data, the start-up weight broadcast, goes over an
        # RCCL group on top of it (broadcast_state_dict); if RCCL cannot come up on a node (e.g. hipIpcGetMemHandle under the
        # legacy IPC mode) the broadcast falls back to the gloo group and says so - a rank never dies at start-up for it.
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        INFO.update(default_backend="gloo", requested_backend=backend,
                    HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), NCCL_DEBUG=os.environ.get("NCCL_DEBUG"))
        if backend == "nccl":
            try:
                INFO["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:                            # noqa: BLE001
                INFO["rccl_version"] = f"unavailable ({type(e).__name__})"
        pin_host_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    return rank, world, local


_RCCL = {"group": None, "tried": False, "error": None}


def rccl_group():
    """The RCCL (backend 'nccl') group for device-to-device collectives, created and PROVEN on first use (a tiny all-reduce,
    synchronised); None when the requested backend is not nccl, in the single-device test mode, or when RCCL failed - the reason
    is kept in INFO['rccl_error'] and every rank agrees on the outcome (the verdict is all-reduced over the gloo group)."""
    if _RCCL["tried"]:
        return _RCCL["group"]
    _RCCL["tried"] = True
    if INFO.get("requested_backend") != "nccl" or single_device_test_mode() or not torch.cuda.is_available():
        INFO["broadcast_backend"] = "gloo (" + ("single-device test mode" if single_device_test_mode() else "no RCCL requested") + \
                                    ": weights staged through host memory)"
        return None
    import datetime

    def agreed(ok: float) -> bool:
        flag = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)           # (default group = gloo)
        return float(flag.item()) >= 1.0
    ok, err, g = 1.0, None, None
    timeout = datetime.timedelta(seconds=float(os.environ.get("CVX_DP_RCCL_TIMEOUT_S", "180")))
    # A communicator that cannot come up on ONE rank leaves the others inside the probe: the default handling of that is the
    # process-group watchdog aborting the process (not an exception).  For the probe only: blocking waits that raise on time-out and
    # no watchdog abort - both are read when the group is constructed.  UNTESTED with more than one GPU (no such node was available
    # in rounds 1-5): README says so.
    saved = {k: os.environ.get(k) for k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_BLOCKING_WAIT")}
    os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"], os.environ["TORCH_NCCL_BLOCKING_WAIT"] = "0", "1"
    try:
        try:
            g = dist.new_group(backend="nccl", timeout=timeout)
        except Exception as e:                                # noqa: BLE001
            ok, err = 0.0, f"{type(e).__name__}: {str(e)[:300]}"
        if agreed(ok):                                        # every rank has a group object: only now does anyone enter an RCCL collective
            try:
                probe = torch.ones(1, device="cuda")
                work = dist.all_reduce(probe, group=g, async_op=True)
                work.wait(timeout)
                torch.cuda.synchronize()
                if float(probe.item()) != float(dist.get_world_size()):
                    raise RuntimeError(f"RCCL all-reduce probe returned {float(probe.item())}")
            except Exception as e:                            # noqa: BLE001
                ok, err = 0.0, f"{type(e).__name__}: {str(e)[:300]}"
        else:
            ok = 0.0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if not agreed(ok):
        INFO["rccl_error"] = err or "another rank failed to bring RCCL up"
        g = None
    INFO["broadcast_backend"] = "nccl (RCCL)" if g is not None else "gloo (RCCL unavailable: weights staged through host memory)"
    _RCCL["group"] = g
    return g


def _on_gloo() -> bool:
    """True when device tensors must be staged through the host for a collective (no usable RCCL group)."""
    return dist.is_initialized() and rccl_group() is None


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


def batch_by_frames(indices: Sequence[int], lengths: Sequence[int], max_batch: int, max_frames: int) -> List[List[int]]:
    """Group a rank's utterances into ragged batches: in the given order, up to max_batch utterances and max_frames frames
    per batch (an utterance longer than max_frames runs alone).  Packed batches have no padding, so the grouping does not
    change the work - only how much of it one launch sequence carries."""
    out: List[List[int]] = []
    cur: List[int] = []
    frames = 0
    for i in indices:
        t = int(lengths[i])
        if cur and (len(cur) >= max_batch or frames + t > max_frames):
            out.append(cur)
            cur, frames = [], 0
        cur.append(i)
        frames += t
    if cur:
        out.append(cur)
    return out


def pack_by_frames(indices: Sequence[int], lengths: Sequence[int], max_frames: int, max_batch: int) -> List[List[int]]:
    """First-fit-decreasing bin packing of a rank's utterances into ragged batches of at most max_frames frames (and max_batch
    utterances); an utterance longer than max_frames runs alone.  Why fill the bins: packed batches have no padding, but the
    large-problem GEMM works in 256-row panels on 256 CUs - with both CFG branches 8192 frames are 64 row panels = whole rounds of
    tiles for every projection of the block (to_out / ff2 256 tiles, to_qkv 768, ff1 1024), while e.g. 6400 frames are 50
    panels = 3.1 rounds of ff1 tiles, paid as 4.  Deterministic (ties by index): every rank computes the same plan."""
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i))
    bins: List[List[int]] = []
    room: List[int] = []
    for i in order:
        t = int(lengths[i])
        for b in range(len(bins)):
            if room[b] >= t and len(bins[b]) < max_batch:
                bins[b].append(i)
                room[b] -= t
                break
        else:
            bins.append([i])
            room.append(max(0, max_frames - t))
    return bins


def launch_cost(frames: int, cus: int = 256, dim: int = 1024) -> float:
    """Relative time of the transformer GEMMs of ONE packed launch of `frames` frames (both CFG branches: 2 x frames rows) in units
    of one round of 256 x 256 tiles at K = dim: the large-problem kernel runs in rounds of one tile per CU, with 256- or 192-row
    tiles, whichever gives fewer, shorter rounds, or whole rounds of 256-row tiles plus a tail launch of 192-row tiles (gemm_f16x3_p8s.hip); per layer to_qkv (12 column tiles), to_out (4), ff1 (16), ff2
    (4, K = 4 dim) and a skip combiner in every other layer (4, K = 2 dim).  A cost model for the packing only."""
    rows = 2 * int(frames)
    up = lambda a, b: -(-a // b)
    ct = lambda n: up(n, 256)

    def rounds(tn, f192=0.75):
        # (a round of 192-row tiles costs 0.75 of a round of 256-row tiles at K = dim, 0.8 at K = 2 dim / 4 dim: gemm_f16x3_p8s.hip)
        best = min(up(up(rows, 256) * tn, cus), f192 * up(up(rows, 192) * tn, cus))
        # whole rounds of 256-row tiles + one launch of 192-row tiles over the rest (round 6; the launch boundary costs ~ a quarter round)
        step = next((r for r in range(8, cus + 1, 8) if (r * tn) % cus == 0), cus)
        r256 = (rows // 256) // step * step
        rest = rows - 256 * r256
        if r256 > 0 and rest > 0:
            best = min(best, r256 * tn // cus + f192 * up(up(rest, 192) * tn, cus) + 0.25)
        return best
    n1 = ct(dim)
    return rounds(3 * n1) + rounds(n1) + rounds(4 * n1) + 4 * rounds(n1, 0.8) + 0.5 * 2 * rounds(n1, 0.8)


def choose_max_frames(lengths: Sequence[int], max_batch: int, cus: int = 256, candidates: Sequence[int] = (8192, 12288, 16384, 24576)) -> int:
    """Frames per launch for a directory: the candidate cap (scaled to the CUs the acoustic stage owns) whose first-fit-decreasing
    packing costs least under launch_cost - a directory whose tail would make a part-empty second launch is better off as one larger
    launch (more rounds: finer quantisation), e.g. 12,799 frames: 8,150 + 4,649 frames cost 13 + 9.75 units, one launch of 12,799 frames
    20.25.  Ties and gains below 1.5 % go to the smaller cap (less workspace).  Deterministic in the lengths: every rank chooses alike."""
    best, best_cost = None, None
    idx = list(range(len(lengths)))
    for c in candidates:
        cap = max(256, c * cus // 256)
        bins = pack_by_frames(idx, lengths, cap, max_batch)
        cost = sum(launch_cost(sum(int(lengths[i]) for i in b), cus) for b in bins)
        if best_cost is None or cost < best_cost * 0.985:          # (a bigger cap - more workspace - has to buy more than the model's noise)
            best, best_cost = cap, cost
    return int(best)


def group_by_padding(lengths: Sequence[int], max_waste: float = 0.25, max_group: int = 32) -> List[List[int]]:
    """Positions of `lengths` grouped so that zero-padding every group to ITS longest item costs at most max_waste of the group's
    real frames: n * max(len) <= (1 + max_waste) * sum(len).  The vocoder runs a ragged batch zero-padded to the longest item
    (its convolutions skip no tiles behind a short item's end), so an 8000-frame utterance that shares a bin with 31 fillers of
    192 frames would be vocoded as 32 x 8000 frames - 30x the work and ~170 GB of channels-last buffers (round-3 advisor finding).
    Longest first, greedy; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    groups: List[List[int]] = []
    cur: List[int] = []
    cur_max = cur_sum = 0
    for i in order:
        t = int(lengths[i])
        if cur and len(cur) < max_group and (len(cur) + 1) * cur_max <= (1.0 + max_waste) * (cur_sum + t):
            cur.append(i); cur_sum += t
        else:
            if cur:
                groups.append(cur)
            cur, cur_max, cur_sum = [i], t, t
    if cur:
        groups.append(cur)
    return groups


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
        g = rccl_group()
        cdev = torch.device("cpu") if g is None else device        # no RCCL (tests, or it failed): through host tensors on gloo
        flat = torch.cat([sd[k].to(device=cdev, dtype=torch.float32).reshape(-1) for k in bucket])
        dist.broadcast(flat, src=src, group=g)
        flat = flat.to(device)
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
    device = torch.device("cpu")                               # scalars: the default (gloo) group
    t = torch.tensor([frames], dtype=torch.float64, device=device)
    m = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(t.item()), float(m.item())


def gather_floats(values: Sequence[float], device: torch.device) -> List[List[float]]:
    """Every rank's list of floats, on every rank (diagnostics of a scaling run)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(values)]
    device = torch.device("cpu")
    buf = torch.tensor(list(values), dtype=torch.float64, device=device)
    allb = [torch.zeros_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(allb, buf)
    return [[float(x) for x in b] for b in allb]
