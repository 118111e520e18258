"""BASELINE config 5 as a two-stage pipeline on one GPU: the text2semantic decode of batch k + 1 runs UNDER the acoustic solve
and the vocoder call of batch k.

The reference runs the three stages one after the other for every dialogue (dialogue_generation.py:272-329: comix_pred ->
covomix -> mel_decode_to_wav; the decode loop itself is covomix/covomix_model/text2semantic.py:749-848).  On MI355X the decode is a
latency chain - 34 dependent launches per token, each a few microseconds on a handful of CUs, 0.4-0.5 TB/s of weight streaming -
while the 64-NFE solve is a power-bound stream of persistent GEMM launches that own every CU they get.  Serially the chip idles
through a quarter of the step (round 4: 245 ms of 960).  Here the two stages run on streams restricted to DISJOINT compute units
(ops.CUPartition: hipExtStreamCreateWithCUMask, 28 + 4 CUs per XCD), each driven by its own host thread:

    side stream, worker thread:   stage1(item k + 1)   text2semantic (encoder + graph-replayed decode chunks + one eos read per chunk)
    main stream, calling thread:  stage2(stage1(item k))  acoustic solve + HiFi-GAN + int16 cast

A plain second stream does not do it: the large-problem GEMM is persistent (one block per CU, 136 KiB of LDS, 2 x 256 VGPRs x 8
waves), so a decode block would only be dispatched at a GEMM launch boundary - every 160-340 us for a chain whose links take 4-10 us.
The kernels of both stages are deterministic functions of their inputs whatever CUs they run on, so the pipelined schedule returns
the SAME BITS as running stage1 and stage2 alternately on the same two streams (`overlap=False`; tests/test_pipeline_gpu.py).
"""
from __future__ import annotations

import queue
import sys
import threading
from typing import Callable, Iterable, List, Optional

import torch

from . import ops


class _Failure:
    def __init__(self, exc: BaseException):
        self.exc = exc


def run_two_stage(items: Iterable, stage1: Callable, stage2: Callable, device=None, overlap: bool = True, depth: int = 2,
                  partition: Optional[ops.CUPartition] = None, collate: Optional[Callable] = None,
                  finish: Optional[Callable] = None) -> List:
    """[stage2(y) for y in collate(stage1(item) for item in items)], stage1 on the side stream of the device's CU partition and
    stage2 on its main stream.  collate (default: identity) is a generator function that turns the stream of stage-1 results into
    stage-2 inputs - the two stages need not batch alike (text2semantic decodes 8 utterances per pass whatever the acoustic batch is).
    overlap=True: stage1 runs in a worker thread, at most `depth` results ahead of stage2; overlap=False: the same calls on the same
    streams one after the other (the reference point of the bit-identity tests and of the speed-up).  A stage-1 result is handed over
    after its stream has drained (device tensors in it are safe to read on the main stream); stage2's results are returned after
    the main stream has drained.  An exception in either stage is re-raised in the calling thread.
    finish: stage2 only ENQUEUES its batch and returns a handle (results on their way into pinned memory behind an event);
    finish(handle) -> result is called after the NEXT batch has been enqueued, so the host work between two batches (assembly, input
    copies, the Python of ~2,500 launches) happens under the previous batch's kernels instead of between them."""
    part = partition if partition is not None else ops.cu_partition(device)
    dev = part.device
    out: List = []

    def consume(results):
        feed = collate(results) if collate is not None else results
        with torch.cuda.device(dev), torch.cuda.stream(part.main):
            pend = None
            for y in feed:
                h = stage2(y)
                if finish is None:
                    out.append(h)
                    continue
                if pend is not None:
                    out.append(finish(pend[0]))
                pend = (h,)
            if pend is not None:
                out.append(finish(pend[0]))
        part.main.synchronize()

    if not overlap:
        def results_serial():
            for it in items:
                with torch.cuda.device(dev), torch.cuda.stream(part.side):
                    x = stage1(it)
                    part.side.synchronize()
                yield x
        consume(results_serial())
        return out

    q: "queue.Queue" = queue.Queue(maxsize=max(1, int(depth)))
    with torch.cuda.device(dev):
        fill = _fill_stream(dev)
    stop = threading.Event()
    done = object()

    def worker():
        try:
            first = True
            for it in items:
                if stop.is_set():
                    return
                # the FIRST item has nothing to hide under (stage 2 waits for it): it runs on an unrestricted stream, all CUs, and the
                # pipeline fills in half the time (the decode's bits do not depend on the CUs it runs on)
                st = fill if first else part.side
                first = False
                with torch.cuda.device(dev), torch.cuda.stream(st):
                    x = stage1(it)
                    st.synchronize()
                    while not stop.is_set():
                        try:
                            q.put(x, timeout=0.1)
                            break
                        except queue.Full:
                            continue
            q.put(done)
        except BaseException as e:          # noqa: BLE001 - handed to the calling thread
            q.put(_Failure(e))

    def results_overlap():
        while True:
            x = q.get()
            if x is done:
                return
            if isinstance(x, _Failure):
                raise x.exc
            yield x

    # two Python threads share the interpreter lock: the solve's launch sequence is Python-heavy, and with the default 5 ms switch
    # interval the decode thread would wait that long every time it comes back from a stream wait
    interval = sys.getswitchinterval()
    sys.setswitchinterval(min(interval, 2e-4))
    t = threading.Thread(target=worker, name="covomix-t2s-stage", daemon=True)
    t.start()
    try:
        consume(results_overlap())
    finally:
        sys.setswitchinterval(interval)
        stop.set()
        while t.is_alive():                 # (unblock a worker waiting on a full queue, then let it finish its current call)
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=0.1)
    return out


_FILL: dict = {}


def _fill_stream(dev) -> "torch.cuda.Stream":
    """One plain (all CUs) stream per device for the first stage-1 item of a pipelined run."""
    key = torch.device(dev).index
    if key not in _FILL:
        _FILL[key] = torch.cuda.Stream(device=dev)
    return _FILL[key]


def regroup(n: int) -> Callable:
    """collate for run_two_stage: stage 1 yields LISTS of records (one per utterance), stage 2 gets lists of n of them (the last
    one shorter) in the same order."""
    def collate(results):
        pool: List = []
        for recs in results:
            pool.extend(recs)
            while len(pool) >= n:
                yield pool[:n]
                del pool[:n]
        if pool:
            yield pool
    return collate


def frames_per_launch(device=None, partition: Optional[ops.CUPartition] = None) -> int:
    """Frames per packed acoustic launch that fill whole rounds of the large-problem GEMM's 256-row panels on the main stream's CUs
    with both CFG branches: one row panel per main-stream CU for the N = 1024 products (4 column tiles), i.e. n_main / 4 panels of
    256 rows = n_main * 32 frames (7168 on 224 CUs; the unpartitioned chip: 8192)."""
    part = partition if partition is not None else ops.cu_partition(device)
    return part.n_main * 32
