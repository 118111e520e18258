#!/usr/bin/env python3
"""Dev: WHERE the host thread spends a generation.run() pass of tools/ragged_dir.py - a sampler thread reads the main thread's stack
every 2 ms (second pass only) and prints the most frequent innermost four frames.  Found with it (round 5): the vocoder packed its weights
inside the first forward, with pageable host-to-device copies that waited for the whole solve enqueued before them (286 ms)."""
import os, sys, time, threading, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PASSES"] = "1"
import torch
from covomix_amd import generation
main_id = threading.get_ident()
samples = collections.Counter()
on = [False]
def sampler():
    while True:
        time.sleep(0.002)
        if not on[0]:
            continue
        fr = sys._current_frames().get(main_id)
        st = traceback.extract_stack(fr)
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st[-4:]))
        samples[key] += 1
threading.Thread(target=sampler, daemon=True).start()
run = generation.run
n = [0]
def run2(*a, **k):
    n[0] += 1
    on[0] = n[0] == 2
    r = run(*a, **k)
    on[0] = False
    return r
generation.run = run2
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ragged_dir.py"), run_name="__main__")
for k, v in samples.most_common(14):
    print(f"{v * 2:6d} ms  {k}", file=sys.__stderr__)
