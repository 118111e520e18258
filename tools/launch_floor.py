#!/usr/bin/env python3
"""Dev: the floor of a dependent launch chain inside a HIP graph on this box: 34 x 64 one-element kernels (a text2semantic step is
34 dependent launches), replayed.  us per launch = what a step pays before any of its kernels does work."""
import torch
x = torch.zeros(1, device="cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for n in (34 * 64,):
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (5 * n)
        print(f"{n} dependent one-element launches in a graph: {us:.2f} us per launch -> {34 * us:.0f} us per 34-launch step")
