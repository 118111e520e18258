#!/usr/bin/env python3
"""Dev: does a process that created CU-masked streams exit cleanly?  Runs variants in subprocesses and prints their exit codes
(PROFILE=1: every variant under rocprofv3 --kernel-trace).  Found in round 5: a pinned non_blocking copy on a masked stream + destroying
that stream at exit = SIGSEGV in torch's pinned-memory allocator when the pinned block outlives the stream (so t2s._decode_chunks reads its
state record through a device-side staging copy and a plain helper stream instead); NOT destroying the streams crashes rocprofv3's finaliser."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = {
    "create only": "part = ops.cu_partition(dev)",
    "create + kernels": "part = ops.cu_partition(dev)\nfor st in (part.main, part.side):\n    with torch.cuda.stream(st):\n        x = torch.ones(1024, device=dev) * 2\n    st.synchronize()",
    "create + kernels + no atexit": "import atexit\npart = ops.cu_partition(dev)\natexit.unregister(ops._destroy_partitions)\nfor st in (part.main, part.side):\n    with torch.cuda.stream(st):\n        x = torch.ones(1024, device=dev) * 2\n    st.synchronize()",
    "create + kernels + explicit destroy": "part = ops.cu_partition(dev)\nfor st in (part.main, part.side):\n    with torch.cuda.stream(st):\n        x = torch.ones(1024, device=dev) * 2\n    st.synchronize()\nops._destroy_partitions()",
    "thread + kernels": "import threading\npart = ops.cu_partition(dev)\ndef w():\n    with torch.cuda.stream(part.side):\n        y = torch.ones(1024, device=dev) * 3\n        part.side.synchronize()\nt = threading.Thread(target=w, daemon=True); t.start(); t.join()",
    "graph replay on side": "part = ops.cu_partition(dev)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(1024, device=dev)\n    part.side.synchronize()\n    g = torch.cuda.CUDAGraph()\n    cap = torch.cuda.Stream(device=dev)\n    with torch.cuda.graph(g, stream=cap):\n        y = x * 2\n    g.replay(); part.side.synchronize()",
    "pinned + event": "part = ops.cu_partition(dev)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(32, device=dev)\n    p = torch.empty(32).pin_memory(); ev = torch.cuda.Event(); p.copy_(x, non_blocking=True); ev.record(); ev.synchronize()",
    "pinned + event, no atexit": "import atexit\npart = ops.cu_partition(dev)\natexit.unregister(ops._destroy_partitions)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(32, device=dev)\n    p = torch.empty(32).pin_memory(); ev = torch.cuda.Event(); p.copy_(x, non_blocking=True); ev.record(); ev.synchronize()",
    "pinned + event, plain torch stream": "st = torch.cuda.Stream(device=dev)\nwith torch.cuda.stream(st):\n    x = torch.ones(32, device=dev)\n    p = torch.empty(32).pin_memory(); ev = torch.cuda.Event(); p.copy_(x, non_blocking=True); ev.record(); ev.synchronize()",
    "pinned + event, host cache emptied before exit": "part = ops.cu_partition(dev)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(32, device=dev)\n    p = torch.empty(32).pin_memory(); ev = torch.cuda.Event(); p.copy_(x, non_blocking=True); ev.record(); ev.synchronize()\ndel p, ev\ntorch.cuda.synchronize()\ntorch._C._host_emptyCache()",
    "pinned copy on side, no event": "part = ops.cu_partition(dev)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(32, device=dev)\n    p = torch.empty(32).pin_memory(); p.copy_(x, non_blocking=True); part.side.synchronize()",
    "event only on side": "part = ops.cu_partition(dev)\nwith torch.cuda.stream(part.side):\n    x = torch.ones(32, device=dev)\n    ev = torch.cuda.Event(); ev.record(); ev.synchronize()",
}
for name, body in BODY.items():
    code = f"import sys, torch\nsys.path.insert(0, {ROOT!r})\nfrom covomix_amd import ops\ndev = torch.device('cuda:0')\n{body}\nprint('body done', flush=True)\n"
    cmd = [sys.executable, "-c", code]
    if os.environ.get("PROFILE") == "1":
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", "/tmp/exit_probe_prof", "--"] + cmd
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
    print(f"{name:40s} rc={r.returncode}  {'body done' in r.stdout}  {([l for l in r.stderr.strip().splitlines() if 'amdgpu.ids' not in l] or [''])[-1][:150] if r.returncode else ''}", flush=True)
