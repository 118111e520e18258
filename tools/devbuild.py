#!/usr/bin/env python3
"""Dev: build a variant of libcovomix_hip.so WITHOUT touching the shipped one.
  python tools/devbuild.py NAME [-DFLAG ...] [--files a.hip,b.hip]
-> tools/dev_NAME.so (select with CVX_LIB_PATH).  The extra flags go to the listed files only (default: every file), objects are
cached under /tmp/cvx_devobj by (source, headers, flags), so a variant of one kernel recompiles one file."""
import glob, hashlib, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covomix_amd import build as B
name = sys.argv[1]
extra = [a for a in sys.argv[2:] if a.startswith("-")and not a.startswith("--files")]
files = None
for a in sys.argv[2:]:
    if a.startswith("--files="):
        files = set(a.split("=", 1)[1].split(","))
os.makedirs("/tmp/cvx_devobj", exist_ok=True)
def one(src):
    fl = B.BASE_FLAGS + (extra if (files is None or os.path.basename(src) in files) else [])
    key = B._digest([src] + B.HEADERS, fl)
    obj = f"/tmp/cvx_devobj/{os.path.basename(src)[:-4]}.{key}.o"
    if not os.path.isfile(obj):
        subprocess.run(["/opt/rocm/bin/hipcc", *fl, "-c", src, "-o", obj], check=True, stderr=subprocess.DEVNULL)
    return obj
with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(one, B.SOURCES))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"dev_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
