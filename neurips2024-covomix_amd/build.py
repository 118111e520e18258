"""Build libcovomix_hip.so for gfx950 with hipcc (in-tree, so it travels with the snapshot)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
OUT = os.path.join(HERE, "libcovomix_hip.so")


def needs_build() -> bool:
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SOURCES + glob.glob(os.path.join(HERE, "csrc", "*.h")) + [os.path.join(HERE, "..", "include", "covomix_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcovomix_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
           *os.environ.get("CVX_HIPCC_FLAGS", "").split(),          # dev A/B builds (extra -D flags)
           "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=HERE)
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
