"""Build libcovomix_hip.so for gfx950 with hipcc (in-tree, so it travels with the snapshot).

Every csrc/*.hip is compiled to its own object (in parallel, cached by a hash of the source, the headers and the flags
under csrc/.obj/) and the objects are linked into one shared library.  `source_hash()` identifies what a given .so was
built from: build_library() writes it next to the library and prints it, so a driver can tell a rebuild from a reuse."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HEADERS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(HERE, "..", "include", "covomix_hip.h")]
OUT = os.path.join(HERE, "libcovomix_hip.so")
STAMP = OUT + ".srchash"
OBJ_DIR = os.path.join(HERE, "csrc", ".obj")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _flags() -> list:
    return BASE_FLAGS + os.environ.get("CVX_HIPCC_FLAGS", "").split()          # dev A/B builds (extra -D flags)


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(e.encode())
    return h.hexdigest()[:16]


def source_hash() -> str:
    """Hash of every source, header and compile flag that goes into the library."""
    return _digest(SOURCES + HEADERS, _flags())


def needs_build() -> bool:
    if not os.path.isfile(OUT) or not os.path.isfile(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcovomix_hip.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = _flags()

    def compile_one(src: str) -> str:
        key = _digest([src] + HEADERS, flags)
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + "." + key + ".o")
        if force or not os.path.isfile(obj):
            for old in glob.glob(os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".*.o")):
                os.remove(old)
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True, cwd=HERE)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, check=True, cwd=HERE)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True), source_hash())
