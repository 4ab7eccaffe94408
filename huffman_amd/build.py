"""Builds libhfgpu.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhfgpu.so")
SOURCES = [os.path.join(CSRC, "hfgpu.hip"), os.path.join(CSRC, "codebook.cpp")]
HEADERS = [os.path.join(ROOT, "include", "hfgpu.h"), os.path.join(CSRC, "hfgpu_internal.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    common = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
    objs = [os.path.join(CSRC, "hfgpu.o"), os.path.join(CSRC, "codebook.o")]
    dev = [hipcc(), "--offload-arch=gfx950", *common, *inc, "-c", SOURCES[0], "-o", objs[0]]
    if verbose:
        dev.insert(1, "-Rpass-analysis=kernel-resource-usage")
    host = ["g++", *common, *inc, "-c", SOURCES[1], "-o", objs[1]]
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    for cmd in (dev, host, link):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose="-v" in sys.argv)
    print(LIB)
