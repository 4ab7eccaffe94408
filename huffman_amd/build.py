"""Builds libhfgpu.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhfgpu.so")
SOURCES = [os.path.join(CSRC, "hfgpu.hip"), os.path.join(CSRC, "codebook.cpp")]
HEADERS = [os.path.join(ROOT, "include", "hfgpu.h"), os.path.join(CSRC, "hfgpu_internal.h")] + [
    os.path.join(CSRC, h) for h in ("dev_common.hpp", "kern_hist.hpp", "kern_encode.hpp",
                                    "kern_decode.hpp", "kern_gap.hpp")]
# host programs on top of the C++ mirror (include/hfgpu.hpp):
# (source, executable, extra compiler flags, extra linker flags)
ORACLE_DIR = os.path.join(ROOT, "oracle")
PROGRAMS = [
    (os.path.join(HERE, "cli", "main.cpp"), os.path.join(HERE, "bin", "hfgpu"), [], []),
    (os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp"),
     os.path.join(ROOT, "tests", "cpp", "test_mirror"), [], []),
    (os.path.join(ROOT, "tools", "src", "adaptor_rate.cpp"),
     os.path.join(ROOT, "tools", "bin", "adaptor_rate"), [], []),
    # test program: the mirror's adaptors with small pieces against the CPU oracle
    (os.path.join(ROOT, "tests", "fuzz", "fuzz_mirror.cpp"),
     os.path.join(ROOT, "tests", "fuzz", "fuzz_mirror"), ["-DHFGPU_PIECE_DIVISOR=4096"],
     ["-L" + ORACLE_DIR, "-loracle", "-Wl,-rpath,$ORIGIN/../../oracle"]),
]


# device micro-tests that a GPU test runs: (source, executable)
UBENCH = [(os.path.join(ROOT, "tools", "ubench", "shift_hazard.hip"),
           os.path.join(ROOT, "tools", "ubench", "shift_hazard"))]


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


def build_programs(force: bool = False, verbose: bool = False) -> list[str]:
    """Host-only C++ programs; they reach the GPU through libhfgpu.so alone."""
    hpp = os.path.join(ROOT, "include", "hfgpu.hpp")
    out = []
    for src, exe, cflags, ldflags in PROGRAMS:
        out.append(exe)
        deps = [src, hpp, LIB, *HEADERS]
        if ldflags and not os.path.exists(os.path.join(ORACLE_DIR, "liboracle.so")):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        if not force and os.path.exists(exe) and all(
                os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
            continue
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        obj = exe + ".o"
        rel = os.path.relpath(HERE, os.path.dirname(exe))
        cmds = [
            ["g++", "-O2", "-std=c++17", "-Wall", *cflags, "-I" + os.path.join(ROOT, "include"), "-c",
             src, "-o", obj],
            [hipcc(), obj, "-o", exe, "-L" + HERE, "-lhfgpu", "-Wl,-rpath,$ORIGIN/" + rel, *ldflags],
        ]
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        os.remove(obj)
    return out


def build_ubench(force: bool = False, verbose: bool = False) -> None:
    for src, exe in UBENCH:
        if not force and os.path.exists(exe) and os.path.getmtime(src) <= os.path.getmtime(exe):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-w", "-o", exe, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False) -> str:
    build_ubench(force, verbose)
    if not force and not stale():
        build_programs(False, verbose)
        return LIB
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    common = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
    objs = [os.path.join(CSRC, "hfgpu.o"), os.path.join(CSRC, "codebook.o")]
    # the compiler's intermediate files are kept for the duration of the build: the
    # device assembly of this very compilation is scanned below
    temps = tempfile.mkdtemp(prefix="hfgpu_build_")
    tmp_obj = os.path.join(temps, "hfgpu.o")
    dev = [hipcc(), "--offload-arch=gfx950", *common, *inc, "-save-temps=obj", "-c", SOURCES[0], "-o",
           tmp_obj]
    if verbose:
        dev.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if os.environ.get("HF_ENC_TRACE") == "1":  # phase stamps in the encode kernel (tools/trace_enc.py)
        dev.insert(1, "-DHF_ENC_TRACE")
    if os.environ.get("HF_DEC_TRACE") == "1":  # the same for the decode kernel (tools/trace_dec.py)
        dev.insert(1, "-DHF_DEC_TRACE")
    host = ["g++", *common, *inc, "-c", SOURCES[1], "-o", objs[1]]
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    try:
        if verbose:
            print(" ".join(dev), file=sys.stderr)
        subprocess.check_call(dev)
        check_device_assembly(temps)
        shutil.move(tmp_obj, objs[0])
    finally:
        shutil.rmtree(temps, ignore_errors=True)
    for cmd in (host, link):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    build_programs(True, verbose)
    return LIB


def check_device_assembly(temps: str) -> None:
    """gfx950 mis-reads the amount of a 64-bit shift that sits in the last VGPR of a
    wave's allocation (tools/check_isa.py, tools/ubench/shift_hazard.hip): a build whose
    kernels contain that pattern is refused here, with the kernels named."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_isa
    finally:
        sys.path.pop(0)
    hits = []
    for name in os.listdir(temps):
        if name.endswith(".s") and "amdgcn" in name:
            hits += check_isa.scan(os.path.join(temps, name))
    if hits:
        raise RuntimeError("64-bit shift with its amount in the last allocated VGPR (see "
                           "HF_TOUCH_VGPR in huffman_amd/csrc/dev_common.hpp):\n  " + "\n  ".join(hits))


if __name__ == "__main__":
    build(force=True, verbose="-v" in sys.argv)
    print(LIB)
