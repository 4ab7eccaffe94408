"""Builds a variant of libhfgpu.so beside the production one for A/B timing
(tools/ab_libs.sh, HF_TOOLS_LIB): python tools/build_variant.py <name> [-DFLAG ...]
-> tools/_trace/lib_<name>.so"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
name, flags = sys.argv[1], sys.argv[2:]
os.makedirs("tools/_trace", exist_ok=True)
inc = ["-Iinclude", "-Ihuffman_amd/csrc"]
common = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
obj = f"tools/_trace/{name}.o"
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "--offload-arch=gfx950", *common, *inc, "-c",
                       "huffman_amd/csrc/hfgpu.hip", "-o", obj])
host_obj = f"tools/_trace/{name}_codebook.o"
subprocess.check_call(["g++", *common, *inc, "-c", "huffman_amd/csrc/codebook.cpp", "-o", host_obj])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, host_obj, "-o",
                       f"tools/_trace/lib_{name}.so"])
os.remove(obj)
os.remove(host_obj)
print(f"tools/_trace/lib_{name}.so")
