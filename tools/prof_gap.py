"""Runs hf_build_gap_array and hf_decode_bare on a stream of [kind] [MiB] (for
rocprofv3 --kernel-trace / --pmc) and times them with the host clock."""
import sys
import time
import torch
import os
sys.path.insert(0, ".")
from huffman_amd import _lib
if os.environ.get("HF_TOOLS_LIB"):  # another build of the library, kept beside the production one
    _lib.LIB_PATH = os.path.abspath(os.environ["HF_TOOLS_LIB"])
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
data = synth.generate_device(kind, n, seed=2024)
codes, lens = hd.build_codebook(hd.histogram(data))
model = hd.DeviceModel(codes, lens)
enc = hd.encode(data, model)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb, so = hd.build_gap_array(enc.data, enc.nbytes, model, n)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    hd.decode_bare(enc.data, enc.nbytes, model, n, out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("ok", kind, torch.equal(cb[: n // 64], enc.chunk_bits[: n // 64]), torch.equal(out, data),
      "gap array %.3f ms, bare decode %.3f ms (host clock, second call)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
