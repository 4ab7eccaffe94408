"""Runs a few iterations of the three kernels for rocprofv3 (no timing logic)."""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, ".")
from huffman_amd import _lib
if os.environ.get("HF_TOOLS_LIB"):  # another build of the library, kept beside the production one
    _lib.LIB_PATH = os.path.abspath(os.environ["HF_TOOLS_LIB"])
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
debug = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # the library's "debug" option (kernel choices)
n = mib << 20
hd.Context.get("cuda").set_option("debug", debug)
data = synth.generate_device(kind, n, seed=2024)
hist = hd.histogram(data)
codes, lens = hd.build_codebook(hist)
model = hd.DeviceModel(codes, lens)
bufs = hd.EncodeBuffers(model, n, cs)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
dres = torch.zeros(4, dtype=torch.int64, device="cuda")
d_hist = torch.empty(256, dtype=torch.int64, device="cuda")
nbytes = (int((hist * lens).sum()) + 7) // 8
plan = hd.new_batch_hist(n, data.device)  # the path bench.py times by default
for _ in range(iters):
    hd.histogram_async(data, d_hist, batch_hist=plan)
    hd.encode_async(data, model, bufs, plan=plan)
    hd.decode_async(bufs.data, nbytes, model, n, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
torch.cuda.synchronize()
assert os.environ.get("HF_TOOLS_NOCHECK") or torch.equal(out, data)  # (experiments that decode wrongly on purpose)
print("ok", kind, mib, cs, hd.Context.get("cuda").decode_kernel())
