"""Times hf_decode_bare (HIP-event kernel times, host clock) on one stream: python tools/time_bare.py [kind] [MiB]
(HF_TOOLS_LIB selects another build of the library)"""
import os
import sys
import time
import torch
sys.path.insert(0, ".")
from huffman_amd import _lib
if os.environ.get("HF_TOOLS_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HF_TOOLS_LIB"])
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
data = synth.generate_device(kind, n, seed=2024)
codes, lens = hd.build_codebook(hd.histogram(data))
model = hd.DeviceModel(codes, lens)
enc = hd.encode(data, model)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx = model.ctx
ctx.set_option("kernel_timing", 1)
hd.decode_bare(enc.data, enc.nbytes, model, n, out)
assert torch.equal(out, data)
torch.cuda.synchronize()
ctx.kernel_ms(3), ctx.kernel_ms(4)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    hd.decode_bare(enc.data, enc.nbytes, model, n, out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
a, b = ctx.kernel_ms(3), ctx.kernel_ms(4)
print(f"{kind} {n >> 20} MiB: bare decode {ms:.3f} ms per call; gap_sync_kernel {a[0] / reps:.3f} ms, gap_decode_kernel {b[0] / reps:.3f} ms")
