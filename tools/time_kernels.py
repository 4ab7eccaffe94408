"""Times the three kernels back to back (HIP events, many repetitions) for A/B
comparisons of kernel changes: python tools/time_kernels.py [kind] [mib] [reps]
(HF_TOOLS_LIB=path/to/other/libhfgpu.so times another build on the same box)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import huffman_amd._lib as _lib
if os.environ.get("HF_TOOLS_LIB"):  # A/B runs: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["HF_TOOLS_LIB"])
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = mib << 20
if os.environ.get("HF_TOOLS_DEBUG"):  # the library's "debug" option (kernel choices), set before the model exists
    hd.Context.get("cuda").set_option("debug", int(os.environ["HF_TOOLS_DEBUG"]))
data = synth.generate_device(kind, n, seed=2024)
hist = hd.histogram(data)
codes, lens = hd.build_codebook(hist)
model = hd.DeviceModel(codes, lens)
bufs = hd.EncodeBuffers(model, n, 64)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
dres = torch.zeros(4, dtype=torch.int64, device="cuda")
d_hist = torch.empty(256, dtype=torch.int64, device="cuda")
nbytes = (int((hist * lens).sum()) + 7) // 8


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


timed(lambda: hd.encode_async(data, model, bufs))  # clocks settle during the first passes
h = timed(lambda: hd.histogram_async(data, d_hist))
e = timed(lambda: hd.encode_async(data, model, bufs))
plan = hd.new_batch_hist(n, data.device)
hb = timed(lambda: hd.histogram_async(data, d_hist, batch_hist=plan))
ep = timed(lambda: hd.encode_async(data, model, bufs, plan=plan))
d = timed(lambda: hd.decode_async(bufs.data, nbytes, model, n, 64, bufs.chunk_bits, bufs.seg_off, out, dres))
assert os.environ.get("HF_TOOLS_NOCHECK") or torch.equal(out, data)  # (experiments that decode wrongly on purpose)
print(f"{kind} {mib} MiB  median/min ms: hist {h[0]:.4f}/{h[1]:.4f}  enc {e[0]:.4f}/{e[1]:.4f}  dec {d[0]:.4f}/{d[1]:.4f}"
      f"  | hist+batches {hb[0]:.4f}/{hb[1]:.4f}  enc planned {ep[0]:.4f}/{ep[1]:.4f}")
