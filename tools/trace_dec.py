"""Phase timeline of the decode kernel.  Needs a library built with
HF_DEC_TRACE=1 python huffman_amd/build.py: threads 0 and 192 of every
work-group stamp the wall clock at every phase boundary into the buffer given
by the context option "dec_trace".  python tools/trace_dec.py [kind]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, ".")
from huffman_amd import _lib
if os.environ.get("HF_TOOLS_LIB"):  # the trace build, kept beside the production library
    _lib.LIB_PATH = os.path.abspath(os.environ["HF_TOOLS_LIB"])
from huffman_amd import device as hd, synth

n = 1 << 30
kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
data = synth.generate_device(kind, n, seed=2024)
hist = hd.histogram(data)
codes, lens = hd.build_codebook(hist)
model = hd.DeviceModel(codes, lens)
bufs = hd.EncodeBuffers(model, n, 64)
ctx = model.ctx
hd.encode_async(data, model, bufs)
nbytes = (int((hist * lens).sum()) + 7) // 8
out = torch.empty(n, dtype=torch.uint8, device="cuda")
dres = torch.zeros(4, dtype=torch.int64, device="cuda")
n_seg = n // 16384
trace = torch.zeros(n_seg * 16, dtype=torch.int64, device="cuda")
if len(sys.argv) > 2:
    ctx.set_option("debug", int(sys.argv[2]))
ctx.set_option("dec_trace", trace.data_ptr())
for _ in range(3):
    hd.decode_async(bufs.data, nbytes, model, n, 64, bufs.chunk_bits, bufs.seg_off, out, dres)
torch.cuda.synchronize()
ctx.set_option("dec_trace", 0)
assert torch.equal(out, data)
tr = trace.cpu().numpy().reshape(n_seg, 2, 8)
a, b = tr[:, 0, :], tr[:, 1, :]
print("kernel span us:", (tr[:, :, 7].max() - tr[:, :, 0].min()) / 100.0)
life = (a[:, 7] - a[:, 0]) / 100.0
print("WG lifetime us: mean %.2f median %.2f p10 %.2f p90 %.2f" % (life.mean(), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
spans = [(0, 1, "gap array+table+scan"), (1, 2, "stage stream image"), (2, 3, "barrier"),
         (3, 4, "decode the chunk"), (4, 7, "transpose + store (dec_kernel_quad: barrier, rows out)")]
sel = slice(4000, n_seg - 2000)
for who, x in (("thread 0 (wave 0)", a), ("thread 192 (wave 3)", b)):
    print(who)
    for k0, k1, name in spans:
        d = (x[sel, k1] - x[sel, k0]) / 100.0
        print("  %-22s mean %6.2f  median %6.2f  p90 %6.2f us" % (name, d.mean(), np.median(d), np.percentile(d, 90)))
mid = (a[:, 0].min() + a[:, 7].max()) // 2
print("WGs alive at mid-kernel:", int(((a[:, 0] <= mid) & (a[:, 7] >= mid)).sum()))
