"""Phase timeline of the encode kernel.  Needs a library built with
HF_ENC_TRACE=1 python huffman_amd/build.py: under the context option debug=16
its kernel stamps the wall clock at every phase boundary into the tail of the
output buffer.  python tools/trace_enc.py [kind] [planned]
(The stamps are in enc_pack_kernel; a planned encode is sent there with debug
bit 512 instead of to the one-lookup kernel.)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth

n = 1 << 30
kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
data = synth.generate_device(kind, n, seed=2024)
hist = hd.histogram(data)
codes, lens = hd.build_codebook(hist)
model = hd.DeviceModel(codes, lens)
bufs = hd.EncodeBuffers(model, n, 64)
ctx = model.ctx
plan = None
if len(sys.argv) > 2 and sys.argv[2] == "planned":
    plan = hd.new_batch_hist(n, data.device)
    hd.histogram_async(data, batch_hist=plan)
ctx.set_option("debug", 16 | 512)
for _ in range(3):
    hd.encode_async(data, model, bufs, plan=plan)
torch.cuda.synchronize()
ctx.set_option("debug", 0)
n_batches = n // 65536
cap = bufs.data.numel()
off = (cap - (n // 16384 // 4 + 2) * 320) & ~15
assert (int((hist * lens).sum()) + 7) // 8 < off, "stream reaches into the trace area"
tr = bufs.data[off: off + n_batches * 320].cpu().numpy().view(np.uint64).reshape(n_batches, 2, 20).astype(np.int64)
t00 = tr[:, 0, 0].min()
a = tr[:, 0, :]  # thread 0
b = tr[:, 1, :]  # thread 448
print("kernel span us:", (tr[:, :, 18].max() - t00) / 100.0)
life = (a[:, 18] - a[:, 0]) / 100.0
print("WG lifetime us: mean %.2f median %.2f p10 %.2f p90 %.2f" % (life.mean(), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
names = ["start->ticket+staging", "loads+len sums", "barrier", "scan", "lookback (wave0)", "barrier",
         "pack t0", None, "finish t0", "pack t1", None, "finish t1", "pack t2", None, "finish t2", "pack t3", None, "finish t3"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 13, 15, 16, 18]
lab = ["ticket+staging", "loads+len sums", "barrier1", "scan", "lookback", "barrier2", "pack0", "finish0", "pack1", "finish1",
       "pack2", "finish2", "pack3", "finish3"]
sel = slice(2000, n_batches - 1000)
for who, x in (("thread 0 (wave 0)", a), ("thread 448 (wave 7)", b)):
    print(who)
    for k in range(len(idx) - 1):
        d = (x[sel, idx[k + 1]] - x[sel, idx[k]]) / 100.0
        print("  %-16s mean %6.2f  median %6.2f  p90 %6.2f us" % (lab[k], d.mean(), np.median(d), np.percentile(d, 90)))
pl = a[sel, 19] & 0xffffffff; st = a[sel, 19] >> 32
if plan is None:
    print("lookback polls mean %.2f p90 %d; steps mean %.2f p90 %d" % (pl.mean(), np.percentile(pl, 90), st.mean(), np.percentile(st, 90)))
# concurrency: batches alive at mid time
mid = (a[:, 0].min() + a[:, 18].max()) // 2
alive = ((a[:, 0] <= mid) & (a[:, 18] >= mid)).sum()
print("WGs alive at mid-kernel:", int(alive))
if plan is None:
    # how late are the predecessors' totals compared with this batch's own?
    t4 = a[:, 4].astype(np.float64) / 100.0
    t5 = a[:, 5].astype(np.float64) / 100.0
    for win in (8, 32, 64, 128):
        idxs = np.arange(2000, n_batches - 1000)
        m = np.max(np.stack([t4[idxs - k] for k in range(1, win + 1)]), axis=0)
        lag = m - t4[idxs]
        tail = t5[idxs] - np.maximum(m, t4[idxs])
        print("window %3d: slowest predecessor total vs own: mean %+.2f us (p10 %+.2f, p90 %+.2f); look-back end after both: mean %.2f us"
              % (win, lag.mean(), np.percentile(lag, 10), np.percentile(lag, 90), tail.mean()))
    start = a[:, 0].astype(np.float64) / 100.0
    print("start spacing to predecessor: mean %.3f us, p10 %.3f p90 %.3f" % ((start[idxs] - start[idxs - 1]).mean(), np.percentile(start[idxs] - start[idxs - 1], 10), np.percentile(start[idxs] - start[idxs - 1], 90)))
