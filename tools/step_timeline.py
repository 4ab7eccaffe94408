"""Host-side timeline of one bench.py step (1 GiB Zipf, one GPU): where the time
between the kernels goes.  python tools/step_timeline.py [kind] [mib]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
dev = torch.device("cuda:0")
data = synth.generate_device(kind, n, seed=2024, device=dev)
d_hist = torch.empty(256, dtype=torch.int64, device=dev)
batch_hist = hd.new_batch_hist(n, dev)
out = torch.empty(n, dtype=torch.uint8, device=dev)
dres = torch.zeros(4, dtype=torch.int64, device=dev)
bufs = None
marks = []

def step(rec):
    global bufs
    t = [time.perf_counter()]
    hd.histogram_async(data, d_hist, batch_hist=batch_hist); t.append(time.perf_counter())
    hist = d_hist.cpu().numpy().astype(np.uint64); t.append(time.perf_counter())
    codes, lens = hd.build_codebook(hist); t.append(time.perf_counter())
    model = hd.DeviceModel(codes, lens, device=dev); t.append(time.perf_counter())
    if bufs is None:
        bufs = hd.EncodeBuffers(model, n, 64, out_cap=(n + 8 + 15) & ~15)
    hd.encode_async(data, model, bufs, lead_bits=0, plan=batch_hist); t.append(time.perf_counter())
    nbytes = (int((hist * lens).sum()) + 7) // 8; t.append(time.perf_counter())
    hd.decode_async(bufs.data, nbytes, model, n, 64, bufs.chunk_bits, bufs.seg_off, out, dres); t.append(time.perf_counter())
    if rec:
        marks.append(t)

for _ in range(5):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    step(True)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
m = np.diff(np.array(marks), axis=1).mean(axis=0) * 1e6
names = ["hist launch", "hist wait + D2H", "codebook", "model", "encode launch", "stream length", "decode launch"]
for k, v in zip(names, m):
    print("%-18s %7.1f us" % (k, v))
print("step %.1f us; host between hist result and encode launch %.1f us" % (el * 1e6, m[2] + m[3] + m[4]))
