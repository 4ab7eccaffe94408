import sys, time, torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth
data = synth.generate_device("zipf", 1 << 24, seed=1)
hist = hd.histogram(data)
t0 = time.perf_counter()
for _ in range(200):
    codes, lens = hd.build_codebook(hist)
t1 = time.perf_counter()
for _ in range(200):
    m = hd.DeviceModel(codes, lens)
torch.cuda.synchronize()
t2 = time.perf_counter()
d_hist = torch.empty(256, dtype=torch.int64, device="cuda")
for _ in range(200):
    hd.histogram_async(data[:4096], d_hist); h = d_hist.cpu()
t3 = time.perf_counter()
print("codebook %.1f us  model %.1f us  tiny hist + D2H %.1f us" % ((t1-t0)/200*1e6, (t2-t1)/200*1e6, (t3-t2)/200*1e6))
