"""Encoder time with and without the gap-array outputs (1 GiB Zipf, planned)."""
import sys, torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth
n = 1 << 30
data = synth.generate_device(sys.argv[1] if len(sys.argv) > 1 else "zipf", n, seed=2024)
bh = hd.new_batch_hist(n, data.device)
hist = hd.histogram_async(data, batch_hist=bh).cpu().numpy()
codes, lens = hd.build_codebook(hist)
model = hd.DeviceModel(codes, lens)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for gap in (True, False, True, False):
    bufs = hd.EncodeBuffers(model, n, 64, with_gap=gap, out_cap=n + 64)
    print("gap arrays", gap, "planned %.4f ms" % t(lambda: hd.encode_async(data, model, bufs, plan=bh)), flush=True)
