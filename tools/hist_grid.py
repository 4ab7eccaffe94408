import sys, torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1) << 30
data = synth.generate_device("zipf", n, seed=2024)
d_hist = torch.empty(256, dtype=torch.int64, device="cuda")
plan = hd.new_batch_hist(n, data.device)
ctx = hd.Context.get(data.device)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for g in (0, 256, 384, 512, 768, 1024, 1280, 2048):
    ctx.set_option("hist_grid", g)
    print(g, "plain %.4f  batches %.4f" % (t(lambda: hd.histogram_async(data, d_hist)), t(lambda: hd.histogram_async(data, d_hist, batch_hist=plan))), flush=True)
