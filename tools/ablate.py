"""Times hf_encode_async with phases skipped (ctx option "debug"); results of
those runs are wrong on purpose."""
import sys
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
def run(mode, dbg, reps=5):
    ctx.set_option("enc_mode", mode)
    ctx.set_option("debug", dbg)
    for _ in range(2):
        hd.encode_async(data, model, bufs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hd.encode_async(data, model, bufs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
names = {0: "full", 4: "full, blockIdx as ticket", 9: "stageA only (no lookback)", 13: "stageA only, blockIdx ticket",
         1: "stageA+lookback", 5: "stageA+lookback, blockIdx ticket", 2: "no pack (finish only)",
         8: "full minus lookback"}
for mode in (1,):
    for dbg, name in names.items():
        print(f"mode {mode} debug {dbg:2d} {name:28s} {run(mode, dbg):.3f} ms", flush=True)
ctx.set_option("debug", 0)
