"""A/B of the planned encoder's two kernels on one box: the one-lookup kernel
(default) against the two-lookup kernel (context option debug = 512), with
their outputs compared: python tools/ab_once.py [kind ...] [--mib=N]
(kind skew<r>: 256 symbols with probabilities proportional to r^-rank)"""
import sys
import torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
mib = 1024
for a in sys.argv[1:]:
    if a.startswith("--mib="):
        mib = int(a[6:])
n = mib << 20
for kind in args or ["zipf", "testfile"]:
    if kind.startswith("skew"):  # skew<r>: 256 symbols, probabilities proportional to r^-rank
        import numpy as np
        w = float(kind[4:]) ** -np.arange(256, dtype=np.float64)
        cdf = torch.tensor(np.cumsum(w / w.sum()), device="cuda")
        data = torch.empty(n, dtype=torch.uint8, device="cuda")
        g = torch.Generator(device="cuda")
        g.manual_seed(2024)
        for off in range(0, n, 1 << 26):
            m = min(1 << 26, n - off)
            u = torch.rand(m, generator=g, device="cuda", dtype=torch.float64)
            data[off:off + m] = torch.bucketize(u, cdf, right=True).clamp_(max=255).to(torch.uint8)
    else:
        data = synth.generate_device(kind, n, seed=2024)
    plan = hd.new_batch_hist(n, data.device)
    d_hist = torch.empty(256, dtype=torch.int64, device="cuda")
    hd.histogram_async(data, d_hist, batch_hist=plan)
    hist = d_hist.cpu().numpy().astype("uint64")
    codes, lens = hd.build_codebook(hist)
    model = hd.DeviceModel(codes, lens)
    ctx = model.ctx
    res = {}
    for rnd in range(2):
        for dbg in (512, 0):
            ctx.set_option("debug", dbg)
            bufs = hd.EncodeBuffers(model, n, 64)
            ts = []
            for i in range(25):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                hd.encode_async(data, model, bufs, plan=plan)
                e1.record()
                torch.cuda.synchronize()
                if i >= 5:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            print(f"{kind} {mib} MiB debug {dbg:3d}: planned encode median {ts[len(ts) // 2]:.4f} min {ts[0]:.4f} ms",
                  flush=True)
            nbytes = (int(bufs.result[0].item()) + 7) // 8  # (the buffer's tail is not written)
            res[dbg] = (bufs.data[:nbytes].clone(), bufs.chunk_bits.clone(), bufs.seg_off.clone(),
                        bufs.result.clone())
            del bufs
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[512]))
    print(f"{kind}: outputs identical: {same}; flags {int(res[0][3][1])}", flush=True)
    ctx.set_option("debug", 0)
    del res, data, plan
    torch.cuda.empty_cache()
