import sys, numpy as np, torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth
for kind, n in (("zipf", 1 << 28), ("zipf", 1 << 30)):
    data = synth.generate_device(kind, n, seed=2024)
    codes, lens = hd.build_codebook(hd.histogram(data))
    model = hd.DeviceModel(codes, lens)
    enc = hd.encode(data, model)
    out = hd.decode_bare(enc.data, enc.nbytes, model, n)
    d = data.cpu().numpy(); o = out.cpu().numpy()
    bad = np.nonzero(d != o)[0]
    print(kind, n, "mismatches", bad.size, bad[:10], "maxlen", int(lens.max()))
    if bad.size:
        i = bad[0]
        print(" want", d[max(0,i-8):i+24]); print(" got ", o[max(0,i-8):i+24])
        l = np.asarray(lens)[d[: i + 1]].astype(np.int64); cum = np.concatenate([[0], np.cumsum(l)])
        print(" mismatch runs:", np.nonzero(np.diff(bad) > 1)[0].size + 1, "last bad", bad[-1])
        print(" first bad symbol", i, "starts at bit", cum[i], "sub", cum[i] // 512, "offset in sub", cum[i] % 512)
