"""Runs hf_build_gap_array once on a 1 GiB stream (for rocprofv3 --kernel-trace)."""
import sys
import torch
sys.path.insert(0, ".")
from huffman_amd import device as hd, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n = 1 << 30
data = synth.generate_device(kind, n, seed=2024)
codes, lens = hd.build_codebook(hd.histogram(data))
model = hd.DeviceModel(codes, lens)
enc = hd.encode(data, model)
cb, so = hd.build_gap_array(enc.data, enc.nbytes, model, n)
torch.cuda.synchronize()
print("ok", torch.equal(cb[: n // 64], enc.chunk_bits[: n // 64]))
