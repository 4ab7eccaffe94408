#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (hf_encode_host /
hf_decode_host): host slice in, host slice out, one synchronous call each.
Reported next to bench.py's HBM-resident numbers, never instead of them."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huffman_amd import _lib, synth  # noqa: E402
from huffman_amd._lib import u8p, u64p  # noqa: E402

u16p = C.POINTER(C.c_uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--kind", default="zipf")
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    L = _lib.lib()
    n = a.mib << 20
    dev_data = synth.generate_device(a.kind, n, seed=2024, device=torch.device("cuda", 0))

    def host_buf(nbytes, dtype=torch.uint8):
        t = torch.empty(nbytes, dtype=dtype, pin_memory=a.pinned)
        return t

    data_t = host_buf(n)
    data_t.copy_(dev_data.cpu())
    del dev_data
    ctx = C.c_void_p()
    assert L.hf_ctx_create(0, C.byref(ctx)) == 0
    hist = np.zeros(256, dtype=np.uint64)
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    t0 = time.perf_counter()
    assert L.hf_histogram_host(ctx, C.cast(data_t.data_ptr(), u8p), n, hist.ctypes.data_as(u64p)) == 0
    t_hist = time.perf_counter() - t0
    assert L.hf_build_codebook(hist.ctypes.data_as(u64p), 256, codes.ctypes.data_as(u64p),
                               lens.ctypes.data_as(u64p)) == 0
    model = C.c_void_p()
    assert L.hf_model_create(ctx, codes.ctypes.data_as(u64p), lens.ctypes.data_as(u64p),
                             C.byref(model)) == 0
    cap = int(L.hf_encode_bound(model, n, 0))
    out_t = host_buf(cap)
    nc, ns = C.c_uint64(), C.c_uint64()
    L.hf_gap_layout(n, 64, C.byref(nc), C.byref(ns))
    cb_t = host_buf(nc.value, torch.int16)
    so_t = host_buf(ns.value, torch.int64)
    back_t = host_buf(n)
    bits = C.c_uint64()
    enc, dec = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        rc = L.hf_encode_host(ctx, model, C.cast(data_t.data_ptr(), u8p), n, 0,
                              C.cast(out_t.data_ptr(), u8p), cap, 64,
                              C.cast(cb_t.data_ptr(), u16p), C.cast(so_t.data_ptr(), u64p),
                              C.byref(bits))
        enc.append(time.perf_counter() - t0)
        assert rc == 0
        nbytes = (bits.value + 7) // 8
        t0 = time.perf_counter()
        rc = L.hf_decode_host(ctx, model, C.cast(out_t.data_ptr(), u8p), nbytes, n, 64,
                              C.cast(cb_t.data_ptr(), u16p), C.cast(so_t.data_ptr(), u64p),
                              C.cast(back_t.data_ptr(), u8p))
        dec.append(time.perf_counter() - t0)
        assert rc == 0
    assert torch.equal(back_t, data_t)
    # the whole compress pipeline: separate calls (two uploads) against the fused one
    sep, fused = [], []
    c2 = np.zeros(256, dtype=np.uint64)
    l2 = np.zeros(256, dtype=np.uint64)
    for _ in range(a.reps):
        t0 = time.perf_counter()
        assert L.hf_histogram_host(ctx, C.cast(data_t.data_ptr(), u8p), n, hist.ctypes.data_as(u64p)) == 0
        assert L.hf_build_codebook(hist.ctypes.data_as(u64p), 256, c2.ctypes.data_as(u64p),
                                   l2.ctypes.data_as(u64p)) == 0
        m2 = C.c_void_p()
        assert L.hf_model_create(ctx, c2.ctypes.data_as(u64p), l2.ctypes.data_as(u64p), C.byref(m2)) == 0
        assert L.hf_encode_host(ctx, m2, C.cast(data_t.data_ptr(), u8p), n, 0,
                                C.cast(out_t.data_ptr(), u8p), cap, 64,
                                C.cast(cb_t.data_ptr(), u16p), C.cast(so_t.data_ptr(), u64p),
                                C.byref(bits)) == 0
        L.hf_model_destroy(m2)
        sep.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        assert L.hf_compress_host(ctx, C.cast(data_t.data_ptr(), u8p), n, c2.ctypes.data_as(u64p),
                                  l2.ctypes.data_as(u64p), C.cast(out_t.data_ptr(), u8p), cap, 64,
                                  C.cast(cb_t.data_ptr(), u16p), C.cast(so_t.data_ptr(), u64p),
                                  C.byref(bits)) == 0
        fused.append(time.perf_counter() - t0)
    print(json.dumps({
        "workload": f"{a.mib} MiB {a.kind}, host buffers {'pinned' if a.pinned else 'pageable'}",
        "histogram_host_GBps": round(n / t_hist / 1e9, 2),
        "encode_host_GBps": round(n / min(enc) / 1e9, 2),
        "decode_host_GBps": round(n / min(dec) / 1e9, 2),
        "compress_separate_calls_GBps": round(n / min(sep) / 1e9, 2),
        "compress_host_GBps": round(n / min(fused) / 1e9, 2),
        "compressed_ratio": round(nbytes / n, 4),
    }))
    L.hf_model_destroy(model)
    L.hf_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
