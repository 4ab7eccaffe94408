"""Randomised differential run of the host-buffer entry points, of arbitrary
(also not prefix-free) models and of sharded streams against the CPU oracle.
python tests/fuzz/fuzz_host.py [cases] [first seed]

Per case one of:
  model   a random table of distinct left-aligned keys (key 0 included, codes
          may be prefixes of each other): the oracle encodes random symbols, the
          library rebuilds the gap array from the bare stream and decodes, and
          must return what the reference's sequential predecessor decode returns;
  host    hf_compress_host / hf_encode_host / hf_decode_host (with and without
          gap arrays) / hf_decode_window_host / hf_decode_range_host with random
          piece sizes, windows and run lengths;
  shards  one stream cut into 2..8 shards at random places, each encoded with its
          lead bits, placed with hf_place_shard in random order, decoded per shard
          with bit_base.
A search tool, not part of the test suite."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle as orc  # noqa: E402
from huffman_amd import _lib, device as dev  # noqa: E402
from huffman_amd._lib import u8p, u64p  # noqa: E402

orc.build()
L = _lib.lib()
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CTX = dev.Context.get(torch.device("cuda:0"))
H = CTX.handle


def check(rc, what):
    if rc != _lib.HF_OK:
        raise AssertionError(f"{what}: status {rc} {L.hf_last_error(H).decode()}")


def random_data(rng, n):
    a = int(rng.choice([1, 2, 5, 17, 64, 256]))
    k = np.arange(1, a + 1, dtype=np.float64)
    w = k ** -rng.uniform(0.0, 2.5)
    perm = rng.permutation(256)[:a].astype(np.uint8)
    return perm[rng.choice(a, size=n, p=w / w.sum())]


def case_model(rng, tag):
    nsym = int(rng.integers(2, 40))
    maxlen = int(rng.integers(2, 40))
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    syms = rng.permutation(256)[:nsym]
    keys = set()
    used = []
    for i, s in enumerate(syms):
        for _ in range(50):
            ln = int(rng.integers(1, maxlen + 1))
            code = 0 if i == 0 else int(rng.integers(0, 1 << ln))
            key = code << (maxlen - ln)
            if key not in keys:
                break
        else:
            continue
        keys.add(key)
        codes[s], lens[s] = code, ln
        used.append(s)
    sent = int(lens.max())
    # keys are aligned to the real sentinel (the longest length in use)
    if len({int(codes[s]) << (sent - int(lens[s])) for s in used}) != len(used):
        return tag + " (duplicate keys after alignment, skipped)"
    n = int(np.exp(rng.uniform(0, np.log(300000))))
    data = np.array(used, dtype=np.uint8)[rng.integers(0, len(used), size=n)]
    stream, _ = orc.encode_slice(data, codes, lens)
    try:
        want = orc.decode_read(stream, codes, lens, n)
    except ValueError as e:  # the reference's own debug assertion (decode/mod.rs:207-210)
        return tag + f" (reference refuses: {e})"
    # with codes that are prefixes of each other the reference's loop can return more than
    # goalsbyte symbols (it only stops at the end of the data); the library returns n_out
    assert want.size >= n
    want = want[:n]
    tag += f" nsym={len(used)} sentinel={sent} n={n}"
    model = dev.DeviceModel(codes, lens)
    back = np.zeros(n, dtype=np.uint8)
    rc = L.hf_decode_host(H, model.handle, stream.ctypes.data_as(u8p), stream.size, n, 64, None, None,
                          back.ctypes.data_as(u8p))
    check(rc, tag + " hf_decode_host without gap array")
    if not np.array_equal(back, want):
        bad = np.nonzero(back != want)[0]
        raise AssertionError(tag + f" decode differs from the reference's at {bad[:6].tolist()} ({bad.size})")
    if n <= 50000:
        t = torch.from_numpy(np.concatenate([stream, np.zeros(16, np.uint8)])).cuda()
        got = dev.decode_serial(t, stream.size, model, n).cpu().numpy()
        assert np.array_equal(got, want), tag + " serial decode"
    return tag + " ok"


def case_host(rng, tag):
    n = int(np.exp(rng.uniform(0, np.log(6e6))))
    data = random_data(rng, n)
    cs = int(rng.choice([64, 128, 256]))
    piece = int(rng.choice([1, 1, 2, 64]))
    CTX.set_option("host_piece_mib", piece)
    tag += f" n={n} cs={cs} piece={piece}"
    hist = orc.histogram(data)
    ocodes, olens = orc.generate_extended_codewords(hist)
    want, _ = orc.encode_slice(data, ocodes, olens)
    nc, ns = dev.gap_layout(n, cs)
    cb = np.zeros(max(nc, 1), dtype=np.uint16)
    so = np.zeros(max(ns, 1), dtype=np.uint64)
    out = np.full(n + 64, 0x5A, dtype=np.uint8)
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    bits = C.c_uint64()
    check(L.hf_compress_host(H, data.ctypes.data_as(u8p), n, codes.ctypes.data_as(u64p),
                             lens.ctypes.data_as(u64p), out.ctypes.data_as(u8p), out.size, cs,
                             cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p), C.byref(bits)),
          tag + " hf_compress_host")
    assert np.array_equal(codes, ocodes) and np.array_equal(lens, olens), tag + " codebook"
    nbytes = (bits.value + 7) // 8
    assert nbytes == want.size and np.array_equal(out[:nbytes], want), tag + " compress stream"
    model = dev.DeviceModel(codes, lens)
    lead = int(rng.choice([0, 0, 5]))
    out2 = np.full(n + 64, 0xC3, dtype=np.uint8)
    cb2 = np.zeros_like(cb)
    so2 = np.zeros_like(so)
    check(L.hf_encode_host(H, model.handle, data.ctypes.data_as(u8p), n, lead, out2.ctypes.data_as(u8p),
                           out2.size, cs, cb2.ctypes.data_as(u16p), so2.ctypes.data_as(u64p),
                           C.byref(bits)), tag + " hf_encode_host")
    if lead == 0:
        assert np.array_equal(out2[:nbytes], want), tag + " encode_host stream"
        assert np.array_equal(cb2[:nc], cb[:nc]) and np.array_equal(so2[:ns], so[:ns]), tag + " gap arrays"
    else:
        allbits = np.concatenate([np.zeros(lead, np.uint8), np.unpackbits(want)[: bits.value]])
        assert np.array_equal(out2[: (lead + bits.value + 7) // 8], np.packbits(allbits)), tag + " lead"
    stream = np.ascontiguousarray(out[:nbytes])
    for gap in (True, False):
        back = np.zeros(n, dtype=np.uint8)
        check(L.hf_decode_host(H, model.handle, stream.ctypes.data_as(u8p), nbytes, n, cs,
                               cb.ctypes.data_as(u16p) if gap else None,
                               so.ctypes.data_as(u64p) if gap else None, back.ctypes.data_as(u8p)),
              tag + f" hf_decode_host gap={gap}")
        assert np.array_equal(back, data), tag + f" decode_host gap={gap}"
    # a window walking the bare stream
    window = int(rng.choice([1 << 16, 1 << 18, 1 << 20]))
    base, start_bit, done, pieces = 0, 0, 0, []
    while done < n:
        win = np.ascontiguousarray(stream[base: base + window])
        last = base + window >= stream.size
        n_max = min(n - done, win.size * 8 + 8)
        o = np.zeros(n_max, dtype=np.uint8)
        got, end = C.c_uint64(), C.c_uint64()
        whole = last and n_max == n - done
        check(L.hf_decode_window_host(H, model.handle, win.ctypes.data_as(u8p), win.size, start_bit, n_max,
                                      1 if whole else 0, o.ctypes.data_as(u8p), C.byref(got), C.byref(end)),
              tag + f" window {window} at byte {base}")
        if got.value == 0:
            window *= 2  # a window that holds no whole segment: the caller widens it
            assert window <= 1 << 26, tag + " window decode makes no progress"
            continue
        pieces.append(o[: got.value])
        done += got.value
        base += end.value // 8
        start_bit = end.value % 8
    assert np.array_equal(np.concatenate(pieces), data), tag + f" window decode ({window})"
    # runs of segments with the gap array
    per = int(rng.integers(1, 12))
    seg_syms = 256 * cs
    got = []
    for s0 in range(0, ns, per):
        s1 = min(ns, s0 + per)
        n_piece = min(n, s1 * seg_syms) - s0 * seg_syms
        first_byte = (int(so[s0]) // 8) & ~15
        end_bit = int(so[s1]) if s1 < ns else int(bits.value) if lead == 0 else None
        if end_bit is None:
            end_bit = 8 * nbytes
        win = np.ascontiguousarray(stream[first_byte: (end_bit + 7) // 8 + 16])
        o = np.zeros(n_piece, dtype=np.uint8)
        check(L.hf_decode_range_host(H, model.handle, win.ctypes.data_as(u8p), win.size, first_byte * 8,
                                     n_piece, cs, cb[s0 * 256:].ctypes.data_as(u16p),
                                     so[s0:].ctypes.data_as(u64p), o.ctypes.data_as(u8p)),
              tag + f" range {s0}..{s1}")
        got.append(o)
    assert np.array_equal(np.concatenate(got), data), tag + " range decode"
    return tag + " ok"


def case_shards(rng, tag):
    n = int(np.exp(rng.uniform(np.log(8), np.log(3e6))))
    data = random_data(rng, n)
    world = int(rng.integers(2, 9))
    cuts = np.sort(rng.integers(0, n + 1, size=world - 1))
    bounds = [0, *cuts.tolist(), n]
    cs = int(rng.choice([64, 128, 256]))
    tag += f" n={n} world={world} cs={cs}"
    hists = np.stack([orc.histogram(data[bounds[r]: bounds[r + 1]]) for r in range(world)]).astype(np.uint64)
    codes, lens = orc.generate_extended_codewords(hists.sum(axis=0))
    want, _ = orc.encode_slice(data, codes, lens)
    bits = np.zeros(world, np.uint64)
    start = np.zeros(world, np.uint64)
    lead = np.zeros(world, np.uint32)
    bf = np.zeros(world, np.uint64)
    bl = np.zeros(world, np.uint64)
    total = C.c_uint64()
    check(L.hf_shard_plan(np.ascontiguousarray(hists).ctypes.data_as(u64p), lens.ctypes.data_as(u64p), world,
                          bits.ctypes.data_as(u64p), start.ctypes.data_as(u64p), lead.ctypes.data_as(u32p),
                          bf.ctypes.data_as(u64p), bl.ctypes.data_as(u64p), C.byref(total)), tag + " plan")
    assert (int(total.value) + 7) // 8 == want.size, tag + " total bits"
    model = dev.DeviceModel(codes, lens)
    whole = np.zeros(want.size + 8, dtype=np.uint8)
    shards = []
    for r in range(world):
        part = torch.from_numpy(np.ascontiguousarray(data[bounds[r]: bounds[r + 1]])).cuda()
        m = part.numel()
        if m == 0:
            shards.append(None)
            assert int(bits[r]) == 0
            continue
        planned = bool(rng.integers(0, 2))
        bh = dev.new_batch_hist(m, part.device) if planned else None
        if planned:
            dev.histogram_async(part, batch_hist=bh)
        bufs = dev.EncodeBuffers(model, m, cs, lead_bits=int(lead[r]))
        dev.encode_async(part, model, bufs, lead_bits=int(lead[r]), plan=bh)
        res = bufs.result.cpu().numpy()
        assert int(res[1]) == 0 and int(res[0]) == int(bits[r]), tag + f" shard {r} result"
        assert int(bl[r] - bf[r]) == (int(lead[r]) + int(bits[r]) + 7) // 8, tag + f" shard {r} byte range"
        shards.append((part, bufs))
    for r in rng.permutation(world):
        if shards[r] is None:
            continue
        part, bufs = shards[r]
        nb = int(bl[r] - bf[r])
        local = bufs.data[:nb].cpu().numpy()
        check(L.hf_place_shard(whole.ctypes.data_as(u8p), whole.size, local.ctypes.data_as(u8p),
                               int(start[r]), int(bits[r])), tag + f" place {r}")
    assert np.array_equal(whole[: want.size], want), tag + " assembled stream"
    # every shard decodes from its own buffer (bit_base 0 skips nothing: the lead bits are
    # part of seg_off) and from the assembled stream with bit_base = 8 * byte_first
    dw = torch.from_numpy(np.concatenate([whole, np.zeros(32, np.uint8)])).cuda()
    for r in range(world):
        if shards[r] is None:
            continue
        part, bufs = shards[r]
        m = part.numel()
        out = torch.zeros(m, dtype=torch.uint8, device="cuda")
        dres = torch.zeros(4, dtype=torch.int64, device="cuda")
        nb = int(bl[r] - bf[r])
        dev.decode_async(bufs.data, nb, model, m, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
        assert int(dres[1].item()) == 0 and torch.equal(out, part), tag + f" shard {r} decode"
        first = int(bf[r]) & ~3  # hf_decode wants a 4-byte aligned buffer
        view = dw[first:]
        so = bufs.seg_off + (8 * (int(bf[r]) - first))
        out.zero_()
        dev.decode_async(view, int(whole.size) + 32 - first, model, m, cs, bufs.chunk_bits, so, out, dres)
        assert int(dres[1].item()) == 0 and torch.equal(out, part), tag + f" shard {r} decode in place"
    return tag + " ok"


if __name__ == "__main__":
    t0 = time.time()
    kinds = [case_model, case_host, case_shards]
    for s in range(seed0, seed0 + cases):
        rng = np.random.default_rng(s)
        fn = kinds[int(rng.integers(0, 3))]
        msg = fn(rng, f"seed {s} {fn.__name__}:")
        if (s - seed0) % 100 == 0:
            print(f"[{time.time() - t0:6.1f}s] {msg}", flush=True)
    CTX.set_option("host_piece_mib", 64)
    print(f"{cases} cases passed in {time.time() - t0:.1f} s")
