"""Randomised differential run of the device path against the CPU oracle.
python tests/fuzz/fuzz_gpu.py [cases] [first seed]

Every case draws an alphabet, a weight family (uniform, Zipf, geometric,
Fibonacci-like for very long codes, Dirichlet), a size, an input alignment, a
chunk size and lead bits, and checks: both counting passes against the oracle's
histogram, the codebook, the stream of the three encoder modes bit for bit,
the gap arrays (encoder's against the rebuilt one), and the decoders.
It is a search tool, not part of the test suite: the suite holds the cases it
found as fixed parametrisations."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle as orc  # noqa: E402
from huffman_amd import _lib, device as dev  # noqa: E402

orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # only inputs of 8..40 M symbols


def weights(rng, a):
    fam = rng.integers(0, 6)
    k = np.arange(1, a + 1, dtype=np.float64)
    if fam == 0:
        w = np.ones(a)
    elif fam == 1:
        w = k ** -rng.uniform(0.3, 3.0)
    elif fam == 2:
        w = rng.uniform(0.2, 0.8) ** k
    elif fam == 3:  # Fibonacci-like: code lengths grow by one per symbol
        w = 1.0 / (1.6180339887 ** np.minimum(k, rng.integers(8, 60)))
    elif fam == 4:
        w = rng.dirichlet(np.full(a, rng.uniform(0.05, 2.0))) + 1e-12
    else:  # a few heavy symbols and a flat tail
        w = np.full(a, 1.0)
        w[: max(1, a // 16)] = rng.uniform(10, 1000)
    return fam, w / w.sum()


def shifted(stream, lead, total_bits):
    bits = np.unpackbits(stream)[:total_bits]
    out = np.concatenate([np.zeros(lead, dtype=np.uint8), bits])
    return np.packbits(out)


def diagnose_decode(ctx, model, bufs, nbytes, n, cs, t):
    """What a wrong decode looks like: again as it was (is it reproducible?), with the
    other image layout, without the pair table, through the BitReader kernel; where in
    the chunks, lanes and segments the wrong symbols sit; and segments 256..511 on
    their own (as work-groups 0..255 of a launch: is it the data or the place?)."""
    seg = 256 * cs
    dres = torch.zeros(4, dtype=torch.int64, device="cuda")
    for dbg, what in ((0, "as it was"), (0, "again"), (64, "other image layout"),
                      (128, "without pair table"), (256, "BitReader kernel")):
        ctx.set_option("debug", dbg)
        o2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dev.decode_async(bufs.data, nbytes, model, n, cs, bufs.chunk_bits, bufs.seg_off, o2, dres)
        w = torch.nonzero(o2 != t).flatten()
        segs = sorted(set((w // seg).tolist()))
        print(f"  {what}: {w.numel()} wrong symbols in {len(segs)} segments {segs[:12]}", flush=True)
        if dbg == 0 and segs:
            units = torch.bincount(((w % cs) // 16).int(), minlength=cs // 16).tolist()
            lanes = torch.bincount(((w // cs) % 256).int(), minlength=256)
            firsts = {}
            for pos, c in zip(w.tolist(), (w // cs).tolist()):
                firsts.setdefault(c, pos % cs)
            fh = np.bincount(np.array(list(firsts.values())), minlength=cs)
            print(f"    per 16-byte unit of the chunk {units}; per wave {lanes.view(4, 64).sum(dim=1).tolist()}; "
                  f"first wrong symbol of a chunk: " + " ".join(f"{i}:{int(v)}" for i, v in enumerate(fh) if v),
                  flush=True)
    ctx.set_option("debug", 0)
    if n > 600 * seg:
        o3 = torch.zeros(256 * seg, dtype=torch.uint8, device="cuda")
        dev.decode_async(bufs.data, nbytes, model, 256 * seg, cs, bufs.chunk_bits[256 * 256:],
                         bufs.seg_off[256:], o3, dres)
        print(f"  segments 256..511 in a launch of their own: "
              f"{int((o3 != t[256 * seg: 512 * seg]).sum().item())} wrong symbols", flush=True)


def run_case(seed):
    rng = np.random.default_rng(seed)
    foreign = rng.integers(0, 3) == 0
    a = int(rng.choice([1, 2, 3, 5, 17, 64, 200, 256]))
    if foreign:
        a = int(rng.integers(2, 70))
    fam, w = weights(rng, a)
    n = int(np.exp(rng.uniform(0, np.log(3e6 if rng.integers(0, 50) else 4e7))))
    if BIG:
        n = int(rng.integers(8_000_000, 40_000_000))
    if rng.integers(0, 8) == 0 and not BIG:
        n = int(rng.choice([1, 63, 64, 65, 16384, 65536, 65537, 4 * 65536, 1 << 20]))
    perm = rng.permutation(256)[:a].astype(np.uint8)
    shape = int(rng.integers(0, 4))
    if shape == 0 and n > 4:
        # runs of one symbol with log-uniform lengths: the bit density changes along the
        # stream, from tiles of 1-bit codes to tiles of the longest ones
        starts = np.sort(rng.integers(0, n, size=max(1, int(n / np.exp(rng.uniform(0, 10))))))
        seg_sym = rng.choice(a, size=starts.size + 1, p=w)
        host = perm[seg_sym[np.searchsorted(starts, np.arange(n), side="right")]]
    elif shape == 1 and n > 4:
        # the second part follows the reversed weights: what was rare becomes frequent
        cut = int(rng.integers(0, n))
        host = np.concatenate([perm[rng.choice(a, size=cut, p=w)],
                               perm[rng.choice(a, size=n - cut, p=w[::-1])]])
    else:
        host = perm[rng.choice(a, size=n, p=w)]
    off = int(rng.choice([0, 0, 1, 5, 16]))
    cs = int(rng.choice([64, 64, 128, 256]))
    lead = int(rng.choice([0, 0, 0, 3, 7]))
    tag = f"seed {seed}: a={a} fam={fam} foreign={int(foreign)} n={n} off={off} cs={cs} lead={lead}"
    buf = torch.empty(n + off, dtype=torch.uint8, device="cuda")
    buf[off:] = torch.from_numpy(host).cuda()
    t = buf[off:]
    ctx = dev.Context.get(t.device)

    want_hist = orc.histogram(host)
    model_hist = want_hist
    if foreign:
        # a codebook that does not come from this data: counts that grow like
        # r^k give code lengths up to the number of symbols, and the data uses
        # the long codes as often as the short ones
        r = rng.uniform(1.3, 2.6)
        model_hist = np.zeros(256, dtype=np.uint64)
        grow = np.minimum(r ** np.arange(a, dtype=np.float64), 2.0 ** 62)
        model_hist[perm.astype(np.int64)] = np.maximum(1, grow[::-1]).astype(np.uint64)
    bh = dev.new_batch_hist(n, t.device)
    h1 = dev.histogram_async(t).cpu().numpy().astype(np.uint64)
    h2 = dev.histogram_async(t, batch_hist=bh).cpu().numpy().astype(np.uint64)
    assert np.array_equal(h1, want_hist), tag + " plain histogram"
    assert np.array_equal(h2, want_hist), tag + " batch histogram"
    try:
        ocodes, olens = orc.generate_extended_codewords(model_hist)
    except ValueError:
        # lengths of 64 and more: the reference's shifts overflow there, both sides refuse
        try:
            dev.build_codebook(model_hist)
        except dev.HfError as e:
            return tag + f" codebook refused by both ({e.status})"
        raise AssertionError(tag + " the oracle refuses this histogram, the library does not")
    codes, lens = dev.build_codebook(model_hist)
    assert np.array_equal(codes, ocodes) and np.array_equal(lens, olens), tag + " codebook"
    maxlen = int(lens.max())
    try:
        model = dev.DeviceModel(codes, lens)
    except dev.HfError as e:
        assert maxlen > 57 and e.status == _lib.HF_ERR_UNSUPPORTED, tag + f" model: {e}"
        return tag + f" maxlen={maxlen} (unsupported, as documented)"
    want, _ = orc.encode_slice(host, codes, lens)
    total_bits = int((want_hist.astype(object) * lens.astype(object)).sum())
    want = shifted(want, lead, total_bits) if lead else want
    streams = {}
    for mode in ("three-kernel", "chained", "planned"):
        ctx.set_option("enc_mode", 0 if mode == "three-kernel" else 1)
        bufs = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
        bufs.data.fill_(0xA5)
        dev.encode_async(t, model, bufs, lead_bits=lead, plan=bh if mode == "planned" else None)
        res = bufs.result.cpu().numpy()
        assert int(res[1]) == 0, tag + f" {mode}: flags {int(res[1])}"
        assert int(res[0]) == total_bits, tag + f" {mode}: total bits"
        nbytes = (lead + total_bits + 7) // 8
        got = bufs.data[:nbytes].cpu().numpy()
        assert np.array_equal(got, want), tag + f" {mode}: stream differs"
        streams[mode] = bufs
    ctx.set_option("enc_mode", 1)
    for plan in (None, bh):  # the entry point that allocates nothing, chained and planned
        b = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
        b.data.fill_(0x3C)
        dev.encode_noalloc_async(t, model, b, lead_bits=lead, plan=plan)
        res = b.result.cpu().numpy()
        assert int(res[1]) == 0 and int(res[0]) == total_bits, tag + " noalloc result"
        assert np.array_equal(b.data[: (lead + total_bits + 7) // 8].cpu().numpy(), want), tag + " noalloc stream"
    bufs = streams["planned"]
    nbytes = (lead + total_bits + 7) // 8
    nc, ns = dev.gap_layout(n, cs)
    for mode, b in streams.items():
        assert torch.equal(b.chunk_bits[:nc], bufs.chunk_bits[:nc]), tag + f" {mode}: chunk_bits"
        assert torch.equal(b.seg_off[:ns], bufs.seg_off[:ns]), tag + f" {mode}: seg_off"
    if n <= 300000 and maxlen <= 57:
        # Encoder::write's return value and counters, call by call, against the oracle's encoder
        oe = orc.Encoder(codes, lens)
        state = [0, 64, 0]
        at = 0
        while at < n:
            k = int(min(n - at, rng.integers(1, 70000)))
            want_ret = oe.write(host[at: at + k])
            ret, state = dev.encoder_write_value(t[at: at + k], codes, lens, state)
            assert ret == want_ret and state[2] == oe.writeout, tag + f" write() value at {at}: {ret} vs {want_ret}"
            at += k
    # a decode-only model built from the key table (what a file header carries)
    keys, ksyms, kl = orc.to_btreemap(codes, lens)
    tmodel = dev.DeviceModel(table=(keys, ksyms, kl, orc.sentinel(lens)))
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dres = torch.zeros(4, dtype=torch.int64, device="cuda")
    if lead == 0:
        dev.decode_async(bufs.data, (total_bits + 7) // 8, tmodel, n, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
        assert int(dres[1].item()) == 0 and torch.equal(out, t), tag + " decode with a table model"
        out.zero_()
        dev.decode_async(bufs.data, nbytes, model, n, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
        if int(dres[1].item()) != 0 or not torch.equal(out, t):
            bad = torch.nonzero(out != t).flatten()
            diagnose_decode(ctx, model, bufs, nbytes, n, cs, t)
            raise AssertionError(tag + f" decode with gap array: flags {int(dres[1].item())}, maxlen {maxlen}, "
                                 f"{bad.numel()} wrong symbols, first at {bad[:8].tolist()}, "
                                 f"lens {sorted(set(int(x) for x in lens if x))}")
        cb, so = dev.build_gap_array(bufs.data, nbytes, model, n, cs)
        assert torch.equal(cb[:nc], bufs.chunk_bits[:nc]), tag + " rebuilt chunk_bits"
        assert torch.equal(so[:ns], bufs.seg_off[:ns]), tag + " rebuilt seg_off"
        if n <= 200000:
            out2 = dev.decode_serial(bufs.data, nbytes, model, n)
            assert torch.equal(out2, t), tag + " serial decode"
    else:
        # the stream starts `lead` bits into the buffer: decode with bit_base = -lead is not
        # part of the interface, so the shifted stream is checked through its bytes only
        pass
    return tag + f" maxlen={maxlen} ok"


if __name__ == "__main__":
    t0 = time.time()
    failures = 0
    for s in range(seed0, seed0 + cases):
        try:
            msg = run_case(s)
        except AssertionError as e:
            if not BIG:
                raise
            failures += 1
            msg = "FAILED " + str(e)[:160]
            print(msg, flush=True)
        if (s - seed0) % (10 if BIG else 500) == 0:
            print(f"[{time.time() - t0:6.1f}s] {msg}", flush=True)
    print(f"{cases - failures} of {cases} cases passed in {time.time() - t0:.1f} s")
