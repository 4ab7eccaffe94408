"""GPU parity tests: every result of libhfgpu.so (through its C ABI) against the
CPU oracle on the same seeded inputs -- bit exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from huffman_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (there is no CPU fallback)")
    from huffman_amd import device

    return device


def to_gpu(a: np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def numbers_model(orc, words):
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for w in words:
        codes[w] = w
        lens[w] = orc.calculate_length(w)
    return codes, lens


def fib_model(orc, nsym=40):
    fib = [1, 1]
    while len(fib) < nsym:
        fib.append(fib[-1] + fib[-2])
    hist = np.zeros(256, dtype=np.uint64)
    hist[:nsym] = fib[::-1]
    return orc.generate_extended_codewords(hist)


SIZES = [1, 2, 63, 64, 65, 255, 4096, 16383, 16384, 16385, 40000, 3 * 16384 + 17, 1 << 20]


# ---------------------------------------------------------------- histogram

@pytest.mark.parametrize("kind", ["testfile", "zipf", "uniform", "geometric"])
@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 4097, 100003, (1 << 22) + 5])
def test_histogram_matches_oracle(dev, orc, kind, n):
    data = synth.generate_host(kind, n, seed=n + 1)
    got = dev.histogram(to_gpu(data)) if n else dev.histogram(torch.empty(0, dtype=torch.uint8, device="cuda"))
    assert np.array_equal(got, orc.histogram(data))


@pytest.mark.parametrize("offset", [1, 3, 7, 13])
def test_histogram_unaligned_input(dev, orc, offset):
    data = synth.generate_host("zipf", 70001, seed=3)
    t = to_gpu(data)
    got = dev.histogram(t[offset:])
    assert np.array_equal(got, orc.histogram(data[offset:]))


@pytest.mark.parametrize("grid", [1, 3, 7])
@pytest.mark.parametrize("n,off", [(20 * 65536 + 12345, 0), (9 * 65536, 16), (7 * 65536 + 15, 5)])
def test_histogram_few_workgroups_many_spans(dev, orc, grid, n, off):
    # a workgroup that stays for several 64 KiB spans keeps refilling its load
    # registers: with the default grid only inputs of 32 MiB and more get there
    host = synth.generate_host("zipf", n + off, seed=n + grid)
    data = to_gpu(host)[off:]
    ctx = dev.Context.get(data.device)
    ctx.set_option("hist_grid", grid)
    try:
        plain = dev.histogram_async(data)
        bh = dev.new_batch_hist(n, data.device)
        with_batches = dev.histogram_async(data, batch_hist=bh)
        torch.cuda.synchronize()
    finally:
        ctx.set_option("hist_grid", 0)
    want = orc.histogram(host[off:])
    assert np.array_equal(plain.cpu().numpy().astype(np.uint64), want)
    assert np.array_equal(with_batches.cpu().numpy().astype(np.uint64), want)
    nb = (n + 65535) // 65536
    got = bh[: nb * 256].cpu().numpy().reshape(nb, 256)
    for b in range(nb):
        assert np.array_equal(got[b], np.bincount(host[off + b * 65536: off + min(n, (b + 1) * 65536)],
                                                  minlength=256)), b


@pytest.mark.parametrize("off", [0, 16, 3])
def test_histogram_streaming_loads_large_input(dev, off):
    # inputs of 256 MiB and more are read with non-temporal loads
    n = (1 << 28) + 12345
    t = synth.generate_device("zipf", n + off, seed=7)[off:]
    want = torch.bincount(t.int(), minlength=256).cpu().numpy()
    plain = dev.histogram_async(t)
    bh = dev.new_batch_hist(n, t.device)
    with_batches = dev.histogram_async(t, batch_hist=bh)
    torch.cuda.synchronize()
    assert np.array_equal(plain.cpu().numpy(), want)
    assert np.array_equal(with_batches.cpu().numpy(), want)
    nb = (n + 65535) // 65536
    sums = bh[: nb * 256].view(nb, 256).to(torch.int64).sum(dim=0).cpu().numpy()
    assert np.array_equal(sums, want)
    last = torch.bincount(t[(nb - 1) * 65536:].int(), minlength=256).cpu().numpy()
    assert np.array_equal(bh[(nb - 1) * 256: nb * 256].cpu().numpy(), last)


def test_histogram_kat(dev, kats):
    k = kats["histogram"]
    h = dev.histogram(to_gpu(np.array(k["data"], dtype=np.uint8)))
    for sym, cnt in k["counts"].items():
        assert h[int(sym)] == cnt


# ------------------------------------------------------------------- encode

def check_encode(dev, orc, data, codes, lens, chunk_syms=64, mode=1):
    model = dev.DeviceModel(codes, lens)
    model.ctx.set_option("enc_mode", mode)
    try:
        enc = dev.encode(to_gpu(data), model, chunk_syms=chunk_syms)
    finally:
        model.ctx.set_option("enc_mode", 1)
    want, _ = orc.encode_slice(data, codes, lens)
    l = np.asarray(lens, dtype=np.uint64)[data].astype(np.uint64)
    assert enc.total_bits == int(l.sum())
    assert enc.nbytes == want.size
    got = enc.stream().cpu().numpy()
    assert np.array_equal(got, want)
    # gap array: bit length of every chunk, start bit of every 256th chunk
    nch = (data.size + chunk_syms - 1) // chunk_syms
    pad = np.zeros(nch * chunk_syms, dtype=np.uint64)
    pad[: data.size] = l
    per_chunk = pad.reshape(nch, chunk_syms).sum(axis=1)
    cb = enc.chunk_bits.cpu().numpy().view(np.uint16)[:nch].astype(np.uint64)
    assert np.array_equal(cb, per_chunk)
    starts = np.concatenate([[0], np.cumsum(per_chunk)[:-1]])
    so = enc.seg_off.cpu().numpy().view(np.uint64)[: (nch + 255) // 256]
    assert np.array_equal(so, starts[::256])
    return model, enc


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", SIZES)
def test_encode_zipf_matches_oracle(dev, orc, n, mode):
    data = synth.generate_host("zipf", n, seed=n)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    check_encode(dev, orc, data, codes, lens, mode=mode)


@pytest.mark.parametrize("kind", ["testfile", "uniform", "geometric"])
@pytest.mark.parametrize("n", [1, 777, 16384, 200001, 1 << 20])
def test_encode_other_distributions(dev, orc, kind, n):
    data = synth.generate_host(kind, n, seed=11 + n)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    check_encode(dev, orc, data, codes, lens)


@pytest.mark.parametrize("n,off,cs", [(1, 0, 64), (15, 0, 64), (16, 0, 128), (65536, 0, 64), (65537, 3, 256),
                                      (3 * 65536 + 17, 16, 64), ((1 << 22) + 5, 0, 128), ((1 << 22) + 5, 1, 64)])
def test_encode_all_codes_eight_bits_is_a_byte_map(dev, orc, n, off, cs):
    """Every symbol with an 8-bit code (incompressible data): the encoder maps
    bytes (enc_bytes_kernel); stream, gap array and totals against the oracle,
    the general encoder (lead bits force it) against the same stream shifted,
    and a buffer that is too small."""
    hist = np.full(256, 1000, dtype=np.uint64)
    hist[::7] += 1  # ties broken differently from the identity: codes are a permutation
    codes, lens = orc.generate_extended_codewords(hist)
    assert set(int(x) for x in lens) == {8}
    rng = np.random.default_rng(n)
    host = rng.integers(0, 256, size=n + off, dtype=np.uint8)
    t = to_gpu(host)[off:]
    model = dev.DeviceModel(codes, lens)
    want, _ = orc.encode_slice(host[off:], codes, lens)
    assert want.size == n
    enc = dev.encode(t, model, chunk_syms=cs)
    assert enc.total_bits == 8 * n
    assert np.array_equal(enc.stream().cpu().numpy(), want)
    nc, ns = dev.gap_layout(n, cs)
    cb = enc.chunk_bits[:nc].cpu().numpy().astype(np.uint16).astype(np.int64)
    assert int(cb.sum()) == 8 * n and (cb[:-1] == 8 * cs).all()
    assert np.array_equal(enc.seg_off[:ns].cpu().numpy(), np.arange(ns, dtype=np.int64) * 8 * 256 * cs)
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), host[off:])
    # planned entry point and the one that allocates nothing take the same path
    bh = dev.new_batch_hist(n, t.device)
    dev.histogram_async(t, batch_hist=bh)
    for fn in (dev.encode_async, dev.encode_noalloc_async):
        bufs = dev.EncodeBuffers(model, n, cs)
        bufs.data.fill_(0x77)
        fn(t, model, bufs, plan=bh)
        assert bufs.result.tolist()[:2] == [8 * n, 0]
        assert np.array_equal(bufs.data[:n].cpu().numpy(), want)
        assert torch.equal(bufs.chunk_bits[:nc], enc.chunk_bits[:nc])
    # lead bits: the general encoder, same bits three places further on
    shifted = dev.encode(t, model, chunk_syms=cs, lead_bits=3)
    bits = np.concatenate([np.zeros(3, np.uint8), np.unpackbits(want)])
    assert np.array_equal(shifted.stream().cpu().numpy(), np.packbits(bits))
    if n > 16:
        with pytest.raises(dev.HfError) as e:
            dev.encode(t, model, chunk_syms=cs, out_cap=(n - 1) & ~3)
        assert e.value.status == _lib_status("HF_ERR_CAPACITY")


def _lib_status(name):
    from huffman_amd import _lib

    return getattr(_lib, name)


@pytest.mark.parametrize("chunk_syms", [128, 256])
def test_encode_chunk_sizes(dev, orc, chunk_syms):
    data = synth.generate_host("zipf", 123457, seed=5)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    check_encode(dev, orc, data, codes, lens, chunk_syms=chunk_syms)


def test_product_codebook_equals_oracle_codebook_on_gpu_histogram(dev, orc):
    data = synth.generate_host("zipf", 1 << 20, seed=99)
    hist = dev.histogram(to_gpu(data))
    codes, lens = dev.build_codebook(hist)
    ocodes, olens = orc.generate_extended_codewords(orc.histogram(data))
    assert np.array_equal(codes, ocodes) and np.array_equal(lens, olens)


def test_encode_numbers_kat(dev, orc, kats):
    k = kats["encode_numbers"]
    codes, lens = numbers_model(orc, k["words"])
    data = np.array(k["words"], dtype=np.uint8)
    enc = dev.encode(to_gpu(data), dev.DeviceModel(codes, lens))
    assert list(enc.stream().cpu().numpy()) == k["bytes"]
    ret, state = dev.encoder_write_value(to_gpu(data), codes, lens, [0, 64, 0])
    assert ret == k["write_return"]


def test_encode_stream_kat(dev, orc, kats):
    k = kats["encode_stream"]
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for sym, code in k["codes"].items():
        codes[int(sym)] = code
        lens[int(sym)] = orc.calculate_length(code)
    data = np.array(k["symbols"], dtype=np.uint8)
    enc = dev.encode(to_gpu(data), dev.DeviceModel(codes, lens))
    assert list(enc.stream().cpu().numpy()) == k["bytes"]
    ret, _ = dev.encoder_write_value(to_gpu(data), codes, lens, [0, 64, 0])
    assert ret == k["write_return"]


def test_write_value_follows_reference_schedule_across_calls(dev, orc):
    data = synth.generate_host("testfile", 20000, seed=2)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    oenc = orc.Encoder(codes, lens)
    state = [0, 64, 0]
    t = to_gpu(data)
    for lo, hi in [(0, 1), (1, 4097), (4097, 8193), (8193, 20000)]:
        want = oenc.write(data[lo:hi])
        got, state = dev.encoder_write_value(t[lo:hi], codes, lens, state)
        assert got == want
        assert state[2] == oenc.writeout


def test_encode_long_codes(dev, orc):
    codes, lens = fib_model(orc, 40)
    assert lens.max() > 32
    rng = np.random.default_rng(8)
    data = rng.integers(0, 40, size=50000, dtype=np.uint8)
    model, enc = check_encode(dev, orc, data, codes, lens)
    out = dev.decode(enc, model)
    assert np.array_equal(out.cpu().numpy(), data)


def test_encode_merged_groups_that_overflow(dev, orc):
    # a model with code lengths 1..28: the encoder merges four codes per append
    # on the expectation that a group fits 32 bits; data that the model calls
    # improbable (runs of the rarest symbols) makes groups overflow, and those
    # lanes must fall back without damaging their neighbours' bits
    hist = np.zeros(256, dtype=np.uint64)
    for k in range(29):
        hist[k] = 1 << (28 - k)
    codes, lens = orc.generate_extended_codewords(hist)
    assert 16 < lens.max() <= 32
    rng = np.random.default_rng(12)
    data = rng.geometric(0.5, size=3 * 16384 + 333).clip(1, 29).astype(np.uint8) - 1
    data[1000:1100] = 28                       # every group overflows here
    data[16384 - 3: 16384 + 5] = 27            # across a tile edge
    data[40000:40004] = [0, 28, 0, 28]         # a single overflowing group
    data[-2:] = 28                             # at the very end
    for mode in (0, 1):
        model, enc = check_encode(dev, orc, data, codes, lens, mode=mode)
    out = dev.decode(enc, model)
    assert np.array_equal(out.cpu().numpy(), data)
    worst = np.full(70000, 28, dtype=np.uint8)  # nothing but the longest code
    model, enc = check_encode(dev, orc, worst, codes, lens)
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), worst)


def test_encode_symbols_without_code_add_no_bits(dev, orc):
    # absent symbols carry code 0 / length 0 (src/huffman/mod.rs:194-205); if
    # they occur in the data the reference appends nothing for them
    hist = np.zeros(256, dtype=np.uint64)
    hist[:4] = [5, 3, 2, 1]
    codes, lens = orc.generate_extended_codewords(hist)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 8, size=70000, dtype=np.uint8)
    data[16384 - 40: 16384 + 200] = 7   # a run of empty symbols across a tile edge
    check_encode(dev, orc, data, codes, lens)


@pytest.mark.parametrize("lead", [1, 3, 7])
def test_encode_lead_bits_shift_the_stream(dev, orc, lead):
    data = synth.generate_host("zipf", 50001, seed=lead)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, lead_bits=lead)
    want, _ = orc.encode_slice(data, codes, lens)
    bits = np.unpackbits(want)[: enc.total_bits]
    shifted = np.concatenate([np.zeros(lead, dtype=np.uint8), bits])
    assert np.array_equal(enc.stream().cpu().numpy(), np.packbits(shifted))


def test_encode_exact_capacity_and_too_small(dev, orc):
    from huffman_amd._lib import HfError, HF_ERR_CAPACITY

    data = synth.generate_host("zipf", 100000, seed=1)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, out_cap=want.size)
    assert np.array_equal(enc.stream().cpu().numpy(), want)
    with pytest.raises(HfError) as ei:
        dev.encode(to_gpu(data), model, out_cap=want.size - 1)
    assert ei.value.status == HF_ERR_CAPACITY


def test_model_validation_errors(dev):
    from huffman_amd._lib import (HfError, HF_ERR_BAD_CODEBOOK, HF_ERR_CODELEN, HF_ERR_EMPTY,
                                  HF_ERR_UNSUPPORTED)

    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    with pytest.raises(HfError) as ei:
        dev.DeviceModel(codes, lens)
    assert ei.value.status == HF_ERR_EMPTY
    lens[1] = 65
    with pytest.raises(HfError) as ei:
        dev.DeviceModel(codes, lens)
    assert ei.value.status == HF_ERR_CODELEN   # InvalidData in the reference
    lens[1] = 60
    with pytest.raises(HfError) as ei:
        dev.DeviceModel(codes, lens)
    assert ei.value.status == HF_ERR_UNSUPPORTED
    lens[1] = 3
    codes[1] = 8
    with pytest.raises(HfError) as ei:
        dev.DeviceModel(codes, lens)
    assert ei.value.status == HF_ERR_BAD_CODEBOOK


# ------------------------------------------------------------------- decode

@pytest.mark.parametrize("kind", ["testfile", "zipf", "uniform", "geometric"])
@pytest.mark.parametrize("n", [1, 63, 129, 32768, 32769, 300007, 1 << 20])
def test_decode_roundtrip_and_oracle(dev, orc, kind, n):
    data = synth.generate_host(kind, n, seed=17 + n)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    out = dev.decode(enc, model).cpu().numpy()
    assert np.array_equal(out, data)
    stream = enc.stream().cpu().numpy()
    assert np.array_equal(out, orc.decode_read(stream, codes, lens, n))


@pytest.mark.parametrize("chunk_syms", [128, 256])
def test_decode_chunk_sizes(dev, orc, chunk_syms):
    data = synth.generate_host("zipf", 200003, seed=4)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, chunk_syms=chunk_syms)
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), data)


def test_decode_numbers_not_prefix_free(dev, orc, kats):
    # predecessor-search semantics: code 1010 is a prefix of 10100101
    words = kats["decode_numbers"]["words"]
    codes, lens = numbers_model(orc, words)
    data = np.array(words, dtype=np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    stream = enc.stream().cpu().numpy()
    want = orc.decode_read(stream, codes, lens, data.size)
    assert list(want) == words
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), want)
    assert np.array_equal(dev.decode_serial(enc.data, enc.nbytes, model, data.size).cpu().numpy(),
                          want)


def test_decode_not_prefix_free_long_input_fails_like_the_reference(dev, orc, kats):
    # repeated 50 times the same model loses sync and meets a window below its
    # smallest key: the reference panics there, the library reports HF_ERR_DECODE
    from huffman_amd._lib import HfError, HF_ERR_DECODE

    words = kats["decode_numbers"]["words"]
    codes, lens = numbers_model(orc, words)
    data = np.array(words * 50, dtype=np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    with pytest.raises(ValueError):
        orc.decode_read(enc.stream().cpu().numpy(), codes, lens, data.size)
    with pytest.raises(HfError) as ei:
        dev.decode_serial(enc.data, enc.nbytes, model, data.size)
    assert ei.value.status == HF_ERR_DECODE


def test_decode_histogram_encoded_kat(dev, orc, kats):
    k = kats["decode_numbers_histogram_encoded"]
    hist = np.zeros(256, dtype=np.uint64)
    hist[: len(k["histogram"])] = k["histogram"]
    codes, lens = dev.build_codebook(hist)
    data = np.array(k["origin"], dtype=np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    assert list(dev.decode(enc, model).cpu().numpy()) == k["origin"]
    assert list(dev.decode_serial(enc.data, enc.nbytes, model, data.size).cpu().numpy()) == k["origin"]


def test_vault_sentences_roundtrip(dev, orc, kats):
    for sentence in kats["vault_roundtrips"]["sentences"]:
        data = np.frombuffer(sentence.encode(), dtype=np.uint8)
        t = to_gpu(data)
        codes, lens = dev.build_codebook(dev.histogram(t))
        model = dev.DeviceModel(codes, lens)
        enc = dev.encode(t, model)
        want, _ = orc.encode_slice(data, codes, lens)
        assert np.array_equal(enc.stream().cpu().numpy(), want)
        assert dev.decode(enc, model).cpu().numpy().tobytes() == sentence.encode()


def test_decode_from_header_table(dev, orc):
    data = synth.generate_host("zipf", 70000, seed=21)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    enc = dev.encode(to_gpu(data), dev.DeviceModel(codes, lens))
    keys, syms, kl = orc.to_btreemap(codes, lens)
    dmodel = dev.DeviceModel(table=(keys, syms, kl, orc.sentinel(lens)))
    assert np.array_equal(dev.decode(enc, dmodel).cpu().numpy(), data)


def test_decode_window_below_smallest_key_is_an_error(dev, orc):
    from huffman_amd._lib import HfError, HF_ERR_DECODE

    # a model without key 0: windows of zero bits have no predecessor
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    codes[1], lens[1] = 1, 1
    codes[2], lens[2] = 1, 2
    model = dev.DeviceModel(codes, lens)
    data = np.array([1, 2, 1, 1], dtype=np.uint8)
    enc = dev.encode(to_gpu(data), model)
    with pytest.raises(ValueError):
        orc.decode_read(enc.stream().cpu().numpy(), codes, lens, 8)
    enc.n = 8  # ask for more symbols than the stream holds: the zero padding is hit
    with pytest.raises(HfError) as ei:
        dev.decode(enc, model)
    assert ei.value.status == HF_ERR_DECODE


# ------------------------------------------------- full-size properties

@pytest.mark.parametrize("kind", ["zipf", "uniform", "geometric"])
def test_large_roundtrip_properties(dev, kind):
    n = 1 << 28
    t = synth.generate_device(kind, n, seed=42)
    hist = dev.histogram(t)
    assert int(hist.sum()) == n
    ref = torch.bincount(t[: 1 << 24].to(torch.int64), minlength=256).cpu().numpy()
    assert np.array_equal(dev.histogram(t[: 1 << 24]), ref.astype(np.uint64))
    codes, lens = dev.build_codebook(hist)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(t, model)
    assert enc.total_bits == int((hist * lens).sum())
    # both encode schedules must give the same bytes
    model.ctx.set_option("enc_mode", 0)
    try:
        enc0 = dev.encode(t, model)
    finally:
        model.ctx.set_option("enc_mode", 1)
    assert enc0.total_bits == enc.total_bits
    assert torch.equal(enc0.stream(), enc.stream())
    assert torch.equal(enc0.chunk_bits, enc.chunk_bits) and torch.equal(enc0.seg_off, enc.seg_off)
    out = dev.decode(enc, model)
    assert torch.equal(out, t)


# ------------------------------------------- gap array rebuilt from the stream

@pytest.mark.parametrize("kind", ["zipf", "testfile", "geometric", "uniform"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4096, 16385, 100003, (1 << 21) + 77])
def test_build_gap_array_matches_the_encoders(dev, orc, kind, n):
    data = synth.generate_host(kind, n, seed=n + 3)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n)
    nc, ns = dev.gap_layout(n, 64)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc])
    assert torch.equal(so[:ns], enc.seg_off[:ns])
    bare = dev.Encoded(enc.data, enc.total_bits, 0, n, 64, cb, so)
    assert np.array_equal(dev.decode(bare, model).cpu().numpy(), data)
    # the same walks with the symbols written in the last one: no gap array at all
    assert np.array_equal(dev.decode_bare(enc.data, enc.nbytes, model, n).cpu().numpy(), data)


@pytest.mark.parametrize("chunk_syms", [128, 256])
def test_build_gap_array_chunk_sizes(dev, orc, chunk_syms):
    data = synth.generate_host("zipf", 300001, seed=5)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, chunk_syms=chunk_syms)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, data.size, chunk_syms)
    nc, ns = dev.gap_layout(data.size, chunk_syms)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc])
    assert torch.equal(so[:ns], enc.seg_off[:ns])


def test_build_gap_array_model_that_is_not_prefix_free(dev, orc):
    # overlapping codes, as in decode_numbers (src/huffman/decode/mod.rs:240-258),
    # but with a key 0 so that every bit string decodes: '0', '1', '11', '101'.
    # The code boundaries are those of the sequential predecessor-search decode,
    # not those of the symbols that were encoded.
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for sym, (c, l) in enumerate([(0b0, 1), (0b1, 1), (0b11, 2), (0b101, 3)]):
        codes[sym], lens[sym] = c, l
    rng = np.random.default_rng(4)
    data = rng.integers(0, 4, size=90000, dtype=np.uint8)
    stream, _ = orc.encode_slice(data, codes, lens)
    n_out = data.size // 3          # cannot run past the end of the stream
    # the reference's slice decoder only honours the count in its drain loop
    # (src/huffman/decode/mod.rs:187-229): it returns every symbol of the stream
    want = orc.decode_read(stream, codes, lens, n_out)[:n_out]
    assert not np.array_equal(want, data[:n_out])   # the model really is ambiguous
    model = dev.DeviceModel(codes, lens)
    t = to_gpu(np.concatenate([stream, np.zeros(16, dtype=np.uint8)]))
    cb, so = dev.build_gap_array(t, stream.size, model, n_out)
    bare = dev.Encoded(t, stream.size * 8, 0, n_out, 64, cb, so)
    assert np.array_equal(dev.decode(bare, model).cpu().numpy(), want)
    assert np.array_equal(dev.decode_serial(t, stream.size, model, n_out).cpu().numpy(), want)
    assert np.array_equal(dev.decode_bare(t, stream.size, model, n_out).cpu().numpy(), want)


def test_build_gap_array_long_codes(dev, orc):
    codes, lens = fib_model(orc, 40)
    rng = np.random.default_rng(18)
    data = rng.integers(0, 40, size=70001, dtype=np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, data.size)
    nc, ns = dev.gap_layout(data.size, 64)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc])
    assert torch.equal(so[:ns], enc.seg_off[:ns])
    assert np.array_equal(dev.decode_bare(enc.data, enc.nbytes, model, data.size).cpu().numpy(), data)


def test_build_gap_array_stream_too_short(dev, orc):
    data = synth.generate_host("zipf", 50000, seed=1)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    with pytest.raises(dev.HfError) as ei:
        dev.build_gap_array(enc.data, enc.nbytes // 2, model, data.size)
    assert ei.value.status == dev._lib.HF_ERR_DECODE
    with pytest.raises(dev.HfError) as ei:
        dev.decode_bare(enc.data, enc.nbytes // 2, model, data.size)
    assert ei.value.status == dev._lib.HF_ERR_DECODE


def test_build_gap_array_256mib_property(dev):
    n = 1 << 28
    data = synth.generate_device("zipf", n, seed=77)
    codes, lens = dev.build_codebook(dev.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(data, model)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n)
    nc, ns = dev.gap_layout(n, 64)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc])
    assert torch.equal(so[:ns], enc.seg_off[:ns])
    assert torch.equal(dev.decode_bare(enc.data, enc.nbytes, model, n), data)


@pytest.mark.parametrize("kind,n", [("geometric", (1 << 28) + 12345), ("uniform", 1 << 27),
                                    ("testfile", (1 << 27) + 1)])
def test_decode_bare_large_streams(dev, kind, n):
    # many work-groups: every one assumes its first start from a walk of its
    # predecessor's last sub-sequence, and the check behind the launch has to agree
    data = synth.generate_device(kind, n, seed=78)
    codes, lens = dev.build_codebook(dev.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(data, model)
    out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    dev.decode_bare(enc.data, enc.nbytes, model, n, out[3: 3 + n])  # (output at an odd address)
    assert torch.equal(out[3: 3 + n], data)
    assert int(out[:3].sum()) == 0 and int(out[3 + n:].sum()) == 0


def test_decode_bare_stream_that_resynchronises_late(dev, orc):
    """A fixed-length code never resynchronises: a walk that starts at a wrong
    offset stays wrong to the end of its sub-sequence.  13-bit codes do not divide
    the 1024-bit sub-sequences, so the assumed starts (offset 0, and the probes at
    the sub-sequences' ends) are wrong nearly everywhere, and only the repeated
    rounds -- inside the work-groups and across them -- reach the sequential decode."""
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    rng = np.random.default_rng(5)
    syms = rng.permutation(256)[:200]
    for i, s in enumerate(syms):
        codes[s], lens[s] = i * 40, 13
    n = 600001
    data = syms[rng.integers(0, len(syms), size=n)].astype(np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model)
    want, _ = orc.encode_slice(data, codes, lens)
    assert np.array_equal(enc.stream().cpu().numpy(), want)
    assert np.array_equal(dev.decode_bare(enc.data, enc.nbytes, model, n).cpu().numpy(), data)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n)
    nc, ns = dev.gap_layout(n, 64)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc]) and torch.equal(so[:ns], enc.seg_off[:ns])


# ----------------------------------- the benchmark configurations, full size

def _oracle_windows(orc, data, enc, codes, lens, win=32 << 20):
    """The first and the last `win` symbols of a full-size stream against the
    oracle's stream of those symbols under the same codebook.  The last window
    starts at a segment boundary, whose bit offset the gap array gives: the
    oracle's bits must appear in the device stream from that bit on, and the
    stream must end with them."""
    n = data.numel()
    l64 = np.asarray(lens, dtype=np.uint64)
    head = data[:win].cpu().numpy()
    want, _ = orc.encode_slice(head, codes, lens)
    nbits = int(l64[head].sum())
    got = enc.data[: want.size].cpu().numpy()
    assert np.array_equal(got[: nbits // 8], want[: nbits // 8])
    if nbits % 8:
        mask = (0xff << (8 - nbits % 8)) & 0xff
        assert (got[nbits // 8] & mask) == (want[nbits // 8] & mask)
    seg_syms = 64 * 256
    s0 = (n - win) // seg_syms * seg_syms
    tail = data[s0:].cpu().numpy()
    want, _ = orc.encode_slice(tail, codes, lens)
    nbits = int(l64[tail].sum())
    off = int(enc.seg_off[s0 // seg_syms].item())
    assert off + nbits == enc.total_bits
    got = enc.data[off // 8: enc.nbytes].cpu().numpy()
    bits = np.unpackbits(got)[off % 8: off % 8 + nbits]
    assert np.array_equal(bits, np.unpackbits(want)[:nbits])


@pytest.mark.parametrize("kind,gib", [("zipf", 1), ("uniform", 1), ("geometric", 4), ("zipf", 16)])
def test_full_size_round_trip_properties(dev, orc, kind, gib):
    # size-independent properties at the sizes the benchmark configurations name:
    # the stream length is sum(hist * len), the gap array sums to it, the
    # decode returns the input, and a second encode gives the same bytes
    n = gib << 30
    data = synth.generate_device(kind, n, seed=31)
    hist = dev.histogram(data)
    assert int(hist.sum()) == n
    codes, lens = dev.build_codebook(hist)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(data, model)
    assert enc.total_bits == int((hist * lens).sum())
    nc, ns = dev.gap_layout(n, 64)
    cb = enc.chunk_bits[:nc].to(torch.int32) & 0xffff
    assert int(cb.sum(dtype=torch.int64)) == enc.total_bits
    seg = cb.view(-1, 256).sum(dim=1, dtype=torch.int64).cumsum(0)
    assert torch.equal(enc.seg_off[1:ns], seg[: ns - 1])
    out = dev.decode(enc, model)
    assert torch.equal(out, data)
    del out
    # the configuration's own codebook (11-bit Zipf, 30-bit geometric codes) at
    # size: both ends of the stream bit for bit against the oracle
    _oracle_windows(orc, data, enc, codes, lens)
    if gib >= 16:
        assert enc.total_bits > 1 << 36 and int(enc.seg_off[ns - 1].item()) > 1 << 36
    # ... the planned encoder (batch histograms of the counting pass) included;
    # at 4 GiB its start bits exceed 32 bits
    again = dev.encode(data, model)
    assert torch.equal(again.stream(), enc.stream())
    del again
    bh = dev.new_batch_hist(n, data.device)
    d_hist = dev.histogram_async(data, batch_hist=bh)
    assert np.array_equal(d_hist.cpu().numpy().astype(np.uint64), hist)
    assert int(bh[: (n >> 16) * 256].sum(dtype=torch.int64)) == n
    enc.data[: enc.nbytes].zero_()  # same buffers, so that 4 GiB cases fit: clear, re-encode, decode
    bufs = dev.EncodeBuffers.__new__(dev.EncodeBuffers)
    bufs.n, bufs.chunk_syms, bufs.cap = n, 64, enc.data.numel()
    bufs.data, bufs.chunk_bits, bufs.seg_off = enc.data, enc.chunk_bits, enc.seg_off
    bufs.result = torch.zeros(4, dtype=torch.int64, device=data.device)
    keep_cb = enc.chunk_bits.clone()
    dev.encode_async(data, model, bufs, plan=bh)
    torch.cuda.synchronize()
    assert bufs.result.tolist()[:2] == [enc.total_bits, 0]
    assert torch.equal(keep_cb, enc.chunk_bits)
    out = dev.decode(enc, model)
    assert torch.equal(out, data)


# ------------------------------------------------------------ random models

def _random_histogram(rng):
    kind = rng.integers(0, 6)
    nsym = int(rng.integers(1, 257))
    syms = rng.permutation(256)[:nsym]
    hist = np.zeros(256, dtype=np.uint64)
    if kind == 0:      # flat
        hist[syms] = rng.integers(1, 4, size=nsym)
    elif kind == 1:    # geometric: long codes
        hist[syms] = np.maximum(1, (2.0 ** (40 - 1.0 * np.arange(nsym))).astype(np.uint64))
    elif kind == 2:    # Fibonacci: the longest codes a histogram can give
        fib = [1, 1]
        while len(fib) < min(nsym, 50):
            fib.append(fib[-1] + fib[-2])
        fib = fib[: nsym]
        hist[syms[: len(fib)]] = fib[::-1]
    elif kind == 3:    # Zipf with a random exponent
        hist[syms] = np.maximum(1, (1e9 / (np.arange(nsym) + 1.0) ** rng.uniform(0.5, 2.5))).astype(np.uint64)
    elif kind == 4:    # two levels
        hist[syms] = rng.choice([1, 1000], size=nsym)
    else:              # random counts
        hist[syms] = rng.integers(1, 1 << 20, size=nsym)
    return hist, syms


@pytest.mark.parametrize("seed", range(60))
def test_random_models_and_inputs(dev, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    hist, syms = _random_histogram(rng)
    codes, lens = orc.generate_extended_codewords(hist)
    if lens.max() > 57:
        pytest.skip("longer than the codec takes")
    n = int(rng.choice([1, 63, 64, 65, 1000, 16384, 16385, 50001, 150000]))
    # symbols drawn by the model's weights, with runs of the rarest ones mixed in
    p = hist[syms].astype(np.float64)
    data = rng.choice(syms, size=n, p=p / p.sum()).astype(np.uint8)
    live = syms[hist[syms] > 0]
    rare = live[np.argsort(hist[live])[: 3]]
    for _ in range(int(rng.integers(0, 4))):
        at = int(rng.integers(0, n))
        data[at: at + int(rng.integers(1, 200))] = rng.choice(rare)
    cs = int(rng.choice([64, 64, 128, 256]))
    model, enc = check_encode(dev, orc, data, codes, lens, chunk_syms=cs, mode=int(rng.integers(0, 2)))
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), data)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n, cs)
    nc, ns = dev.gap_layout(n, cs)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc])
    assert torch.equal(so[:ns], enc.seg_off[:ns])


# ------------------------------------------------- buffers that are not 16-byte aligned

@pytest.mark.parametrize("in_off,out_off", [(1, 0), (3, 4), (13, 8), (0, 12), (16, 4)])
def test_unaligned_buffers(dev, orc, in_off, out_off):
    n = 3 * 16384 * 4 + 777
    data = synth.generate_host("zipf", n, seed=in_off * 31 + out_off)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    model = dev.DeviceModel(codes, lens)
    src = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    src[in_off: in_off + n] = to_gpu(data)
    bufs = dev.EncodeBuffers(model, n, 64)
    big = torch.zeros(bufs.cap + 64, dtype=torch.uint8, device="cuda")
    bufs.data = big[out_off: out_off + bufs.cap]          # 4-byte aligned, mostly not 16
    dev.encode_async(src[in_off: in_off + n], model, bufs)
    torch.cuda.synchronize()
    assert int(bufs.result[1].item()) == 0
    bits = int(bufs.result[0].item())
    nbytes = (bits + 7) // 8
    assert nbytes == want.size
    assert np.array_equal(bufs.data[:nbytes].cpu().numpy(), want)
    # decode from a stream at a 4-byte offset into an output at an odd offset
    comp = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda")
    comp[out_off: out_off + nbytes] = bufs.data[:nbytes]
    out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    res = torch.zeros(4, dtype=torch.int64, device="cuda")
    dev.decode_async(comp[out_off:], nbytes, model, n, 64, bufs.chunk_bits, bufs.seg_off,
                     out[in_off: in_off + n], res)
    torch.cuda.synchronize()
    assert int(res[1].item()) == 0
    assert np.array_equal(out[in_off: in_off + n].cpu().numpy(), data)
    cb, so = dev.build_gap_array(comp[out_off:], nbytes, model, n)
    nc, ns = dev.gap_layout(n, 64)
    assert torch.equal(cb[:nc], bufs.chunk_bits[:nc]) and torch.equal(so[:ns], bufs.seg_off[:ns])


def test_models_created_and_used_on_different_streams(dev, orc):
    """Tables are uploaded asynchronously on the stream current at creation; a
    kernel on another stream has to wait for that upload, and a destroyed
    model's device block is reused by the next model without stalling."""
    import torch

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(12):
        kind = ["zipf", "testfile", "geometric", "uniform"][rnd % 4]
        data = synth.generate_host(kind, 200003 + 977 * rnd, seed=100 + rnd)
        codes, lens = orc.generate_extended_codewords(orc.histogram(data))
        want, _ = orc.encode_slice(data, codes, lens)
        make, use = (s1, s2) if rnd % 2 == 0 else (s2, s1)
        with torch.cuda.stream(make):
            gpu_data = to_gpu(data)
            model = dev.DeviceModel(codes, lens)  # replaces (and destroys) the previous one
        use.wait_stream(make)  # for the input tensor; the tables are the library's business
        with torch.cuda.stream(use if rnd % 3 else make):
            enc = dev.encode(gpu_data, model)
            out = dev.decode(enc, model)
            got = enc.stream().cpu().numpy()
            back = out.cpu().numpy()
        assert np.array_equal(got, want)
        assert np.array_equal(back, data)


@pytest.mark.parametrize("chunk_syms", [64, 128, 256])
@pytest.mark.parametrize("shape", ["runs", "iid"])
def test_decode_many_segments_per_cu(dev, orc, chunk_syms, shape):
    """More segments than CUs, so that several work-groups share a CU: the
    128-symbol decoder once went wrong only in work-groups that were not the
    first on their CU (from segment 256 on), on any data.  Runs of one symbol
    make every chunk the same size, the other extreme for the image layout."""
    rng = np.random.default_rng(chunk_syms)
    n = 700 * 256 * chunk_syms + 12345 if chunk_syms < 256 else 400 * 256 * chunk_syms + 77
    alphabet = np.array([86, 111, 214], dtype=np.uint8)
    if shape == "runs":
        starts = np.sort(rng.integers(0, n, size=n // 40))
        sym = rng.integers(0, 3, size=starts.size + 1)
        host = alphabet[sym[np.searchsorted(starts, np.arange(n), side="right")]]
    else:
        host = alphabet[rng.integers(0, 3, size=n)]
    t = to_gpu(host)
    codes, lens = dev.build_codebook(dev.histogram(t))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(t, model, chunk_syms=chunk_syms)
    for _ in range(3):
        got = dev.decode(enc, model)
        wrong = torch.nonzero(got != t).flatten()
        assert wrong.numel() == 0, (f"{wrong.numel()} wrong symbols, first in segment "
                                    f"{int(wrong[0]) // (256 * chunk_syms)}")
    sample = slice(0, 1 << 20)
    want, _ = orc.encode_slice(host[sample], codes, lens)
    assert np.array_equal(enc.stream()[: want.size - 1].cpu().numpy(), want[:-1])


# ---- counting pass that records a histogram per 65536 symbols, and the
# ---- encoder that takes every batch's place in the stream from them

@pytest.mark.parametrize("n,off", [(1, 0), (65535, 0), (65536, 0), (65537, 3), (300001, 0),
                                   (5 * 65536, 16), ((1 << 22) + 77, 1)])
def test_batch_histograms_match_numpy(dev, orc, n, off):
    import torch

    host = synth.generate_host("zipf", n + off, seed=n)
    data = to_gpu(host)[off:]
    bh = dev.new_batch_hist(n, data.device)
    d_hist = dev.histogram_async(data, batch_hist=bh)
    torch.cuda.synchronize()
    assert np.array_equal(d_hist.cpu().numpy().astype(np.uint64), orc.histogram(host[off:]))
    nb = (n + 65535) // 65536
    got = bh[: nb * 256].cpu().numpy().reshape(nb, 256)
    want = np.stack([np.bincount(host[off + b * 65536: off + min(n, (b + 1) * 65536)], minlength=256)
                     for b in range(nb)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["zipf", "testfile", "uniform", "geometric"])
@pytest.mark.parametrize("n,off,lead,cs", [(1, 0, 0, 64), (65536, 0, 0, 64), (65537, 0, 3, 128),
                                           (200003, 5, 7, 64), ((1 << 21) + 9, 0, 0, 256),
                                           (3 * 65536, 16, 1, 64)])
def test_planned_encode_matches_oracle(dev, orc, kind, n, off, lead, cs):
    import torch

    host = synth.generate_host(kind, n + off, seed=n + lead)
    data = to_gpu(host)[off:]
    codes, lens = orc.generate_extended_codewords(orc.histogram(host[off:]))
    want, _ = orc.encode_slice(host[off:], codes, lens)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    planned = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
    chained = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
    dev.encode_async(data, model, planned, lead_bits=lead, plan=bh)
    dev.encode_async(data, model, chained, lead_bits=lead)
    torch.cuda.synchronize()
    total = int(np.asarray(lens, dtype=np.uint64)[host[off:]].sum())
    assert planned.result.tolist() == [total, 0, 0, 0]
    nbytes = (lead + total + 7) // 8
    assert torch.equal(planned.data[:nbytes], chained.data[:nbytes])
    assert torch.equal(planned.chunk_bits, chained.chunk_bits)
    assert torch.equal(planned.seg_off, chained.seg_off)
    if lead == 0:
        assert np.array_equal(planned.data[:nbytes].cpu().numpy(), want)


def test_planned_encode_detects_histograms_of_other_data(dev, orc):
    import torch

    n = 400000
    a = synth.generate_host("zipf", n, seed=1)
    b = synth.generate_host("zipf", n, seed=2)
    codes, lens = orc.generate_extended_codewords(orc.histogram(a))
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, "cuda")
    dev.histogram_async(to_gpu(b), batch_hist=bh)  # counted on b ...
    bufs = dev.EncodeBuffers(model, n, 64)
    dev.encode_async(to_gpu(a), model, bufs, plan=bh)  # ... used for a
    torch.cuda.synchronize()
    assert int(bufs.result[1].item()) & 8  # HF_FLAG_PLAN


@pytest.mark.parametrize("kind,cs", [("zipf", 64), ("geometric", 64), ("uniform", 128), ("testfile", 256)])
def test_decode_a_byte_range_of_the_stream_with_bit_base(dev, orc, kind, cs):
    """Multi-GPU decode of one stream: a rank holds only the bytes of its range of
    segments (copied to a buffer of its own), the gap-array slices keep their
    absolute bit offsets, and bit_base tells where the buffer starts."""
    import torch

    n = 3_000_017
    data = synth.generate_host(kind, n, seed=cs + 1)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, chunk_syms=cs)
    seg_syms = 256 * cs
    n_segs = (n + seg_syms - 1) // seg_syms
    so = enc.seg_off.cpu().numpy().view(np.uint64)
    stream = enc.stream()
    world = 3
    bounds = [n_segs * r // world for r in range(world + 1)]
    for r in range(world):
        s0, s1 = bounds[r], bounds[r + 1]
        if s1 == s0:
            continue
        first = (int(so[s0]) // 8) & ~3                       # 4-byte aligned start of the range
        last = enc.nbytes if s1 == n_segs else min(enc.nbytes, int(so[s1]) // 8 + 1)
        mine = stream[first:last].clone()                      # what this rank holds
        sym0, sym1 = s0 * seg_syms, min(n, s1 * seg_syms)
        out = torch.empty(sym1 - sym0, dtype=torch.uint8, device=mine.device)
        res = torch.zeros(4, dtype=torch.int64, device=mine.device)
        dev.decode_async(mine, last - first, model, sym1 - sym0, cs,
                         enc.chunk_bits[s0 * 256:], enc.seg_off[s0:], out, res, bit_base=8 * first)
        torch.cuda.synchronize()
        assert res.tolist()[1] == 0
        assert np.array_equal(out.cpu().numpy(), data[sym0:sym1]), f"rank {r}"


@pytest.mark.parametrize("kind", ["zipf", "geometric"])
def test_shards_with_lead_bits_concatenate_to_the_whole_stream(dev, orc, kind):
    """Multi-GPU encode in one process: per-shard batch histograms give the plan
    (as bench.py derives it from one all-gather), every shard is encoded with the
    planned encoder and its lead bits, and the shards OR-merged at their byte
    ranges are the stream of the whole input."""
    import torch

    from huffman_amd.dist import plan_from_histograms

    world = 3
    n_each = [700_001, 65536 * 5, 412_345]
    parts = [synth.generate_host(kind, m, seed=40 + r) for r, m in enumerate(n_each)]
    whole = np.concatenate(parts)
    hists = np.stack([orc.histogram(p) for p in parts])
    codes, lens = orc.generate_extended_codewords(hists.sum(axis=0).astype(np.uint64))
    want, _ = orc.encode_slice(whole, codes, lens)
    model = dev.DeviceModel(codes, lens)
    merged = np.zeros(want.size, dtype=np.uint8)
    for r in range(world):
        plan = plan_from_histograms(hists, lens, r)
        t = to_gpu(parts[r])
        bh = dev.new_batch_hist(t.numel(), t.device)
        dev.histogram_async(t, batch_hist=bh)
        bufs = dev.EncodeBuffers(model, t.numel(), 64, lead_bits=plan.lead_bits)
        dev.encode_async(t, model, bufs, lead_bits=plan.lead_bits, plan=bh)
        torch.cuda.synchronize()
        assert bufs.result.tolist()[:2] == [plan.bits[r], 0]
        first, last = plan.byte_range(r)
        merged[first:last] |= bufs.data[: last - first].cpu().numpy()
    assert plan.total_bytes == want.size
    assert np.array_equal(merged, want)


def _alphabet_data(rng, n, symbols, skew):
    """n bytes over `symbols` values, probabilities proportional to skew^-rank."""
    w = skew ** -np.arange(symbols, dtype=np.float64)
    return rng.choice(symbols, size=n, p=w / w.sum()).astype(np.uint8)


@pytest.mark.parametrize("symbols,skew", [(2, 1.0), (16, 1.0), (64, 1.0), (128, 1.0), (256, 1.004),
                                          (256, 1.01), (200, 1.02), (100, 1.05)])
def test_planned_encode_of_complete_batches_by_code_length(dev, orc, symbols, skew):
    """The planned encoder takes complete batches of models whose codes have at
    most 11 bits through its one-lookup kernel, in three forms by the longest
    code (four codes per group in one register or in two, two codes per group):
    every form against the chained encoder and the oracle, with a ragged batch
    behind the complete ones."""
    import torch

    rng = np.random.default_rng(symbols * 7 + int(skew * 1000))
    n = 5 * 65536 + 777
    host = _alphabet_data(rng, n, symbols, skew)
    codes, lens = orc.generate_extended_codewords(orc.histogram(host))
    assert int(np.max(lens)) <= 11
    want, _ = orc.encode_slice(host, codes, lens)
    data = to_gpu(host)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    planned = dev.EncodeBuffers(model, n, 64)
    chained = dev.EncodeBuffers(model, n, 64)
    dev.encode_async(data, model, planned, plan=bh)
    dev.encode_async(data, model, chained)
    torch.cuda.synchronize()
    total = int(np.asarray(lens, dtype=np.uint64)[host].sum())
    assert planned.result.tolist() == [total, 0, 0, 0]
    nbytes = (total + 7) // 8
    assert np.array_equal(planned.data[:nbytes].cpu().numpy(), want)
    assert torch.equal(planned.chunk_bits, chained.chunk_bits)
    assert torch.equal(planned.seg_off, chained.seg_off)


@pytest.mark.parametrize("cs", [64, 256])
def test_planned_encode_many_batches_per_workgroup_and_longest_codes(dev, orc, cs):
    """More complete batches than resident work-groups (a work-group then takes
    a run of batches), on data that uses a Zipf codebook's 11-bit codes far more
    often than the codebook expects: stretches of the rarest symbols fill a
    tile's image to its last word."""
    import torch

    n = 1100 * 65536 + 12345
    host = synth.generate_host("zipf", n, seed=77)
    codes, lens = orc.generate_extended_codewords(orc.histogram(host))
    longest = int(np.argmax(lens))
    assert int(lens[longest]) == 11
    host = host.copy()
    host[3 * 16384: 5 * 16384] = longest          # two whole tiles of 11-bit codes
    host[700 * 65536 + 100: 701 * 65536 + 9000] = longest
    want, _ = orc.encode_slice(host, codes, lens)
    data = to_gpu(host)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    planned = dev.EncodeBuffers(model, n, cs)
    chained = dev.EncodeBuffers(model, n, cs)
    dev.encode_async(data, model, planned, plan=bh)
    dev.encode_async(data, model, chained)
    torch.cuda.synchronize()
    total = int(np.asarray(lens, dtype=np.uint64)[host].sum())
    assert planned.result.tolist() == [total, 0, 0, 0]
    nbytes = (total + 7) // 8
    assert np.array_equal(planned.data[:nbytes].cpu().numpy(), want)
    assert torch.equal(planned.chunk_bits, chained.chunk_bits)
    assert torch.equal(planned.seg_off, chained.seg_off)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    dres = torch.zeros(4, dtype=torch.int64, device="cuda")
    dev.decode_async(planned.data, nbytes, model, n, cs, planned.chunk_bits, planned.seg_off, out, dres)
    torch.cuda.synchronize()
    assert int(dres[1].item()) == 0 and torch.equal(out, data)


@pytest.mark.parametrize("cs", [64, 128])
def test_planned_encode_long_codes_with_tiles_left_for_the_second_kernel(dev, orc, cs):
    """A geometric codebook (codes up to about 20 bits at this size, groups of
    four merged speculatively) on data that breaks its assumptions: stretches of
    rare symbols make lanes' groups exceed 31 bits, whole tiles of a long code
    exceed the image, and the one-lookup kernel must hand exactly those tiles to
    the kernel that follows it.  Stream, gap array and totals against the oracle
    and the chained encoder."""
    import torch

    n = 9 * 65536 + 4321
    host = synth.generate_host("geometric", n, seed=5).copy()
    codes, lens = orc.generate_extended_codewords(orc.histogram(host))
    lens_a = np.asarray(lens)
    assert int(lens_a.max()) > 11
    rare = int(np.argmax(lens_a))                       # the longest code
    mid = int(np.flatnonzero(lens_a >= 9)[0])           # four of these exceed 31 bits
    host[1000:1040] = rare                              # a few lanes of one tile
    host[70000:70007] = mid
    host[2 * 65536 + 16384: 2 * 65536 + 3 * 16384] = rare   # two whole tiles: image overflow
    host[5 * 65536 - 50: 5 * 65536 + 50] = rare         # across a batch boundary
    host[8 * 65536 + 65000: 9 * 65536 + 100] = mid      # into the ragged batch
    want, _ = orc.encode_slice(host, codes, lens)
    data = to_gpu(host)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    planned = dev.EncodeBuffers(model, n, cs)
    chained = dev.EncodeBuffers(model, n, cs)
    for _ in range(2):  # the second call finds the list of the first one in the scratch block
        dev.encode_async(data, model, planned, plan=bh)
    assert model.ctx.encode_kernel() == "enc_pack_stream_kernel"
    dev.encode_async(data, model, chained)
    assert model.ctx.encode_kernel() == "enc_pack_kernel"
    torch.cuda.synchronize()
    total = int(lens_a.astype(np.uint64)[host].sum())
    assert planned.result.tolist() == [total, 0, 0, 0]
    nbytes = (total + 7) // 8
    assert np.array_equal(planned.data[:nbytes].cpu().numpy(), want)
    assert torch.equal(planned.chunk_bits, chained.chunk_bits)
    assert torch.equal(planned.seg_off, chained.seg_off)


@pytest.mark.parametrize("kind", ["zipf", "geometric"])
def test_planned_encode_respects_the_output_capacity(dev, orc, kind):
    """The planned encoder's kernels for complete batches against a buffer that
    is exactly large enough, one byte short, and cut at an odd place in the
    middle: nothing is written at or behind the capacity, what lies before it is
    the stream, and the flags say that bytes were dropped."""
    import torch

    n = 6 * 65536 + 99
    host = synth.generate_host(kind, n, seed=21)
    codes, lens = orc.generate_extended_codewords(orc.histogram(host))
    want, _ = orc.encode_slice(host, codes, lens)
    data = to_gpu(host)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    total = int(np.asarray(lens, dtype=np.uint64)[host].sum())
    for cap in (want.size, want.size - 1, want.size // 2 + 3, 17):
        bufs = dev.EncodeBuffers(model, n, 64, out_cap=want.size + 4096)
        bufs.data.fill_(0xAB)
        bufs.cap = cap
        dev.encode_async(data, model, bufs, plan=bh)
        torch.cuda.synchronize()
        got = bufs.data.cpu().numpy()
        assert np.array_equal(got[:cap], want[:cap]), cap
        assert np.all(got[cap:] == 0xAB), cap
        bits, flags = bufs.result.tolist()[:2]
        assert bits == total
        assert flags == (0 if cap == want.size else 1), cap  # HF_FLAG_CAPACITY


@pytest.mark.parametrize("symbols,skew", [(256, 1.02), (256, 1.03), (256, 1.04), (100, 1.15)])
def test_planned_encode_long_codes_merged_two_at_a_time(dev, orc, symbols, skew):
    """Models whose codes reach 13 to 19 bits and whose merged pairs stay below 32
    bits with near certainty: the speculative one-lookup kernel with two codes
    per group.  Pairs of the longest codes (which the data has, and which a few
    inserted stretches make frequent) go to the second kernel."""
    import torch

    rng = np.random.default_rng(5)
    n = 9 * 65536 + 4321
    host = _alphabet_data(rng, n, symbols, skew)
    codes, lens = orc.generate_extended_codewords(orc.histogram(host))
    lens_a = np.asarray(lens)
    assert 11 < int(lens_a.max()) <= 32
    rare = int(np.argmax(lens_a))
    host[4000:4100] = rare
    host[3 * 65536 - 7: 3 * 65536 + 9] = rare
    want, _ = orc.encode_slice(host, codes, lens)
    data = to_gpu(host)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    planned = dev.EncodeBuffers(model, n, 64)
    chained = dev.EncodeBuffers(model, n, 64)
    dev.encode_async(data, model, planned, plan=bh)
    assert model.ctx.encode_kernel() == "enc_pack_stream_kernel"
    dev.encode_async(data, model, chained)
    torch.cuda.synchronize()
    total = int(lens_a.astype(np.uint64)[host].sum())
    assert planned.result.tolist() == [total, 0, 0, 0]
    nbytes = (total + 7) // 8
    assert np.array_equal(planned.data[:nbytes].cpu().numpy(), want)
    assert torch.equal(planned.chunk_bits, chained.chunk_bits)
    assert torch.equal(planned.seg_off, chained.seg_off)
