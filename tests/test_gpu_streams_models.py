"""GPU tests of the library's own bookkeeping: calls in flight on several
streams that share the context's scratch block, models whose tables select the
decoder's rarely used instantiations, contexts on a device that is not the
thread's current one.  All results are compared bit-exactly with the CPU
oracle, through the C ABI."""
import ctypes as C

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


def test_encodes_in_flight_on_two_streams_share_the_scratch_block(dev, orc):
    """Two chained and two planned encodes are queued back to back on two
    streams without a host synchronisation in between.  They all use the one
    scratch block of the context (look-back descriptors and tickets, or the
    plan arrays): the library has to order them, whatever the streams do."""
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    cases = []
    for i, (kind, n) in enumerate([("zipf", (9 << 20) + 12345), ("uniform", (6 << 20) + 77),
                                   ("geometric", (11 << 20) + 3), ("testfile", (7 << 20) + 65536)]):
        data = synth.generate_host(kind, n, seed=500 + i)
        codes, lens = orc.generate_extended_codewords(orc.histogram(data))
        want, _ = orc.encode_slice(data, codes, lens)
        t = to_gpu(data)
        model = dev.DeviceModel(codes, lens)
        bufs = dev.EncodeBuffers(model, n)
        bh = dev.new_batch_hist(n, t.device) if i >= 2 else None
        cases.append((data, want, t, model, bufs, bh, codes, lens))
    torch.cuda.synchronize()
    for rnd in range(3):
        for i, (data, want, t, model, bufs, bh, codes, lens) in enumerate(cases):
            bufs.data.zero_()
        torch.cuda.synchronize()
        for i, (data, want, t, model, bufs, bh, codes, lens) in enumerate(cases):
            with torch.cuda.stream(s[(i + rnd) % 2]):
                if bh is not None:
                    dev.histogram_async(t, batch_hist=bh)
                dev.encode_async(t, model, bufs, plan=bh)
        torch.cuda.synchronize()
        for data, want, t, model, bufs, bh, codes, lens in cases:
            bits, flags = bufs.result.tolist()[:2]
            assert flags == 0
            assert bits == int(np.asarray(lens, dtype=np.uint64)[data].sum())
            assert np.array_equal(bufs.data[: want.size].cpu().numpy(), want)


def test_sync_call_on_another_stream_waits_for_an_encode_in_flight(dev, orc):
    """hf_histogram keeps its result in the same scratch block: called on a
    second stream while a chained encode is still running on the first, it
    must not overwrite that encode's descriptors."""
    n = (24 << 20) + 1001
    data = synth.generate_host("zipf", n, seed=77)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    t = to_gpu(data)
    small = to_gpu(data[:100000])
    model = dev.DeviceModel(codes, lens)
    bufs = dev.EncodeBuffers(model, n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            dev.encode_async(t, model, bufs)
        with torch.cuda.stream(s2):
            h = dev.histogram(small)
        torch.cuda.synchronize()
        assert np.array_equal(h, orc.histogram(data[:100000]))
        assert bufs.result.tolist()[:2] == [int(np.asarray(lens, dtype=np.uint64)[data].sum()), 0]
        assert np.array_equal(bufs.data[: want.size].cpu().numpy(), want)


@pytest.mark.parametrize("chunk_syms", [64, 128, 256])
@pytest.mark.parametrize("width,nsym", [(16, 2), (16, 5), (20, 3), (13, 4), (32, 2), (22, 8)])
def test_fixed_length_long_codes_decode(dev, orc, chunk_syms, width, nsym):
    """A few fixed-length codes of 13..32 bits whose keys are multiples of
    2^(width - 11): the first-level table resolves every one of them (no
    escape entries), yet fewer than five fit a 64-bit window."""
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    step = (1 << width) // nsym
    step -= step % (1 << (width - 11))
    for i in range(nsym):
        codes[10 + i] = i * step
        lens[10 + i] = width
    rng = np.random.default_rng(width * 100 + nsym)
    n = 70000 + chunk_syms + 3
    data = (10 + rng.integers(0, nsym, size=n)).astype(np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, chunk_syms=chunk_syms)
    want, _ = orc.encode_slice(data, codes, lens)
    stream = enc.stream().cpu().numpy()
    assert np.array_equal(stream, want)
    got = dev.decode(enc, model).cpu().numpy()
    assert np.array_equal(got, orc.decode_read(stream, codes, lens, n))
    assert np.array_equal(got, data)
    # the same through a decode-only model built from the key table
    keys, syms, kl = orc.to_btreemap(codes, lens)
    dmodel = dev.DeviceModel(table=(keys, syms, kl, orc.sentinel(lens)))
    assert np.array_equal(dev.decode(enc, dmodel).cpu().numpy(), data)


@pytest.mark.parametrize("chunk_syms", [64, 128, 256])
@pytest.mark.parametrize("nsym,growth,heavy", [(34, 2.0, 2), (34, 2.0, 34), (20, 2.3, 20), (48, 1.7, 6),
                                               (58, 1.62, 58)])
def test_steep_codebook_on_data_that_does_not_follow_it(dev, orc, chunk_syms, nsym, growth, heavy):
    """A codebook for counts that grow like growth^k (lengths 1..nsym - 1: the
    decoder gets a two-symbol table and a speculative window width chosen for
    data that follows the model) applied to data in which the long codes are as
    frequent as the short ones: nearly every window of the fast path overruns,
    so every lane falls back, and all three decoders must still agree with the
    reference's.  (Several completed pairs in one window once wrapped the byte
    that sums their lengths.)"""
    hist = np.zeros(256, dtype=np.uint64)
    hist[100: 100 + nsym] = np.maximum(1, growth ** np.arange(nsym)[::-1]).astype(np.uint64)
    codes, lens = orc.generate_extended_codewords(hist)
    assert int(lens.max()) <= 57
    rng = np.random.default_rng(nsym * 1000 + heavy)
    n = 148255
    w = np.full(nsym, 1.0)
    w[:heavy] = 50.0  # the first `heavy` symbols (the longest codes) dominate
    data = (100 + rng.choice(nsym, size=n, p=w / w.sum())).astype(np.uint8)
    model = dev.DeviceModel(codes, lens)
    enc = dev.encode(to_gpu(data), model, chunk_syms=chunk_syms)
    want, _ = orc.encode_slice(data, codes, lens)
    stream = enc.stream().cpu().numpy()
    assert np.array_equal(stream, want)
    assert np.array_equal(dev.decode(enc, model).cpu().numpy(), data)
    cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n, chunk_syms)
    nc, ns = dev.gap_layout(n, chunk_syms)
    assert torch.equal(cb[:nc], enc.chunk_bits[:nc]) and torch.equal(so[:ns], enc.seg_off[:ns])
    keys, syms, kl = orc.to_btreemap(codes, lens)
    dmodel = dev.DeviceModel(table=(keys, syms, kl, orc.sentinel(lens)))
    assert np.array_equal(dev.decode(enc, dmodel).cpu().numpy(), data)


def test_prefix_code_with_unused_code_space(dev, orc):
    """Three codes, 000000 / 111100011 / 1100111010011011: no code is a prefix
    of another, but most of the code space is unused, and there the reference's
    predecessor rule answers with the nearest smaller key.  A window that holds
    only the first 14 bits of the 16-bit code, zero filled, lies in such a gap
    and reads as the 6-bit code -- which fits the window: a decoder that takes
    several codes from one window on the strength of 'what fits is right' goes
    wrong here, so those shortcuts are reserved for complete codes."""
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for sym, code, ln in [(229, 0b0, 6), (27, 0b111100011, 9), (155, 0b1100111010011011, 16)]:
        codes[sym], lens[sym] = code, ln
    rng = np.random.default_rng(413)
    n = 204355
    data = np.array([229, 27, 155], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    stream, _ = orc.encode_slice(data, codes, lens)
    want = orc.decode_read(stream, codes, lens, n)[:n]
    assert np.array_equal(want, data)
    model = dev.DeviceModel(codes, lens)
    for cs in (64, 128, 256):
        enc = dev.encode(to_gpu(data), model, chunk_syms=cs)
        assert np.array_equal(enc.stream().cpu().numpy(), stream)
        assert np.array_equal(dev.decode(enc, model).cpu().numpy(), want)
        cb, so = dev.build_gap_array(enc.data, enc.nbytes, model, n, cs)
        nc, ns = dev.gap_layout(n, cs)
        assert torch.equal(cb[:nc], enc.chunk_bits[:nc]) and torch.equal(so[:ns], enc.seg_off[:ns])


@pytest.mark.parametrize("seed", [7, 99, 413, 1234, 2024, 31337])
def test_random_key_tables_decode_like_the_reference(dev, orc, seed):
    """Arbitrary tables (distinct left-aligned keys, key 0 present, codes may be
    prefixes of each other or leave code space unused): the bare stream goes
    through the gap-array rebuild and the chunk-parallel decoder and must come
    out as the reference's sequential predecessor decode returns it."""
    import ctypes as C

    from huffman_amd import _lib
    from huffman_amd._lib import u8p

    rng = np.random.default_rng(seed)
    nsym, maxlen = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    keys, used = set(), []
    for i, s in enumerate(rng.permutation(256)[:nsym]):
        for _ in range(50):
            ln = int(rng.integers(1, maxlen + 1))
            code = 0 if i == 0 else int(rng.integers(0, 1 << ln))
            key = code << (maxlen - ln)
            if key not in keys:
                keys.add(key)
                codes[s], lens[s] = code, ln
                used.append(s)
                break
    sent = int(lens.max())
    if len({int(codes[s]) << (sent - int(lens[s])) for s in used}) != len(used):
        pytest.skip("keys collide once aligned to the longest length in use")
    n = 150001
    data = np.array(used, dtype=np.uint8)[rng.integers(0, len(used), size=n)]
    stream, _ = orc.encode_slice(data, codes, lens)
    try:
        want = orc.decode_read(stream, codes, lens, n)[:n]
    except ValueError:
        pytest.skip("the reference's own assertion (decode/mod.rs:207-210) refuses this stream")
    model = dev.DeviceModel(codes, lens)
    back = np.zeros(n, dtype=np.uint8)
    rc = _lib.lib().hf_decode_host(model.ctx.handle, model.handle, stream.ctypes.data_as(u8p),
                                   stream.size, n, 64, None, None, back.ctypes.data_as(u8p))
    assert rc == _lib.HF_OK, _lib.lib().hf_last_error(model.ctx.handle)
    assert np.array_equal(back, want)


def test_context_calls_leave_the_current_device_alone(dev, orc):
    """Every entry point switches to its context's device for the call only."""
    from huffman_amd._lib import lib

    before = torch.cuda.current_device()
    h = C.c_void_p()
    assert lib().hf_ctx_create(before, C.byref(h)) == 0
    assert torch.cuda.current_device() == before
    # a device that does not exist is refused, and nothing changes
    bad = C.c_void_p()
    assert lib().hf_ctx_create(torch.cuda.device_count(), C.byref(bad)) != 0
    assert torch.cuda.current_device() == before
    data = synth.generate_host("zipf", 50000, seed=9)
    t = to_gpu(data)
    hist = np.zeros(256, dtype=np.uint64)
    assert lib().hf_histogram(h, t.data_ptr(), t.numel(),
                              hist.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    assert np.array_equal(hist, orc.histogram(data))
    assert lib().hf_ctx_destroy(h) == 0
    assert torch.cuda.current_device() == before
