"""Pins the CPU oracle against every known-answer test of the reference crate
(values in tests/golden/reference_kats.json, each citing its reference test)."""
import numpy as np
import pytest


def numbers_model(orc, words):
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for w in words:
        codes[w] = w
        lens[w] = orc.calculate_length(w)
    return codes, lens


def test_u64_to_bytes(orc, kats):
    for num, expected in kats["u64_to_bytes"]["cases"]:
        assert orc.u64_to_bytes(num) == expected
        assert orc.bytes_to_u64(expected) == num


def test_histogram(orc, kats):
    k = kats["histogram"]
    h = orc.histogram(bytes(k["data"]))
    for sym, cnt in k["counts"].items():
        assert h[int(sym)] == cnt
    assert h.sum() == len(k["data"])


def test_codeword_lengths_inplace(orc, kats):
    for case in kats["codeword_lengths_inplace"]["cases"]:
        assert orc.codeword_lengths_inplace(case["weights"]) == case["lengths"]


def test_codeword_lengths_edge_cases(orc):
    assert orc.codeword_lengths_inplace([5]) == [0]
    assert orc.codeword_lengths_inplace([5, 3]) == [1, 1]
    with pytest.raises(ValueError):
        orc.codeword_lengths_inplace([])


def test_extended_codewords_sorted_input(orc, kats):
    k = kats["extended_codewords_sorted_input"]
    codes, lens = orc.generate_extended_codewords(k["histogram"])
    for sym, code in k["exact"].items():
        assert codes[int(sym)] == code
    for sym, (lo, hi) in k["ranges"].items():
        assert lo <= codes[int(sym)] <= hi
    d = k["derived_with_ascending_symbol_tie_rule"]
    assert [int(x) for x in codes[:10]] == d["codes"]
    assert [int(x) for x in lens[:10]] == d["lengths"]
    assert not codes[10:].any() and not lens[10:].any()


def test_single_symbol_gets_one_bit(orc):
    hist = np.zeros(256, dtype=np.uint64)
    hist[97] = 6
    codes, lens = orc.generate_extended_codewords(hist)
    assert lens[97] == 1 and codes[97] == 0 and lens.sum() == 1


def test_empty_histogram_is_an_error(orc):
    with pytest.raises(ValueError):
        orc.generate_extended_codewords(np.zeros(256, dtype=np.uint64))


def test_calculate_length(orc, kats):
    for v, expected in kats["calculate_length"]["cases"]:
        assert orc.calculate_length(v) == expected


def test_encode_numbers(orc, kats):
    k = kats["encode_numbers"]
    codes, lens = numbers_model(orc, k["words"])
    enc = orc.Encoder(codes, lens)
    ret = enc.write(bytes(k["words"]))
    enc.flush()
    assert list(enc.inner) == k["bytes"]
    assert ret == k["write_return"]
    assert enc.readbytes == len(k["words"])
    assert enc.writeout == len(k["bytes"])


def test_encode_stream(orc, kats):
    k = kats["encode_stream"]
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for sym, code in k["codes"].items():
        codes[int(sym)] = code
        lens[int(sym)] = orc.calculate_length(code)
    enc = orc.Encoder(codes, lens)
    ret = enc.write(bytes(k["symbols"]))
    enc.flush()
    assert list(enc.inner) == k["bytes"]
    assert ret == k["write_return"]


def test_codelen_over_64_is_invalid_data(orc):
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    lens[1] = 65
    enc = orc.Encoder(codes, lens)
    with pytest.raises(ValueError):
        enc.write(b"\x01")


def test_decode_numbers(orc, kats):
    words = kats["decode_numbers"]["words"]
    codes, lens = numbers_model(orc, words)
    enc = orc.Encoder(codes, lens)
    enc.write(bytes(words))
    enc.flush()
    out = orc.decode_read(enc.inner, codes, lens, enc.readbytes)
    assert list(out) == words


def test_decode_numbers_histogram_encoded(orc, kats):
    k = kats["decode_numbers_histogram_encoded"]
    hist = np.zeros(256, dtype=np.uint64)
    hist[: len(k["histogram"])] = k["histogram"]
    codes, lens = orc.generate_extended_codewords(hist)
    enc = orc.Encoder(codes, lens)
    enc.write(bytes(k["origin"]))
    enc.flush()
    out = orc.decode_read(enc.inner, codes, lens, enc.readbytes)
    assert list(out) == k["origin"]


def test_vault_roundtrips_blockwise_and_at_once(orc, kats):
    k = kats["vault_roundtrips"]
    for sentence, blocksize in zip(k["sentences"], k["blocksizes"]):
        data = sentence.encode()
        codes, lens = orc.generate_extended_codewords(orc.histogram(data))
        stream, _ = orc.encode_slice(data, codes, lens)
        # at once, slice decoder
        assert orc.decode_read(stream, codes, lens, len(data)).tobytes() == data
        # blockwise, streaming decoder
        dec = orc.Decoder(stream, codes, lens, readbytes=len(data))
        got = b""
        while True:
            block = dec.read(blocksize)
            if not block:
                break
            got += block
        assert got == data


def test_header_roundtrip_layout(orc, kats):
    words = kats["header_roundtrip"]["words"]
    codes, lens = numbers_model(orc, words)
    keys, syms, kl = orc.to_btreemap(codes, lens)
    sent = orc.sentinel(lens)
    blob = orc.header_to_binary(keys, syms, kl, sent, 0)
    assert len(blob) == 36 + 10 * len(keys)
    assert blob[:8] == (4).to_bytes(8, "little") and blob[8:12] == b"pzhf"
    n = int.from_bytes(blob[12:20], "little")
    assert n == len(keys) == len(words)
    off = 20
    for i in range(n):
        assert int.from_bytes(blob[off:off + 8], "little") == keys[i]
        assert blob[off + 8] == syms[i] and blob[off + 9] == kl[i]
        off += 10
    assert int.from_bytes(blob[off:off + 8], "little") == sent
    assert int.from_bytes(blob[off + 8:off + 16], "little") == 0


def test_streaming_decoder_matches_slice_decoder_small_reads(orc):
    rng = np.random.default_rng(7)
    w = np.array([20, 17, 6, 3, 2, 2, 2, 1, 1, 1], dtype=np.float64)
    data = rng.choice(10, size=5000, p=w / w.sum()).astype(np.uint8)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    stream, _ = orc.encode_slice(data, codes, lens)
    dec = orc.Decoder(stream, codes, lens, readbytes=data.size)
    got = bytearray()
    while True:
        block = dec.read(15)  # examples/decode.rs:57 uses 15-byte reads
        if not block:
            break
        got += block
    assert bytes(got) == data.tobytes()
    assert orc.decode_read(stream, codes, lens, data.size).tobytes() == data.tobytes()


def test_container_layout(orc):
    data = b"This is a lovely text in a big world"
    blob = orc.compress_with_header(data)
    assert blob[:4] == b"pzhf"
    hlen = int.from_bytes(blob[4:12], "big")
    header = blob[12:12 + hlen]
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    keys, syms, kl = orc.to_btreemap(codes, lens)
    assert header == orc.header_to_binary(keys, syms, kl, orc.sentinel(lens), len(data))
    stream, _ = orc.encode_slice(data, codes, lens)
    assert blob[12 + hlen:] == stream.tobytes()
