"""GPU tests of the host-buffer C entry points, of the C++ mirror of the
reference API (include/hfgpu.hpp) and of the command line tool, against the
CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from huffman_amd import _lib, synth  # noqa: E402
from huffman_amd._lib import u8p, u64p  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIRROR = os.path.join(ROOT, "tests", "cpp", "test_mirror")
CLI = os.path.join(ROOT, "huffman_amd", "bin", "hfgpu")
u16p = C.POINTER(C.c_uint16)


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (there is no CPU fallback)")
    h = C.c_void_p()
    assert _lib.lib().hf_ctx_create(0, C.byref(h)) == _lib.HF_OK
    yield h
    _lib.lib().hf_ctx_destroy(h)


def make_model(ctx, codes, lens):
    m = C.c_void_p()
    rc = _lib.lib().hf_model_create(ctx, codes.ctypes.data_as(u64p), lens.ctypes.data_as(u64p),
                                    C.byref(m))
    assert rc == _lib.HF_OK, _lib.lib().hf_last_error(ctx)
    return m


@pytest.mark.parametrize("n", [0, 1, 4097, 300001])
def test_histogram_host(ctx, orc, n):
    data = synth.generate_host("zipf", n, seed=n + 5)
    hist = np.zeros(256, dtype=np.uint64)
    rc = _lib.lib().hf_histogram_host(ctx, data.ctypes.data_as(u8p), n, hist.ctypes.data_as(u64p))
    assert rc == _lib.HF_OK
    assert np.array_equal(hist, orc.histogram(data))


@pytest.mark.parametrize("kind", ["zipf", "testfile", "geometric"])
@pytest.mark.parametrize("n", [1, 65, 16384, 100003, (1 << 21) + 77])
@pytest.mark.parametrize("gap", [True, False])
def test_encode_decode_host(ctx, orc, kind, n, gap):
    L = _lib.lib()
    data = synth.generate_host(kind, n, seed=n)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    model = make_model(ctx, codes, lens)
    try:
        cap = int(L.hf_encode_bound(model, n, 0))
        out = np.zeros(cap, dtype=np.uint8)
        nc, ns = C.c_uint64(), C.c_uint64()
        assert L.hf_gap_layout(n, 64, C.byref(nc), C.byref(ns)) == _lib.HF_OK
        cb = np.zeros(max(nc.value, 1), dtype=np.uint16)
        so = np.zeros(max(ns.value, 1), dtype=np.uint64)
        bits = C.c_uint64()
        rc = L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), n, 0, out.ctypes.data_as(u8p),
                              cap, 64, cb.ctypes.data_as(u16p) if gap else None,
                              so.ctypes.data_as(u64p) if gap else None, C.byref(bits))
        assert rc == _lib.HF_OK, L.hf_last_error(ctx)
        nbytes = (bits.value + 7) // 8
        assert nbytes == want.size and np.array_equal(out[:nbytes], want)
        back = np.zeros(n, dtype=np.uint8)
        rc = L.hf_decode_host(ctx, model, out.ctypes.data_as(u8p), nbytes, n, 64,
                              cb.ctypes.data_as(u16p) if gap else None,
                              so.ctypes.data_as(u64p) if gap else None, back.ctypes.data_as(u8p))
        assert rc == _lib.HF_OK, L.hf_last_error(ctx)
        assert np.array_equal(back, data)
    finally:
        L.hf_model_destroy(model)


def test_encode_host_capacity_error(ctx, orc):
    L = _lib.lib()
    data = synth.generate_host("uniform", 50000, seed=2)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = make_model(ctx, codes, lens)
    try:
        out = np.zeros(1000, dtype=np.uint8)
        bits = C.c_uint64()
        rc = L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), data.size, 0,
                              out.ctypes.data_as(u8p), out.size, 64, None, None, C.byref(bits))
        assert rc == _lib.HF_ERR_CAPACITY
    finally:
        L.hf_model_destroy(model)


def test_write_value_host_matches_oracle_automaton(ctx, orc):
    L = _lib.lib()
    data = synth.generate_host("testfile", 50000, seed=9)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    enc = orc.Encoder(codes, lens)
    state = (C.c_uint64 * 3)(0, 64, 0)
    for off in range(0, data.size, 4096):
        piece = np.ascontiguousarray(data[off:off + 4096])
        want = enc.write(piece)
        ret = C.c_uint64()
        rc = L.hf_encoder_write_value_host(ctx, piece.ctypes.data_as(u8p), piece.size,
                                           codes.ctypes.data_as(u64p), lens.ctypes.data_as(u64p),
                                           state, C.byref(ret))
        assert rc == _lib.HF_OK
        assert ret.value == want
        assert state[2] == enc.writeout


def test_cpp_mirror_replays_reference_tests():
    assert os.path.exists(MIRROR), "build first: python huffman_amd/build.py"
    r = subprocess.run([MIRROR], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0 and "ALL OK" in r.stdout
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 17


def run_cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300, env=e)


@pytest.mark.parametrize("kind,n", [("testfile", 1 << 20), ("zipf", 3_000_001), ("uniform", 5)])
def test_cli_container_matches_oracle_and_round_trips(tmp_path, orc, kind, n):
    assert os.path.exists(CLI), "build first: python huffman_amd/build.py"
    data = synth.generate_host(kind, n, seed=21)
    src, dst, back = tmp_path / "in.bin", tmp_path / "out.pzhf", tmp_path / "back.bin"
    src.write_bytes(data.tobytes())
    r = run_cli(str(src), str(dst), "h", "c", env={"RUST_LOG": "info"})
    assert r.returncode == 0, r.stderr
    assert "Compressing file" in r.stderr
    assert dst.read_bytes() == orc.compress_with_header(data)
    assert os.path.exists(str(dst) + ".hfgap")
    r = run_cli(str(dst), str(back), "h", "d")
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == data.tobytes()
    # a file as the reference writes it, without the side file: the gap array is
    # rebuilt on the device from the stream
    os.remove(str(dst) + ".hfgap")
    back.unlink()
    r = run_cli(str(dst), str(back), "h", "d")
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == data.tobytes()


def test_cli_headerless_stream_and_argument_errors(tmp_path, orc):
    data = synth.generate_host("testfile", 200001, seed=4)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    src.write_bytes(data.tobytes())
    r = run_cli(str(src), str(dst), "x", "c")
    assert r.returncode == 0, r.stderr
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    assert dst.read_bytes() == want.tobytes()
    assert run_cli(str(src)).returncode == 101
    assert run_cli(str(src), str(dst), "h").returncode == 101
    assert run_cli(str(src), str(dst), "h", "q").returncode == 101
    assert run_cli(str(tmp_path / "missing"), str(dst), "h", "c").returncode == 101


@pytest.mark.parametrize("kind,n", [("zipf", 1), ("testfile", 65536), ("geometric", 300001),
                                    ("uniform", (1 << 21) + 5)])
def test_compress_host_single_upload(ctx, orc, kind, n):
    """from_reader + write + flush in one call: same codebook, stream and gap
    array as the separate host entry points (and the oracle)."""
    L = _lib.lib()
    data = synth.generate_host(kind, n, seed=n + 3)
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    out = np.zeros(n + 16, dtype=np.uint8)
    nc, ns = C.c_uint64(), C.c_uint64()
    assert L.hf_gap_layout(n, 64, C.byref(nc), C.byref(ns)) == _lib.HF_OK
    cb = np.zeros(max(nc.value, 1), dtype=np.uint16)
    so = np.zeros(max(ns.value, 1), dtype=np.uint64)
    bits = C.c_uint64()
    rc = L.hf_compress_host(ctx, data.ctypes.data_as(u8p), n, codes.ctypes.data_as(u64p),
                            lens.ctypes.data_as(u64p), out.ctypes.data_as(u8p), out.size, 64,
                            cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p), C.byref(bits))
    assert rc == _lib.HF_OK, L.hf_last_error(ctx)
    wc, wl = orc.generate_extended_codewords(orc.histogram(data))
    assert np.array_equal(codes, wc) and np.array_equal(lens, wl)
    want, _ = orc.encode_slice(data, wc, wl)
    nbytes = (bits.value + 7) // 8
    assert nbytes == want.size and np.array_equal(out[:nbytes], want)
    l = np.asarray(wl, dtype=np.uint64)[data]
    pad = np.zeros(nc.value * 64, dtype=np.uint64)
    pad[:n] = l
    assert np.array_equal(cb[: nc.value].astype(np.uint64), pad.reshape(-1, 64).sum(axis=1))
    # too small an output buffer and an empty input are reported, not written
    small = np.zeros(8, dtype=np.uint8)
    if want.size > 8:
        rc = L.hf_compress_host(ctx, data.ctypes.data_as(u8p), n, codes.ctypes.data_as(u64p),
                                lens.ctypes.data_as(u64p), small.ctypes.data_as(u8p), small.size,
                                64, None, None, C.byref(bits))
        assert rc == _lib.HF_ERR_CAPACITY
    rc = L.hf_compress_host(ctx, data.ctypes.data_as(u8p), 0, codes.ctypes.data_as(u64p),
                            lens.ctypes.data_as(u64p), out.ctypes.data_as(u8p), out.size, 64,
                            None, None, C.byref(bits))
    assert rc == _lib.HF_ERR_EMPTY


@pytest.mark.parametrize("kind,cs", [("zipf", 64), ("uniform", 128), ("geometric", 256), ("testfile", 64)])
def test_decode_host_in_overlapping_pieces(ctx, orc, kind, cs):
    """A stream with gap arrays that spans several pieces (here 1 MiB instead of
    the default 64 MiB) is uploaded, decoded and downloaded piece by piece."""
    L = _lib.lib()
    n = 5 * (1 << 20) + 12345
    data = synth.generate_host(kind, n, seed=cs)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = make_model(ctx, codes, lens)
    try:
        cap = int(L.hf_encode_bound(model, n, 0))
        out = np.zeros(cap, dtype=np.uint8)
        nc, ns = C.c_uint64(), C.c_uint64()
        assert L.hf_gap_layout(n, cs, C.byref(nc), C.byref(ns)) == _lib.HF_OK
        cb = np.zeros(nc.value, dtype=np.uint16)
        so = np.zeros(ns.value, dtype=np.uint64)
        bits = C.c_uint64()
        assert L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), n, 0, out.ctypes.data_as(u8p),
                                cap, cs, cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p),
                                C.byref(bits)) == _lib.HF_OK
        nbytes = (bits.value + 7) // 8
        assert L.hf_ctx_set_option(ctx, b"host_piece_mib", 1) == _lib.HF_OK
        try:
            back = np.zeros(n, dtype=np.uint8)
            rc = L.hf_decode_host(ctx, model, out.ctypes.data_as(u8p), nbytes, n, cs,
                                  cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p),
                                  back.ctypes.data_as(u8p))
            assert rc == _lib.HF_OK, L.hf_last_error(ctx)
            assert np.array_equal(back, data)
            # a damaged piece is reported: a window below the smallest key cannot occur
            # with a complete code, so shorten the stream instead (the tail decodes zeros)
            back[:] = 0
            rc = L.hf_decode_host(ctx, model, out.ctypes.data_as(u8p), nbytes, n - 7, cs,
                                  cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p),
                                  back.ctypes.data_as(u8p))
            assert rc == _lib.HF_OK and np.array_equal(back[: n - 7], data[: n - 7])
        finally:
            assert L.hf_ctx_set_option(ctx, b"host_piece_mib", 64) == _lib.HF_OK
    finally:
        L.hf_model_destroy(model)


@pytest.mark.parametrize("kind,cs,lead", [("zipf", 64, 0), ("uniform", 128, 3), ("geometric", 256, 0),
                                          ("testfile", 64, 5)])
def test_encode_host_in_overlapping_pieces(ctx, orc, kind, cs, lead):
    """An input of several pieces (1 MiB here) is uploaded, encoded and downloaded
    piece by piece; stream and gap arrays equal those of a single call."""
    L = _lib.lib()
    n = 5 * (1 << 20) + 4321
    data = synth.generate_host(kind, n, seed=cs + lead)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    model = make_model(ctx, codes, lens)
    try:
        cap = int(L.hf_encode_bound(model, n, lead))
        nc, ns = C.c_uint64(), C.c_uint64()
        assert L.hf_gap_layout(n, cs, C.byref(nc), C.byref(ns)) == _lib.HF_OK
        results = []
        for piece_mib in (64, 1):
            assert L.hf_ctx_set_option(ctx, b"host_piece_mib", piece_mib) == _lib.HF_OK
            out = np.zeros(cap, dtype=np.uint8)
            cb = np.zeros(nc.value, dtype=np.uint16)
            so = np.zeros(ns.value, dtype=np.uint64)
            bits = C.c_uint64()
            rc = L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), n, lead,
                                  out.ctypes.data_as(u8p), cap, cs, cb.ctypes.data_as(u16p),
                                  so.ctypes.data_as(u64p), C.byref(bits))
            assert rc == _lib.HF_OK, L.hf_last_error(ctx)
            results.append((bits.value, out[: (lead + bits.value + 7) // 8].copy(), cb, so))
        assert L.hf_ctx_set_option(ctx, b"host_piece_mib", 64) == _lib.HF_OK
        (b0, s0, cb0, so0), (b1, s1, cb1, so1) = results
        assert b0 == b1 and np.array_equal(s0, s1)
        assert np.array_equal(cb0, cb1) and np.array_equal(so0, so1)
        if lead == 0:
            want, _ = orc.encode_slice(data, codes, lens)
            assert np.array_equal(s1, want)
        # too small an output buffer is reported by the pieced path as well
        assert L.hf_ctx_set_option(ctx, b"host_piece_mib", 1) == _lib.HF_OK
        small = np.zeros(s1.size - 5, dtype=np.uint8)
        bits = C.c_uint64()
        rc = L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), n, lead,
                              small.ctypes.data_as(u8p), small.size, cs, None, None, C.byref(bits))
        assert rc == _lib.HF_ERR_CAPACITY
    finally:
        L.hf_ctx_set_option(ctx, b"host_piece_mib", 64)
        L.hf_model_destroy(model)


# ---- a stream decoded window by window / range by range (what Decoder::read does
# ---- with a reader it drains piece by piece)

@pytest.mark.parametrize("kind", ["zipf", "geometric", "testfile", "uniform"])
def test_decode_window_host_walks_a_stream_without_gap_array(ctx, orc, kind):
    L = _lib.lib()
    n = 5_000_011
    data = synth.generate_host(kind, n, seed=77)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    stream, _ = orc.encode_slice(data, codes, lens)
    model = make_model(ctx, codes, lens)
    try:
        window = 1 << 18
        base, start_bit, done = 0, 0, 0
        pieces = []
        calls = 0
        while done < n:
            win = stream[base: base + window]
            last = base + window >= stream.size
            n_max = min(n - done, win.size * 8 + 8)
            out = np.zeros(n_max, dtype=np.uint8)
            got, end = C.c_uint64(), C.c_uint64()
            whole = last and n_max == n - done
            rc = L.hf_decode_window_host(ctx, model, win.ctypes.data_as(u8p), win.size, start_bit,
                                         n_max, 1 if whole else 0, out.ctypes.data_as(u8p),
                                         C.byref(got), C.byref(end))
            assert rc == _lib.HF_OK, L.hf_last_error(ctx)
            calls += 1
            assert got.value > 0
            if not whole:
                assert got.value % (64 * 256) == 0  # whole segments only
            pieces.append(out[: got.value])
            done += got.value
            base += end.value // 8
            start_bit = end.value % 8
        assert calls > 2
        assert np.array_equal(np.concatenate(pieces), data)
    finally:
        L.hf_model_destroy(model)


@pytest.mark.parametrize("kind,cs", [("zipf", 64), ("geometric", 64), ("uniform", 128)])
def test_decode_range_host_decodes_runs_of_segments(ctx, orc, kind, cs):
    L = _lib.lib()
    n = 3_000_017
    data = synth.generate_host(kind, n, seed=78)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    want, _ = orc.encode_slice(data, codes, lens)
    model = make_model(ctx, codes, lens)
    try:
        nch, nsg = (n + cs - 1) // cs, ((n + cs - 1) // cs + 255) // 256
        out = np.zeros(want.size + 64, dtype=np.uint8)
        cb = np.zeros(nch, dtype=np.uint16)
        so = np.zeros(nsg, dtype=np.uint64)
        bits = C.c_uint64()
        rc = L.hf_encode_host(ctx, model, data.ctypes.data_as(u8p), n, 0, out.ctypes.data_as(u8p),
                              out.size, cs, cb.ctypes.data_as(u16p), so.ctypes.data_as(u64p),
                              C.byref(bits))
        assert rc == _lib.HF_OK
        stream = out[: want.size]
        assert np.array_equal(stream, want)
        seg_syms = 256 * cs
        per_piece = 7  # segments
        got = []
        for s0 in range(0, nsg, per_piece):
            s1 = min(nsg, s0 + per_piece)
            n_piece = min(n, s1 * seg_syms) - s0 * seg_syms
            first_byte = (int(so[s0]) // 8) & ~15
            end_bit = int(so[s1]) if s1 < nsg else int(bits.value)
            win = np.ascontiguousarray(stream[first_byte: (end_bit + 7) // 8 + 16])
            piece = np.zeros(n_piece, dtype=np.uint8)
            rc = L.hf_decode_range_host(ctx, model, win.ctypes.data_as(u8p), win.size,
                                        first_byte * 8, n_piece, cs,
                                        cb[s0 * 256:].ctypes.data_as(u16p),
                                        so[s0:].ctypes.data_as(u64p), piece.ctypes.data_as(u8p))
            assert rc == _lib.HF_OK, L.hf_last_error(ctx)
            got.append(piece)
        assert np.array_equal(np.concatenate(got), data)
    finally:
        L.hf_model_destroy(model)


# ---- side files and lengths that cannot be trusted

def test_cli_ignores_side_files_that_do_not_fit_and_rejects_bad_lengths(tmp_path, orc):
    n = 700_001
    a = synth.generate_host("zipf", n, seed=31)
    b = synth.generate_host("testfile", n, seed=32)  # same symbol count, another stream
    fa, fb = tmp_path / "a.bin", tmp_path / "b.bin"
    fa.write_bytes(a.tobytes())
    fb.write_bytes(b.tobytes())
    ca, cbf, back = tmp_path / "a.pzhf", tmp_path / "b.pzhf", tmp_path / "back.bin"
    assert run_cli(str(fa), str(ca), "h", "c").returncode == 0
    assert run_cli(str(fb), str(cbf), "h", "c").returncode == 0
    side_a, side_b = str(ca) + ".hfgap", str(cbf) + ".hfgap"
    good = open(side_a, "rb").read()
    # a foreign side file (same number of symbols, other stream)
    open(side_a, "wb").write(open(side_b, "rb").read())
    r = run_cli(str(ca), str(back), "h", "d")
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == a.tobytes()
    # truncated, and with counts that do not account for the file
    for bad in (good[: len(good) // 2], good[:20], good[:12] + b"\xff" * 24 + good[36:],
                good + b"\0\0"):
        open(side_a, "wb").write(bad)
        back.unlink()
        r = run_cli(str(ca), str(back), "h", "d")
        assert r.returncode == 0, r.stderr
        assert back.read_bytes() == a.tobytes()
    # one chunk length off by one: no longer consistent in itself
    tampered = bytearray(good)
    tampered[36 + 2 * 1000] ^= 1
    open(side_a, "wb").write(bytes(tampered))
    back.unlink()
    r = run_cli(str(ca), str(back), "h", "d")
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == a.tobytes()
    # a header length beyond the file, and a table count beyond the header: loud errors
    whole = bytearray(ca.read_bytes())
    broken = tmp_path / "broken.pzhf"
    big = bytearray(whole)
    big[4:12] = (1 << 40).to_bytes(8, "big")
    broken.write_bytes(bytes(big))
    r = run_cli(str(broken), str(back), "h", "d")
    assert r.returncode == 101 and "header length" in r.stderr
    cnt = bytearray(whole)
    cnt[12 + 12: 12 + 20] = (1 << 50).to_bytes(8, "little")  # entry count of the bincode map
    broken.write_bytes(bytes(cnt))
    r = run_cli(str(broken), str(back), "h", "d")
    assert r.returncode == 101


def test_cli_streams_a_file_larger_than_its_pieces(tmp_path, orc):
    """1 GiB + 96 MiB: beyond the single-upload path, so the histogram is counted
    in 64 MiB pieces, the encoder drains 256 MiB at a time and the decoder works
    through 64 Mi-symbol pieces (with the side file) or 32 MiB windows of the
    stream (without); neither process may hold the whole file."""
    n = (1 << 30) + (96 << 20) + 4321
    data = synth.generate_host("zipf", n, seed=5)
    src, dst, back = tmp_path / "in.bin", tmp_path / "out.pzhf", tmp_path / "back.bin"
    src.write_bytes(data.tobytes())

    def run_measured(*args):
        """The tool as a child of its own; returns the peak resident set of its
        address space in MiB (VmHWM of /proc/<pid>/status, sampled while it runs:
        the rusage figure would include this test process's own size at fork)."""
        import time

        p = subprocess.Popen([CLI, *args], stderr=subprocess.PIPE)
        peak = 0.0
        while p.poll() is None:
            try:
                with open(f"/proc/{p.pid}/status") as f:
                    for line in f:
                        if line.startswith("VmHWM:"):
                            peak = max(peak, int(line.split()[1]) / 1024.0)
            except OSError:
                pass
            time.sleep(0.02)
        assert p.returncode == 0, p.stderr.read().decode()
        return peak

    peaks = [run_measured(str(src), str(dst), "h", "c")]
    want = orc.compress_with_header(data)
    got = np.fromfile(dst, dtype=np.uint8)
    assert got.size == len(want) and np.array_equal(got, np.frombuffer(want, dtype=np.uint8))
    del want, got
    peaks.append(run_measured(str(dst), str(back), "h", "d"))
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), data)
    os.remove(str(dst) + ".hfgap")
    back.unlink()
    peaks.append(run_measured(str(dst), str(back), "h", "d"))
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), data)
    print("peak resident MiB (compress, decompress with side file, without):", peaks)
    # the file is 1.1 GiB, its stream 0.8 GiB: every process stays well below either
    assert max(peaks) < 700, peaks
