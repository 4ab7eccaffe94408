"""ctypes view of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke`` and
by the ``cpu_baseline`` leg of ``bench.py`` as the checker.  The shipped package
``huffman_amd`` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_histogram.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.orc_histogram.restype = None
        L.orc_codeword_lengths_inplace.argtypes = [u64p, C.c_uint64]
        L.orc_codeword_lengths_inplace.restype = C.c_int
        L.orc_sort_by_value.argtypes = [u64p, C.c_uint64, u64p, u64p]
        L.orc_sort_by_value.restype = C.c_int64
        L.orc_codewords_from_lengths.argtypes = [u64p, C.c_uint64, u64p, u64p]
        L.orc_codewords_from_lengths.restype = C.c_int
        L.orc_generate_extended_codewords.argtypes = [u64p, C.c_uint64, u64p, u64p]
        L.orc_generate_extended_codewords.restype = C.c_int
        L.orc_sentinel.argtypes = [u64p]
        L.orc_sentinel.restype = C.c_uint64
        L.orc_to_btreemap.argtypes = [u64p, u64p, u64p, u8p, u8p]
        L.orc_to_btreemap.restype = C.c_uint64
        L.orc_calculate_length.argtypes = [C.c_uint64]
        L.orc_calculate_length.restype = C.c_uint64
        L.orc_encoder_new.argtypes = [u64p, u64p]
        L.orc_encoder_new.restype = C.c_void_p
        L.orc_encoder_free.argtypes = [C.c_void_p]
        L.orc_encoder_free.restype = None
        L.orc_encoder_plain_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_encoder_plain_write.restype = None
        L.orc_encoder_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_encoder_write.restype = C.c_int64
        L.orc_encoder_flush.argtypes = [C.c_void_p]
        L.orc_encoder_flush.restype = None
        L.orc_encoder_data.argtypes = [C.c_void_p]
        L.orc_encoder_data.restype = C.c_void_p
        for name in ("len", "readbytes", "writeout"):
            f = getattr(L, "orc_encoder_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_uint64
        L.orc_encoder_fillbits.argtypes = [C.c_void_p]
        L.orc_encoder_fillbits.restype = C.c_int32
        L.orc_encode_slice.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, C.c_void_p,
                                       C.c_uint64, C.POINTER(C.c_int64)]
        L.orc_encode_slice.restype = C.c_int64
        L.orc_search_key_or_next_small_key.argtypes = [u64p, C.c_uint64, C.c_uint64]
        L.orc_search_key_or_next_small_key.restype = C.c_int64
        L.orc_decode_read.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, C.c_uint64,
                                      C.c_void_p, C.c_uint64]
        L.orc_decode_read.restype = C.c_int64
        L.orc_decoder_new.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, C.c_uint64]
        L.orc_decoder_new.restype = C.c_void_p
        L.orc_decoder_from_table.argtypes = [C.c_void_p, C.c_uint64, u64p, u8p, u8p,
                                             C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_decoder_from_table.restype = C.c_void_p
        L.orc_decoder_free.argtypes = [C.c_void_p]
        L.orc_decoder_free.restype = None
        L.orc_decoder_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_decoder_read.restype = C.c_int64
        L.orc_u64_to_bytes.argtypes = [C.c_uint64, u8p]
        L.orc_u64_to_bytes.restype = None
        L.orc_bytes_to_u64.argtypes = [u8p]
        L.orc_bytes_to_u64.restype = C.c_uint64
        L.orc_header_to_binary.argtypes = [u64p, u8p, u8p, C.c_uint64, C.c_uint64,
                                           C.c_uint64, C.c_void_p]
        L.orc_header_to_binary.restype = C.c_uint64
        L.orc_compress_with_header.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_compress_with_header.restype = C.c_int64
        _lib = L
    return _lib


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p64(a):
    return a.ctypes.data_as(u64p)


def _p8(a):
    return a.ctypes.data_as(u8p)


def _bytes_arr(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def histogram(data) -> np.ndarray:
    d = _bytes_arr(data)
    h = np.zeros(256, dtype=np.uint64)
    lib().orc_histogram(d.ctypes.data, d.size, _p64(h))
    return h


def codeword_lengths_inplace(weights) -> list[int]:
    w = _u64(list(weights)).copy()
    rc = lib().orc_codeword_lengths_inplace(_p64(w), w.size)
    if rc != 0:
        raise ValueError("empty weight list")
    return [int(x) for x in w]


def sort_by_value(store) -> list[tuple[int, int]]:
    s = _u64(list(store))
    idx = np.zeros(256, dtype=np.uint64)
    cnt = np.zeros(256, dtype=np.uint64)
    m = lib().orc_sort_by_value(_p64(s), s.size, _p64(idx), _p64(cnt))
    if m < 0:
        raise ValueError("more than 256 entries")
    return [(int(idx[i]), int(cnt[i])) for i in range(m)]


def codewords_from_lengths(lengths) -> tuple[list[int], list[int]]:
    l = _u64(list(lengths))
    small = np.zeros(l.size, dtype=np.uint64)
    big = np.zeros(l.size, dtype=np.uint64)
    rc = lib().orc_codewords_from_lengths(_p64(l), l.size, _p64(small), _p64(big))
    if rc != 0:
        raise ValueError("bad lengths")
    return [int(x) for x in small], [int(x) for x in big]


def generate_extended_codewords(hist) -> tuple[np.ndarray, np.ndarray]:
    h = _u64(list(hist))
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    rc = lib().orc_generate_extended_codewords(_p64(h), h.size, _p64(codes), _p64(lens))
    if rc != 0:
        raise ValueError(f"generate_extended_codewords failed rc={rc}")
    return codes, lens


def sentinel(lens) -> int:
    return int(lib().orc_sentinel(_p64(_u64(lens))))


def to_btreemap(codes, lens):
    codes = _u64(codes)
    lens = _u64(lens)
    keys = np.zeros(256, dtype=np.uint64)
    syms = np.zeros(256, dtype=np.uint8)
    kl = np.zeros(256, dtype=np.uint8)
    m = lib().orc_to_btreemap(_p64(codes), _p64(lens), _p64(keys), _p8(syms), _p8(kl))
    return keys[:m].copy(), syms[:m].copy(), kl[:m].copy()


def calculate_length(v: int) -> int:
    return int(lib().orc_calculate_length(v))


class Encoder:
    """Restatement of Encoder<Cursor<Vec<u8>>, Huffman> (src/huffman/encode.rs:26-137)."""

    def __init__(self, codes, lens):
        self._codes = _u64(codes).copy()
        self._lens = _u64(lens).copy()
        self._e = lib().orc_encoder_new(_p64(self._codes), _p64(self._lens))

    def __del__(self):
        if getattr(self, "_e", None):
            lib().orc_encoder_free(self._e)
            self._e = None

    def plain_write(self, b: bytes) -> None:
        a = _bytes_arr(b)
        lib().orc_encoder_plain_write(self._e, a.ctypes.data, a.size)

    def write(self, data) -> int:
        a = _bytes_arr(data)
        r = lib().orc_encoder_write(self._e, a.ctypes.data, a.size)
        if r < 0:
            raise ValueError("Codelen > 64")
        return int(r)

    def flush(self) -> None:
        lib().orc_encoder_flush(self._e)

    @property
    def inner(self) -> bytes:
        n = lib().orc_encoder_len(self._e)
        return C.string_at(lib().orc_encoder_data(self._e), n) if n else b""

    @property
    def readbytes(self) -> int:
        return int(lib().orc_encoder_readbytes(self._e))

    @property
    def writeout(self) -> int:
        return int(lib().orc_encoder_writeout(self._e))

    @property
    def fillbits(self):
        f = lib().orc_encoder_fillbits(self._e)
        return None if f < 0 else int(f)


def encode_slice(data, codes, lens) -> tuple[np.ndarray, int]:
    """Encoder::new + write + flush on one slice; returns (bytes, write() value)."""
    d = _bytes_arr(data)
    codes = _u64(codes)
    lens = _u64(lens)
    cap = int(d.size) * 8 + 16
    out = np.empty(cap, dtype=np.uint8)
    ret = C.c_int64(0)
    n = lib().orc_encode_slice(d.ctypes.data, d.size, _p64(codes), _p64(lens),
                               out.ctypes.data, cap, C.byref(ret))
    if n == -2:
        cap = int(d.size) * 8 * 8 + 16
        out = np.empty(cap, dtype=np.uint8)
        n = lib().orc_encode_slice(d.ctypes.data, d.size, _p64(codes), _p64(lens),
                                   out.ctypes.data, cap, C.byref(ret))
    if n < 0:
        raise ValueError(f"encode failed rc={n}")
    return out[:n].copy(), int(ret.value)


def decode_read(data, codes, lens, goalsbyte: int) -> np.ndarray:
    """decode::read(data, model, goalsbyte) (src/huffman/decode/mod.rs:167-231)."""
    d = _bytes_arr(data)
    codes = _u64(codes)
    lens = _u64(lens)
    cap = max(int(goalsbyte), int(d.size) * 64) + 64
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_decode_read(d.ctypes.data, d.size, _p64(codes), _p64(lens), goalsbyte,
                              out.ctypes.data, cap)
    if n < 0:
        raise ValueError(f"decode failed rc={n}")
    return out[:n].copy()


class Decoder:
    """Restatement of the streaming Decoder<R> (src/huffman/decode/mod.rs:15-138)."""

    def __init__(self, inner, codes=None, lens=None, readbytes=0, table=None):
        self._inner = _bytes_arr(inner).copy()
        if table is None:
            c = _u64(codes)
            l = _u64(lens)
            self._d = lib().orc_decoder_new(self._inner.ctypes.data, self._inner.size,
                                            _p64(c), _p64(l), readbytes)
        else:
            keys, syms, kl, sent = table
            keys = _u64(keys)
            syms = np.ascontiguousarray(syms, dtype=np.uint8)
            kl = np.ascontiguousarray(kl, dtype=np.uint8)
            self._d = lib().orc_decoder_from_table(self._inner.ctypes.data, self._inner.size,
                                                   _p64(keys), _p8(syms), _p8(kl), keys.size,
                                                   sent, readbytes)

    def __del__(self):
        if getattr(self, "_d", None):
            lib().orc_decoder_free(self._d)
            self._d = None

    def read(self, n: int) -> bytes:
        buf = np.empty(max(n, 1), dtype=np.uint8)
        r = lib().orc_decoder_read(self._d, buf.ctypes.data, n)
        if r < 0:
            raise ValueError("decode failed")
        return buf[:r].tobytes()


def u64_to_bytes(num: int) -> list[int]:
    out = (C.c_uint8 * 8)()
    lib().orc_u64_to_bytes(num, out)
    return list(out)


def bytes_to_u64(b) -> int:
    arr = (C.c_uint8 * 8)(*list(b))
    return int(lib().orc_bytes_to_u64(arr))


def header_to_binary(keys, syms, klens, sentinel_: int, readbytes: int) -> bytes:
    keys = _u64(keys)
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    klens = np.ascontiguousarray(klens, dtype=np.uint8)
    out = np.empty(36 + 10 * keys.size, dtype=np.uint8)
    n = lib().orc_header_to_binary(_p64(keys), _p8(syms), _p8(klens), keys.size, sentinel_,
                                   readbytes, out.ctypes.data)
    return out[:n].tobytes()


def compress_with_header(data) -> bytes:
    d = _bytes_arr(data)
    cap = int(d.size) * 8 + 4096
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_compress_with_header(d.ctypes.data, d.size, out.ctypes.data, cap)
    if n < 0:
        raise ValueError(f"compress failed rc={n}")
    return out[:n].tobytes()
