"""Host-side pieces of libhfgpu.so (no GPU needed): the library loads, exports
every symbol the header declares, and its codebook construction agrees with the
oracle and with the reference's known answers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from huffman_amd import _lib
from huffman_amd._lib import lib, u8p, u64p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def p64(a):
    return a.ctypes.data_as(u64p)


def build_codebook(hist):
    h = np.ascontiguousarray(np.asarray(hist, dtype=np.uint64))
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    rc = lib().hf_build_codebook(p64(h), h.size, p64(codes), p64(lens))
    return rc, codes, lens


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hfgpu.h")).read()
    declared = set(re.findall(r"HF_API\s+[\w\s\*]+?\b(hf_\w+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.hf_version() >= 1
    assert L.hf_strerror(0) == b"ok"


def test_rust_sys_crate_declares_the_same_functions():
    """bindings/rust/hfgpu-sys/src/lib.rs cannot be compiled in this image (no
    Rust toolchain): at least its list of functions and their parameter counts
    are kept in step with include/hfgpu.h."""
    header = open(os.path.join(ROOT, "include", "hfgpu.h")).read()
    rust = open(os.path.join(ROOT, "bindings", "rust", "hfgpu-sys", "src", "lib.rs")).read()
    c_decl = {m.group(1): m.group(2) for m in
              re.finditer(r"HF_API\s+[\w\s\*]+?\b(hf_\w+)\s*\(([^;]*?)\)\s*;", header, re.S)}
    r_decl = {m.group(1): m.group(2) for m in
              re.finditer(r"pub fn (hf_\w+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", rust, re.S)}
    assert set(c_decl) == set(r_decl), set(c_decl) ^ set(r_decl)

    def arity(params: str) -> int:
        params = params.strip()
        return 0 if params in ("", "void") else params.count(",") + 1

    for name in c_decl:
        assert arity(c_decl[name]) == arity(r_decl[name]), name


def test_lengths_kats(kats):
    for case in kats["codeword_lengths_inplace"]["cases"]:
        w = np.array(case["weights"], dtype=np.uint64)
        assert lib().hf_codeword_lengths_inplace(p64(w), w.size) == 0
        assert list(w) == case["lengths"]


def test_lengths_edge_cases():
    w = np.array([5], dtype=np.uint64)
    assert lib().hf_codeword_lengths_inplace(p64(w), 1) == 0 and w[0] == 0
    w = np.array([5, 3], dtype=np.uint64)
    assert lib().hf_codeword_lengths_inplace(p64(w), 2) == 0 and list(w) == [1, 1]
    assert lib().hf_codeword_lengths_inplace(p64(w), 0) == _lib.HF_ERR_EMPTY


def test_codebook_kat(kats):
    k = kats["extended_codewords_sorted_input"]
    rc, codes, lens = build_codebook(k["histogram"])
    assert rc == 0
    for sym, code in k["exact"].items():
        assert codes[int(sym)] == code
    for sym, (lo, hi) in k["ranges"].items():
        assert lo <= codes[int(sym)] <= hi
    d = k["derived_with_ascending_symbol_tie_rule"]
    assert [int(x) for x in codes[:10]] == d["codes"]
    assert [int(x) for x in lens[:10]] == d["lengths"]


def test_calculate_length(kats):
    for v, expected in kats["calculate_length"]["cases"]:
        assert lib().hf_calculate_length(v) == expected


def test_empty_and_single_symbol():
    rc, _, _ = build_codebook(np.zeros(256, dtype=np.uint64))
    assert rc == _lib.HF_ERR_EMPTY
    h = np.zeros(256, dtype=np.uint64)
    h[97] = 6
    rc, codes, lens = build_codebook(h)
    assert rc == 0 and lens[97] == 1 and codes[97] == 0 and lens.sum() == 1


def random_histograms(rng, count):
    for i in range(count):
        kind = i % 6
        nsym = int(rng.integers(1, 257))
        h = np.zeros(256, dtype=np.uint64)
        idx = rng.choice(256, size=nsym, replace=False)
        if kind == 0:
            h[idx] = rng.integers(1, 1000, size=nsym)
        elif kind == 1:
            h[idx] = rng.integers(1, 4, size=nsym)          # many ties
        elif kind == 2:
            h[idx] = (2.0 ** rng.integers(0, 40, size=nsym)).astype(np.uint64)
        elif kind == 3:
            h[idx] = 1 + rng.zipf(1.3, size=nsym).astype(np.uint64)
        elif kind == 4:
            fib = [1, 1]
            while len(fib) < nsym:
                fib.append(fib[-1] + fib[-2])
            h[idx] = np.array(fib[:nsym], dtype=np.float64).clip(max=2 ** 62).astype(np.uint64)
        else:
            h[idx] = 7                                      # all equal
        yield h


def test_codebook_matches_oracle_on_random_histograms(orc):
    rng = np.random.default_rng(1234)
    checked = 0
    for h in random_histograms(rng, 600):
        try:
            ocodes, olens = orc.generate_extended_codewords(h)
        except ValueError:
            rc, _, _ = build_codebook(h)
            assert rc != 0
            continue
        rc, codes, lens = build_codebook(h)
        assert rc == 0
        assert np.array_equal(lens, olens)
        assert np.array_equal(codes, ocodes)
        checked += 1
    assert checked > 500


def test_sort_and_canonical_codes_match_oracle(orc):
    rng = np.random.default_rng(5)
    for h in random_histograms(rng, 60):
        idx = np.zeros(256, dtype=np.uint64)
        cnt = np.zeros(256, dtype=np.uint64)
        m = C.c_uint64()
        assert lib().hf_sort_by_value(p64(h), 256, p64(idx), p64(cnt), C.byref(m)) == 0
        pairs = orc.sort_by_value(h)
        assert [(int(idx[i]), int(cnt[i])) for i in range(m.value)] == pairs
        lengths = np.array(orc.codeword_lengths_inplace([c for _, c in pairs]), dtype=np.uint64)
        if m.value == 1:
            lengths[0] = 1
        if lengths[-1] > 63:
            continue
        small = np.zeros(m.value, dtype=np.uint64)
        big = np.zeros(m.value, dtype=np.uint64)
        assert lib().hf_codewords_from_lengths(p64(lengths), m.value, p64(small), p64(big)) == 0
        osmall, obig = orc.codewords_from_lengths(lengths)
        assert [int(x) for x in small] == osmall and [int(x) for x in big] == obig


def test_model_table_matches_oracle(orc, kats):
    words = kats["decode_numbers"]["words"]
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    for w in words:
        codes[w] = w
        lens[w] = orc.calculate_length(w)
    keys = np.zeros(256, dtype=np.uint64)
    syms = np.zeros(256, dtype=np.uint8)
    kl = np.zeros(256, dtype=np.uint8)
    m = C.c_uint64()
    rc = lib().hf_model_to_table(p64(codes), p64(lens), p64(keys), syms.ctypes.data_as(u8p),
                                 kl.ctypes.data_as(u8p), C.byref(m))
    assert rc == 0
    okeys, osyms, okl = orc.to_btreemap(codes, lens)
    assert m.value == len(okeys)
    assert np.array_equal(keys[:m.value], okeys)
    assert np.array_equal(syms[:m.value], osyms)
    assert np.array_equal(kl[:m.value], okl)
    assert lib().hf_model_sentinel(p64(lens)) == orc.sentinel(lens) == 8


def test_gap_layout_and_bad_chunk_size():
    nc = C.c_uint64()
    ns = C.c_uint64()
    assert lib().hf_gap_layout(1 << 20, 128, C.byref(nc), C.byref(ns)) == 0
    assert nc.value == 8192 and ns.value == 32
    assert lib().hf_gap_layout(1, 64, C.byref(nc), C.byref(ns)) == 0
    assert nc.value == 1 and ns.value == 1
    assert lib().hf_gap_layout(100, 100, C.byref(nc), C.byref(ns)) == _lib.HF_ERR_BAD_ARG


def test_batch_histogram_size_and_null_context_calls():
    """Host-only parts of the planned-encode and timing entry points."""
    import ctypes as C

    from huffman_amd import _lib

    L = _lib.lib()
    assert L.hf_batch_hist_bytes(0) == 0
    assert L.hf_batch_hist_bytes(1) == 1024
    assert L.hf_batch_hist_bytes(_lib.HF_PLAN_BATCH) == 1024
    assert L.hf_batch_hist_bytes(_lib.HF_PLAN_BATCH + 1) == 2048
    assert L.hf_batch_hist_bytes(1 << 30) == (1 << 30) // 64
    ms, cnt = C.c_double(), C.c_uint64()
    assert L.hf_ctx_kernel_ms(None, 0, C.byref(ms), C.byref(cnt)) == _lib.HF_ERR_BAD_ARG
    assert L.hf_ctx_encode_kernel(None) == b""
    assert L.hf_histogram_batches_async(None, None, 0, None, None) == _lib.HF_ERR_BAD_ARG
    assert L.hf_encode_planned_async(None, None, None, 0, 0, None, None, 0, 64, None, None,
                                     None) == _lib.HF_ERR_BAD_ARG
