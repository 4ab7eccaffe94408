"""ctypes binding of libhfgpu.so (declared in include/hfgpu.h).

The library is the product: there is no CPU fallback.  Loading fails loudly
when the shared object is missing, and every compute call needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HFGPU_LIB names another build of the library (A/B timing of kernel variants)
LIB_PATH = os.path.abspath(os.environ["HFGPU_LIB"]) if os.environ.get("HFGPU_LIB") else \
    os.path.join(_HERE, "libhfgpu.so")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u64p = C.POINTER(C.c_uint64)

HF_OK = 0
HF_ERR_BAD_ARG = 1
HF_ERR_CODELEN = 2
HF_ERR_EMPTY = 3
HF_ERR_CAPACITY = 4
HF_ERR_HIP = 5
HF_ERR_BAD_CODEBOOK = 6
HF_ERR_DECODE = 7
HF_ERR_INTERNAL = 8
HF_ERR_UNSUPPORTED = 9
HF_RES_WORDS = 4
HF_SEG_CHUNKS = 256
HF_PLAN_BATCH = 65536
HF_NOALLOC_MAX_SYMBOLS = 1 << 32

# every symbol include/hfgpu.h declares, with its signature
SIGNATURES = {
    "hf_version": (C.c_int, []),
    "hf_strerror": (C.c_char_p, [C.c_int]),
    "hf_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "hf_ctx_destroy": (C.c_int, [C.c_void_p]),
    "hf_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hf_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "hf_last_error": (C.c_char_p, [C.c_void_p]),
    "hf_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "hf_ctx_kernel_ms": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), u64p]),
    "hf_ctx_encode_kernel": (C.c_char_p, [C.c_void_p]),
    "hf_ctx_decode_kernel": (C.c_char_p, [C.c_void_p]),
    "hf_histogram_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "hf_histogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "hf_batch_hist_bytes": (C.c_uint64, [C.c_uint64]),
    "hf_histogram_batches_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_void_p]),
    "hf_codeword_lengths_inplace": (C.c_int, [u64p, C.c_uint64]),
    "hf_sort_by_value": (C.c_int, [u64p, C.c_uint64, u64p, u64p, u64p]),
    "hf_codewords_from_lengths": (C.c_int, [u64p, C.c_uint64, u64p, u64p]),
    "hf_build_codebook": (C.c_int, [u64p, C.c_uint64, u64p, u64p]),
    "hf_model_sentinel": (C.c_uint64, [u64p]),
    "hf_model_to_table": (C.c_int, [u64p, u64p, u64p, u8p, u8p, u64p]),
    "hf_calculate_length": (C.c_uint64, [C.c_uint64]),
    "hf_model_create": (C.c_int, [C.c_void_p, u64p, u64p, C.POINTER(C.c_void_p)]),
    "hf_model_from_table": (C.c_int, [C.c_void_p, u64p, u8p, u8p, C.c_uint64, C.c_uint64,
                                      C.POINTER(C.c_void_p)]),
    "hf_model_destroy": (C.c_int, [C.c_void_p]),
    "hf_encode_bound": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "hf_gap_layout": (C.c_int, [C.c_uint64, C.c_uint32, u64p, u64p]),
    "hf_encode_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "hf_encode_planned_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_encode_noalloc_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                            C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, u64p]),
    "hf_encoder_write_value": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p, u64p,
                                         u64p]),
    "hf_decode_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                  C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "hf_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                            C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_decode_serial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_void_p]),
    "hf_decode_bare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "hf_build_gap_array": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                     C.c_uint32, C.c_void_p, C.c_void_p]),
    "hf_shard_plan": (C.c_int, [u64p, u64p, C.c_uint32, u64p, u64p, C.POINTER(C.c_uint32), u64p,
                                u64p, u64p]),
    "hf_place_shard": (C.c_int, [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64]),
    "hf_histogram_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "hf_encode_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, u64p]),
    "hf_compress_host": (C.c_int, [C.c_void_p, u8p, C.c_uint64, u64p, u64p, u8p, C.c_uint64,
                                   C.c_uint32, u16p, u64p, u64p]),
    "hf_decode_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_decode_range_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_decode_window_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.c_uint64, C.c_int, C.c_void_p, u64p, u64p]),
    "hf_encoder_write_value_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p,
                                              u64p, u64p]),
}


class HfError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        name = lib().hf_strerror(status).decode()
        super().__init__(f"hfgpu status {status} ({name})" + (f": {detail}" if detail else ""))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m huffman_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
