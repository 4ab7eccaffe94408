"""Device-resident codec API: torch tensors in HBM in, torch tensors out.

PyTorch is used for device memory and streams only; all compute is in
libhfgpu.so.  Work is enqueued on torch's current stream of the tensor's
device, so torch.cuda events and synchronisation see it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import HfError, lib, u8p, u64p

DEFAULT_CHUNK_SYMS = 64


def _np_u64(a) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    if a.shape != (256,):
        raise ValueError("expected 256 entries")
    return a


class Context:
    """hf_ctx for one GPU (include/hfgpu.h)."""

    _cache: dict[int, "Context"] = {}

    def __init__(self, device_index: int):
        if not torch.cuda.is_available():
            raise RuntimeError("huffman_amd needs a GPU: there is no CPU fallback")
        self.device_index = device_index
        self.device = torch.device("cuda", device_index)
        h = C.c_void_p()
        rc = lib().hf_ctx_create(device_index, C.byref(h))
        if rc != _lib.HF_OK:
            raise HfError(rc, "hf_ctx_create")
        self.handle = h

    @classmethod
    def get(cls, device=None) -> "Context":
        if device is None:
            idx = torch.cuda.current_device()
        else:
            d = torch.device(device)
            idx = d.index if d.index is not None else torch.cuda.current_device()
        ctx = cls._cache.get(idx)
        if ctx is None:
            ctx = cls._cache[idx] = Context(idx)
        return ctx

    def bind_stream(self) -> None:
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.check(lib().hf_ctx_set_stream(self.handle, C.c_void_p(s)))

    def check(self, rc: int) -> None:
        if rc != _lib.HF_OK:
            raise HfError(rc, lib().hf_last_error(self.handle).decode())

    def set_option(self, name: str, value: int) -> None:
        self.check(lib().hf_ctx_set_option(self.handle, name.encode(), int(value)))

    def synchronize(self) -> None:
        self.check(lib().hf_ctx_synchronize(self.handle))

    def kernel_ms(self, kind: int) -> tuple[float, int]:
        """Summed HIP-event duration and launch count of the dominant kernel of the
        histogram (0), encode (1) or decode (2) calls since the last collection, or
        of the bare-stream calls' gap_sync_kernel (3) and last walk (4)
        (option "kernel_timing" must be on)."""
        ms = C.c_double()
        cnt = C.c_uint64()
        self.check(lib().hf_ctx_kernel_ms(self.handle, kind, C.byref(ms), C.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def encode_kernel(self) -> str:
        """Name of the kernel the last encode call launched over its complete batches."""
        return (lib().hf_ctx_encode_kernel(self.handle) or b"").decode()

    def decode_kernel(self) -> str:
        """Name of the kernel the last decode call launched."""
        return (lib().hf_ctx_decode_kernel(self.handle) or b"").decode()


def _check_u8(t: torch.Tensor) -> torch.Tensor:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8):
        raise TypeError("expected a uint8 CUDA tensor")
    return t if t.is_contiguous() else t.contiguous()


def histogram_async(t: torch.Tensor, out: torch.Tensor | None = None,
                    batch_hist: torch.Tensor | None = None) -> torch.Tensor:
    """256-bin byte histogram, left in device memory (int64[256]).  With
    batch_hist (see new_batch_hist) the pass also records one histogram per
    65536 symbols, which encode_async(..., plan=batch_hist) turns into the
    place of every batch in the stream."""
    t = _check_u8(t)
    ctx = Context.get(t.device)
    ctx.bind_stream()
    if out is None:
        out = torch.empty(256, dtype=torch.int64, device=t.device)
    if batch_hist is None:
        ctx.check(lib().hf_histogram_async(ctx.handle, t.data_ptr(), t.numel(), out.data_ptr()))
    else:
        if batch_hist.numel() * batch_hist.element_size() < batch_hist_bytes(t.numel()):
            raise ValueError("batch_hist is too small for this input")
        ctx.check(lib().hf_histogram_batches_async(ctx.handle, t.data_ptr(), t.numel(),
                                                   out.data_ptr(), batch_hist.data_ptr()))
    return out


def batch_hist_bytes(n: int) -> int:
    return int(lib().hf_batch_hist_bytes(n))


def new_batch_hist(n: int, device) -> torch.Tensor:
    """Device buffer for the per-batch histograms of an n-symbol input."""
    return torch.empty(max(batch_hist_bytes(n) // 4, 4), dtype=torch.int32, device=device)


def histogram(t: torch.Tensor) -> np.ndarray:
    t = _check_u8(t)
    ctx = Context.get(t.device)
    ctx.bind_stream()
    h = np.zeros(256, dtype=np.uint64)
    ctx.check(lib().hf_histogram(ctx.handle, t.data_ptr(), t.numel(), h.ctypes.data_as(u64p)))
    return h


def build_codebook(hist) -> tuple[np.ndarray, np.ndarray]:
    """generate_extended_codewords on the host (src/huffman/mod.rs:181-206)."""
    h = np.ascontiguousarray(np.asarray(hist, dtype=np.uint64))
    codes = np.zeros(256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint64)
    rc = lib().hf_build_codebook(h.ctypes.data_as(u64p), h.size, codes.ctypes.data_as(u64p),
                                 lens.ctypes.data_as(u64p))
    if rc != _lib.HF_OK:
        raise HfError(rc, "hf_build_codebook")
    return codes, lens


class DeviceModel:
    """hf_model: codebook + decode tables staged on one GPU."""

    def __init__(self, codes=None, lens=None, *, table=None, device=None):
        self.ctx = Context.get(device)
        self.ctx.bind_stream()  # the tables are uploaded on the current stream
        h = C.c_void_p()
        if table is None:
            self.codes = _np_u64(codes)
            self.lens = _np_u64(lens)
            rc = lib().hf_model_create(self.ctx.handle, self.codes.ctypes.data_as(u64p),
                                       self.lens.ctypes.data_as(u64p), C.byref(h))
            self.sentinel = int(self.lens.max())
        else:
            keys, syms, klens, sentinel = table
            keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
            syms = np.ascontiguousarray(np.asarray(syms, dtype=np.uint8))
            klens = np.ascontiguousarray(np.asarray(klens, dtype=np.uint8))
            self.codes = self.lens = None
            rc = lib().hf_model_from_table(self.ctx.handle, keys.ctypes.data_as(u64p),
                                           syms.ctypes.data_as(u8p), klens.ctypes.data_as(u8p),
                                           keys.size, int(sentinel), C.byref(h))
            self.sentinel = int(sentinel)
        self.ctx.check(rc)
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                lib().hf_model_destroy(h)
            except Exception:
                pass
            self.handle = None

    def encode_bound(self, n: int, lead_bits: int = 0) -> int:
        return int(lib().hf_encode_bound(self.handle, n, lead_bits))


def gap_layout(n: int, chunk_syms: int) -> tuple[int, int]:
    nc = C.c_uint64()
    ns = C.c_uint64()
    rc = lib().hf_gap_layout(n, chunk_syms, C.byref(nc), C.byref(ns))
    if rc != _lib.HF_OK:
        raise HfError(rc, "hf_gap_layout")
    return int(nc.value), int(ns.value)


@dataclass
class Encoded:
    """A compressed stream in HBM plus its gap array (side metadata)."""

    data: torch.Tensor            # uint8; only the first nbytes are the stream
    total_bits: int               # lead_bits excluded
    lead_bits: int
    n: int                        # symbols encoded
    chunk_syms: int
    chunk_bits: torch.Tensor | None   # int16 storage of uint16 chunk bit lengths
    seg_off: torch.Tensor | None      # int64 bit offset of every 256th chunk

    @property
    def nbytes(self) -> int:
        return (self.lead_bits + self.total_bits + 7) // 8

    def stream(self) -> torch.Tensor:
        return self.data[: self.nbytes]


class EncodeBuffers:
    """Caller-owned output buffers for repeated encodes of one shape."""

    def __init__(self, model: DeviceModel, n: int, chunk_syms: int = DEFAULT_CHUNK_SYMS,
                 lead_bits: int = 0, with_gap: bool = True, out_cap: int | None = None):
        dev = model.ctx.device
        self.n = n
        self.chunk_syms = chunk_syms
        self.cap = out_cap if out_cap is not None else model.encode_bound(n, lead_bits)
        self.data = torch.empty(max(self.cap, 4), dtype=torch.uint8, device=dev)
        self.result = torch.zeros(_lib.HF_RES_WORDS, dtype=torch.int64, device=dev)
        if with_gap:
            nc, ns = gap_layout(n, chunk_syms)
            self.chunk_bits = torch.empty(max(nc, 1), dtype=torch.int16, device=dev)
            self.seg_off = torch.empty(max(ns, 1), dtype=torch.int64, device=dev)
        else:
            self.chunk_bits = self.seg_off = None


def encode_async(t: torch.Tensor, model: DeviceModel, bufs: EncodeBuffers,
                 lead_bits: int = 0, plan: torch.Tensor | None = None) -> None:
    """Enqueue one encode; results land in bufs (bufs.result holds bits/flags).
    plan = the batch histograms histogram_async recorded for this same input."""
    ctx = model.ctx
    ctx.bind_stream()
    cb = bufs.chunk_bits.data_ptr() if bufs.chunk_bits is not None else None
    so = bufs.seg_off.data_ptr() if bufs.seg_off is not None else None
    if plan is None:
        ctx.check(lib().hf_encode_async(ctx.handle, model.handle, t.data_ptr(), t.numel(),
                                        lead_bits, bufs.data.data_ptr(), bufs.cap,
                                        bufs.chunk_syms, cb, so, bufs.result.data_ptr()))
    else:
        ctx.check(lib().hf_encode_planned_async(ctx.handle, model.handle, t.data_ptr(), t.numel(),
                                                lead_bits, plan.data_ptr(), bufs.data.data_ptr(),
                                                bufs.cap, bufs.chunk_syms, cb, so,
                                                bufs.result.data_ptr()))


def encode_noalloc_async(t: torch.Tensor, model: DeviceModel, bufs: EncodeBuffers,
                         lead_bits: int = 0, plan: torch.Tensor | None = None) -> None:
    """encode_async through the entry point that allocates nothing: its scratch is
    a fixed block inside the library (at most HF_NOALLOC_MAX_SYMBOLS symbols;
    calls that overlap in time on one GPU race)."""
    ctx = model.ctx
    ctx.bind_stream()
    cb = bufs.chunk_bits.data_ptr() if bufs.chunk_bits is not None else None
    so = bufs.seg_off.data_ptr() if bufs.seg_off is not None else None
    ctx.check(lib().hf_encode_noalloc_async(ctx.handle, model.handle, t.data_ptr(), t.numel(),
                                            lead_bits, plan.data_ptr() if plan is not None else None,
                                            bufs.data.data_ptr(), bufs.cap, bufs.chunk_syms, cb,
                                            so, bufs.result.data_ptr()))


def encode(t: torch.Tensor, model: DeviceModel, chunk_syms: int = DEFAULT_CHUNK_SYMS,
           lead_bits: int = 0, with_gap: bool = True, out_cap: int | None = None) -> Encoded:
    """Encoder::write + flush over a device-resident slice (synchronous)."""
    t = _check_u8(t)
    ctx = model.ctx
    ctx.bind_stream()
    bufs = EncodeBuffers(model, t.numel(), chunk_syms, lead_bits, with_gap, out_cap)
    total = C.c_uint64()
    cb = bufs.chunk_bits.data_ptr() if with_gap else None
    so = bufs.seg_off.data_ptr() if with_gap else None
    ctx.check(lib().hf_encode(ctx.handle, model.handle, t.data_ptr(), t.numel(), lead_bits,
                              bufs.data.data_ptr(), bufs.cap, chunk_syms, cb, so,
                              C.byref(total)))
    return Encoded(bufs.data, int(total.value), lead_bits, t.numel(), chunk_syms,
                   bufs.chunk_bits, bufs.seg_off)


def decode_async(enc_data: torch.Tensor, in_bytes: int, model: DeviceModel, n_out: int,
                 chunk_syms: int, chunk_bits: torch.Tensor, seg_off: torch.Tensor,
                 out: torch.Tensor, result: torch.Tensor, bit_base: int = 0) -> None:
    ctx = model.ctx
    ctx.bind_stream()
    ctx.check(lib().hf_decode_async(ctx.handle, model.handle, enc_data.data_ptr(), in_bytes,
                                    bit_base, n_out, chunk_syms, chunk_bits.data_ptr(),
                                    seg_off.data_ptr(), out.data_ptr(), result.data_ptr()))


def decode(enc: Encoded, model: DeviceModel, out: torch.Tensor | None = None) -> torch.Tensor:
    """decode::read driven by the gap array (synchronous)."""
    ctx = model.ctx
    ctx.bind_stream()
    if out is None:
        out = torch.empty(enc.n, dtype=torch.uint8, device=ctx.device)
    if enc.chunk_bits is None:
        return decode_serial(enc.data, enc.nbytes, model, enc.n, out)
    ctx.check(lib().hf_decode(ctx.handle, model.handle, enc.data.data_ptr(), enc.nbytes, 0, enc.n,
                              enc.chunk_syms, enc.chunk_bits.data_ptr(), enc.seg_off.data_ptr(),
                              out.data_ptr()))
    return out


def decode_serial(data: torch.Tensor, in_bytes: int, model: DeviceModel, n_out: int,
                  out: torch.Tensor | None = None) -> torch.Tensor:
    """Gap-free decode (one sequential walk on the device; small inputs only)."""
    ctx = model.ctx
    ctx.bind_stream()
    if out is None:
        out = torch.empty(n_out, dtype=torch.uint8, device=ctx.device)
    ctx.check(lib().hf_decode_serial(ctx.handle, model.handle, data.data_ptr(), in_bytes, 0,
                                     n_out, out.data_ptr()))
    return out


def decode_bare(data: torch.Tensor, in_bytes: int, model: DeviceModel, n_out: int,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """decode::read of a stream without gap array (synchronous)."""
    ctx = model.ctx
    ctx.bind_stream()
    if out is None:
        out = torch.empty(n_out, dtype=torch.uint8, device=ctx.device)
    ctx.check(lib().hf_decode_bare(ctx.handle, model.handle, data.data_ptr(), in_bytes, n_out,
                                   out.data_ptr()))
    return out


def build_gap_array(data: torch.Tensor, in_bytes: int, model: DeviceModel, n_out: int,
                    chunk_syms: int = DEFAULT_CHUNK_SYMS) -> tuple[torch.Tensor, torch.Tensor]:
    """Gap array (chunk_bits, seg_off) of a stream that came without one; the
    pair then drives decode_async / decode."""
    ctx = model.ctx
    ctx.bind_stream()
    nc, ns = gap_layout(n_out, chunk_syms)
    chunk_bits = torch.empty(max(nc, 1), dtype=torch.int16, device=ctx.device)
    seg_off = torch.empty(max(ns, 1), dtype=torch.int64, device=ctx.device)
    ctx.check(lib().hf_build_gap_array(ctx.handle, model.handle, data.data_ptr(), in_bytes, n_out,
                                       chunk_syms, chunk_bits.data_ptr(), seg_off.data_ptr()))
    return chunk_bits, seg_off


def encoder_write_value(t: torch.Tensor, codes, lens, state) -> tuple[int, list[int]]:
    """Return value of one Encoder::write call and the updated
    {buffer, remaining_bits, writeout} state (src/huffman/encode.rs:91-116)."""
    t = _check_u8(t)
    ctx = Context.get(t.device)
    ctx.bind_stream()
    c = _np_u64(codes)
    l = _np_u64(lens)
    st = (C.c_uint64 * 3)(*[int(x) for x in state])
    ret = C.c_uint64()
    ctx.check(lib().hf_encoder_write_value(ctx.handle, t.data_ptr(), t.numel(),
                                           c.ctypes.data_as(u64p), l.ctypes.data_as(u64p), st,
                                           C.byref(ret)))
    return int(ret.value), [int(x) for x in st]
