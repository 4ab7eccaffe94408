"""MI355X-native Huffman codec behind the public API of rscompress-huffman.

Layout mirrors the reference crate (src/lib.rs:23-26):
  huffman_amd.stats            <- src/stats.rs
  huffman_amd.model            <- src/model.rs
  huffman_amd.huffman          <- src/huffman/mod.rs (+ encode, decode, header)
  huffman_amd.device           device-resident (torch tensor) entry points
All compute runs in libhfgpu.so (HIP, gfx950); there is no CPU fallback.
"""
BUF = 4096  # src/lib.rs:23

__all__ = ["BUF"]
