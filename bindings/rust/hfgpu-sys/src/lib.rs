//! Declarations of the C ABI of `libhfgpu.so`, one for one with
//! `include/hfgpu.h` (a test of the repository compares the two name lists).
//! `usize` of the crate's public API is `u64` here; `[usize; 256]` tables are
//! passed as `*const u64`.  Pointers named `d_*` are device pointers.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct hf_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct hf_model {
    _private: [u8; 0],
}

pub const HF_OK: c_int = 0;
pub const HF_ERR_BAD_ARG: c_int = 1;
pub const HF_ERR_CODELEN: c_int = 2; // -> io::ErrorKind::InvalidData (encode.rs:102-104)
pub const HF_ERR_EMPTY: c_int = 3;
pub const HF_ERR_CAPACITY: c_int = 4;
pub const HF_ERR_HIP: c_int = 5;
pub const HF_ERR_BAD_CODEBOOK: c_int = 6;
pub const HF_ERR_DECODE: c_int = 7;
pub const HF_ERR_INTERNAL: c_int = 8;
pub const HF_ERR_UNSUPPORTED: c_int = 9;

pub const HF_RES_TOTAL_BITS: usize = 0;
pub const HF_RES_FLAGS: usize = 1;
pub const HF_RES_END_BIT: usize = 2;
pub const HF_RES_WORDS: usize = 4;
pub const HF_FLAG_CAPACITY: u64 = 1;
pub const HF_FLAG_DECODE: u64 = 2;
pub const HF_FLAG_TIMEOUT: u64 = 4;
pub const HF_FLAG_PLAN: u64 = 8;
pub const HF_SEG_CHUNKS: u64 = 256;
pub const HF_PLAN_BATCH: u64 = 65536;
pub const HF_WRITE_VALUE_MAX: u64 = 1 << 20;
pub const HF_NOALLOC_MAX_SYMBOLS: u64 = 1 << 32;
pub const HF_TIMED_HISTOGRAM: c_int = 0;
pub const HF_TIMED_ENCODE: c_int = 1;
pub const HF_TIMED_DECODE: c_int = 2;
pub const HF_TIMED_GAP_SYNC: c_int = 3;
pub const HF_TIMED_GAP_WALK: c_int = 4;

extern "C" {
    // ---- context
    pub fn hf_version() -> c_int;
    pub fn hf_strerror(status: c_int) -> *const c_char;
    pub fn hf_ctx_create(device: c_int, ctx: *mut *mut hf_ctx) -> c_int;
    pub fn hf_ctx_destroy(ctx: *mut hf_ctx) -> c_int;
    pub fn hf_ctx_set_stream(ctx: *mut hf_ctx, stream: *mut c_void) -> c_int;
    pub fn hf_ctx_synchronize(ctx: *mut hf_ctx) -> c_int;
    pub fn hf_last_error(ctx: *const hf_ctx) -> *const c_char;
    pub fn hf_ctx_set_option(ctx: *mut hf_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn hf_ctx_kernel_ms(ctx: *mut hf_ctx, kind: c_int, sum_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn hf_ctx_encode_kernel(ctx: *const hf_ctx) -> *const c_char;
    pub fn hf_ctx_decode_kernel(ctx: *const hf_ctx) -> *const c_char;

    // ---- src/stats.rs
    pub fn hf_histogram_async(ctx: *mut hf_ctx, d_in: *const u8, n: u64, d_hist: *mut u64) -> c_int;
    pub fn hf_histogram(ctx: *mut hf_ctx, d_in: *const u8, n: u64, hist: *mut u64) -> c_int;
    pub fn hf_batch_hist_bytes(n: u64) -> u64;
    pub fn hf_histogram_batches_async(ctx: *mut hf_ctx, d_in: *const u8, n: u64, d_hist: *mut u64,
                                      d_batch_hist: *mut u32) -> c_int;

    // ---- src/huffman/mod.rs (host arithmetic)
    pub fn hf_codeword_lengths_inplace(weights: *mut u64, n: u64) -> c_int;
    pub fn hf_sort_by_value(store: *const u64, n: u64, idx: *mut u64, cnt: *mut u64, m: *mut u64) -> c_int;
    pub fn hf_codewords_from_lengths(lengths: *const u64, n: u64, small_codes: *mut u64,
                                     big_codes: *mut u64) -> c_int;
    pub fn hf_build_codebook(hist: *const u64, n: u64, codes: *mut u64, lens: *mut u64) -> c_int;
    pub fn hf_model_sentinel(lens: *const u64) -> u64;
    pub fn hf_model_to_table(codes: *const u64, lens: *const u64, keys: *mut u64, syms: *mut u8,
                             klens: *mut u8, m: *mut u64) -> c_int;
    pub fn hf_calculate_length(val: u64) -> u64;
    pub fn hf_model_create(ctx: *mut hf_ctx, codes: *const u64, lens: *const u64,
                           model: *mut *mut hf_model) -> c_int;
    pub fn hf_model_from_table(ctx: *mut hf_ctx, keys: *const u64, syms: *const u8, klens: *const u8,
                               m: u64, sentinel: u64, model: *mut *mut hf_model) -> c_int;
    pub fn hf_model_destroy(model: *mut hf_model) -> c_int;

    // ---- src/huffman/encode.rs
    pub fn hf_encode_bound(model: *const hf_model, n: u64, lead_bits: u32) -> u64;
    pub fn hf_gap_layout(n: u64, chunk_syms: u32, n_chunks: *mut u64, n_segs: *mut u64) -> c_int;
    pub fn hf_encode_async(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, n: u64,
                           lead_bits: u32, d_out: *mut u8, out_cap: u64, chunk_syms: u32,
                           d_chunk_bits: *mut u16, d_seg_off: *mut u64, d_result: *mut u64) -> c_int;
    pub fn hf_encode_planned_async(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, n: u64,
                                   lead_bits: u32, d_batch_hist: *const u32, d_out: *mut u8,
                                   out_cap: u64, chunk_syms: u32, d_chunk_bits: *mut u16,
                                   d_seg_off: *mut u64, d_result: *mut u64) -> c_int;
    pub fn hf_encode_noalloc_async(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, n: u64,
                                   lead_bits: u32, d_batch_hist: *const u32, d_out: *mut u8,
                                   out_cap: u64, chunk_syms: u32, d_chunk_bits: *mut u16,
                                   d_seg_off: *mut u64, d_result: *mut u64) -> c_int;
    pub fn hf_encode(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, n: u64, lead_bits: u32,
                     d_out: *mut u8, out_cap: u64, chunk_syms: u32, d_chunk_bits: *mut u16,
                     d_seg_off: *mut u64, total_bits: *mut u64) -> c_int;
    pub fn hf_encoder_write_value(ctx: *mut hf_ctx, d_in: *const u8, n: u64, codes: *const u64,
                                  lens: *const u64, state: *mut u64, ret: *mut u64) -> c_int;

    // ---- src/huffman/decode/mod.rs
    pub fn hf_decode_async(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, in_bytes: u64,
                           bit_base: u64, n_out: u64, chunk_syms: u32, d_chunk_bits: *const u16,
                           d_seg_off: *const u64, d_out: *mut u8, d_result: *mut u64) -> c_int;
    pub fn hf_decode(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, in_bytes: u64,
                     bit_base: u64, n_out: u64, chunk_syms: u32, d_chunk_bits: *const u16,
                     d_seg_off: *const u64, d_out: *mut u8) -> c_int;
    pub fn hf_decode_serial(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, in_bytes: u64,
                            bit_base: u64, n_out: u64, d_out: *mut u8) -> c_int;
    pub fn hf_build_gap_array(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, in_bytes: u64,
                              n_out: u64, chunk_syms: u32, d_chunk_bits: *mut u16,
                              d_seg_off: *mut u64) -> c_int;
    pub fn hf_decode_bare(ctx: *mut hf_ctx, model: *const hf_model, d_in: *const u8, in_bytes: u64,
                          n_out: u64, d_out: *mut u8) -> c_int;

    // ---- one stream over several GPUs
    pub fn hf_shard_plan(hists: *const u64, lens: *const u64, world: u32, bits: *mut u64,
                         start_bit: *mut u64, lead_bits: *mut u32, byte_first: *mut u64,
                         byte_last: *mut u64, total_bits: *mut u64) -> c_int;
    pub fn hf_place_shard(whole: *mut u8, whole_cap: u64, shard: *const u8, start_bit: u64,
                          shard_bits: u64) -> c_int;

    // ---- host-buffer forms
    pub fn hf_histogram_host(ctx: *mut hf_ctx, h_in: *const u8, n: u64, hist: *mut u64) -> c_int;
    pub fn hf_encode_host(ctx: *mut hf_ctx, model: *const hf_model, h_in: *const u8, n: u64,
                          lead_bits: u32, h_out: *mut u8, out_cap: u64, chunk_syms: u32,
                          h_chunk_bits: *mut u16, h_seg_off: *mut u64, total_bits: *mut u64) -> c_int;
    pub fn hf_compress_host(ctx: *mut hf_ctx, h_in: *const u8, n: u64, codes: *mut u64, lens: *mut u64,
                            h_out: *mut u8, out_cap: u64, chunk_syms: u32, h_chunk_bits: *mut u16,
                            h_seg_off: *mut u64, total_bits: *mut u64) -> c_int;
    pub fn hf_decode_host(ctx: *mut hf_ctx, model: *const hf_model, h_in: *const u8, in_bytes: u64,
                          n_out: u64, chunk_syms: u32, h_chunk_bits: *const u16,
                          h_seg_off: *const u64, h_out: *mut u8) -> c_int;
    pub fn hf_decode_range_host(ctx: *mut hf_ctx, model: *const hf_model, h_win: *const u8,
                                win_bytes: u64, bit_base: u64, n_out: u64, chunk_syms: u32,
                                h_chunk_bits: *const u16, h_seg_off: *const u64,
                                h_out: *mut u8) -> c_int;
    pub fn hf_decode_window_host(ctx: *mut hf_ctx, model: *const hf_model, h_win: *const u8,
                                 win_bytes: u64, start_bit: u32, n_max: u64, last: c_int,
                                 h_out: *mut u8, n_decoded: *mut u64, end_bit: *mut u64) -> c_int;
    pub fn hf_encoder_write_value_host(ctx: *mut hf_ctx, h_in: *const u8, n: u64, codes: *const u64,
                                       lens: *const u64, state: *mut u64, ret: *mut u64) -> c_int;
}
