/*
 * hfgpu -- MI355X (gfx950) Huffman codec behind a flat C ABI.
 *
 * Drop-in boundary for the hot path of the rscompress-huffman crate
 * (histogram -> host codebook -> encode -> decode).  The reference crate has no
 * FFI of its own; each entry point below names the public Rust item it
 * replaces (paths relative to the reference crate root).  Every `usize` of the
 * reference is a uint64_t here.
 *
 * Conventions
 *   - Every function returns an hf_status (0 = HF_OK); nothing unwinds across
 *     the ABI.  hf_last_error(ctx) gives a human-readable detail string.
 *   - Pointers named d_* are DEVICE pointers (HBM of the context's GPU);
 *     all other pointers are host memory.  Buffers are caller-owned.
 *   - *_async entry points only enqueue work on the context's stream and
 *     report device-side conditions through the d_result block; the variants
 *     without the suffix also synchronise the stream and translate the
 *     device-side flags into a status.
 *   - One context per GPU and per host thread; calls on one context must not
 *     overlap.
 */
#ifndef HFGPU_H
#define HFGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_API __attribute__((visibility("default")))

typedef enum hf_status {
    HF_OK = 0,
    HF_ERR_BAD_ARG = 1,      /* null pointer, misaligned buffer, bad chunk size */
    HF_ERR_CODELEN = 2,      /* a code length above 64: io::ErrorKind::InvalidData at
                                src/huffman/encode.rs:102-104 */
    HF_ERR_EMPTY = 3,        /* empty histogram / empty model: the reference panics at
                                src/huffman/mod.rs:72-73 */
    HF_ERR_CAPACITY = 4,     /* output buffer too small */
    HF_ERR_HIP = 5,          /* a HIP runtime call failed */
    HF_ERR_BAD_CODEBOOK = 6, /* code >= 2^len, or a shift of 64 bits in
                                src/huffman/mod.rs:165-168 */
    HF_ERR_DECODE = 7,       /* bit window below the smallest key: the reference
                                underflows and panics at src/huffman/decode/mod.rs:92,189 */
    HF_ERR_INTERNAL = 8,     /* device-side wait timed out */
    HF_ERR_UNSUPPORTED = 9   /* code length 58..64 (the 64-bit buffer of the reference's
                                put/cleanup automaton loses bits there and emits a damaged
                                stream) or sentinel above 57 */
} hf_status;

typedef struct hf_ctx hf_ctx;     /* one GPU + stream + scratch */
typedef struct hf_model hf_model; /* device-resident codebook + decode tables */

/* Device-side result block written by the *_async calls (4 x uint64_t). */
enum { HF_RES_TOTAL_BITS = 0, HF_RES_FLAGS = 1, HF_RES_END_BIT = 2, HF_RES_WORDS = 4 };
enum {
    HF_FLAG_CAPACITY = 1u,  /* an output store was suppressed: buffer too small */
    HF_FLAG_DECODE = 2u,    /* window below the smallest key */
    HF_FLAG_TIMEOUT = 4u,   /* a bounded inter-workgroup wait expired */
    HF_FLAG_PLAN = 8u       /* batch histograms do not belong to this input */
};

/* Granularity constants of the side metadata (the "gap array"). */
enum { HF_SEG_CHUNKS = 256 }; /* chunks per segment */
/* Symbols per batch of the per-batch histograms (hf_histogram_batches_async). */
enum { HF_PLAN_BATCH = 65536 };

/* ------------------------------------------------------------------ context */

HF_API int hf_version(void);
HF_API const char *hf_strerror(int status);

/* Creates a context on HIP device `device` using the null stream. */
HF_API int hf_ctx_create(int device, hf_ctx **ctx);
HF_API int hf_ctx_destroy(hf_ctx *ctx);
/* `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).
 * A stream must stay alive until the models created or last used while it was
 * bound have been destroyed (hf_model_destroy records an event on it) and the
 * work queued on it through this library has completed. */
HF_API int hf_ctx_set_stream(hf_ctx *ctx, void *stream);
HF_API int hf_ctx_synchronize(hf_ctx *ctx);
HF_API const char *hf_last_error(const hf_ctx *ctx);
/* Knobs: "enc_mode" (1 = single pass with a chained scan between workgroups,
 * the default; 0 = three kernels, no waiting between workgroups),
 * "hist_grid" (workgroups of the histogram kernel, 0 = four per CU),
 * "host_piece_mib" (hf_encode_host and hf_decode_host move larger buffers
 * through the device in pieces of this many MiB, default 64, so that upload,
 * compute and download overlap),
 * "timeout_ms" (bound on the waits of the chained scan; when it expires the
 * call returns HF_ERR_INTERNAL), "debug" (bit mask that skips phases of the
 * encoder for timing experiments -- results are then wrong on purpose). */
HF_API int hf_ctx_set_option(hf_ctx *ctx, const char *name, int64_t value);
/* With the option "kernel_timing" set, every histogram / encode / decode call
 * brackets its dominant kernel (hist_kernel or hist_batches_kernel, the
 * enc_pack_kernel / enc_pack_stream_kernel launch over the complete batches
 * -- with enc_fix_tiles_kernel behind it where that kernel runs --,
 * the decode kernel) with HIP events on the stream it is launched on.  This waits for the context's
 * stream, returns the summed duration and the number of launches since the
 * last collection (at most 4096 are kept) and starts over.  A measuring aid
 * without counterpart in the reference.
 * The bare-stream calls (hf_build_gap_array, hf_decode_bare) bracket each launch of
 * gap_sync_kernel (HF_TIMED_GAP_SYNC) and their last walk, gap_decode_kernel or
 * gap_emit_kernel (HF_TIMED_GAP_WALK). */
enum {
    HF_TIMED_HISTOGRAM = 0,
    HF_TIMED_ENCODE = 1,
    HF_TIMED_DECODE = 2,
    HF_TIMED_GAP_SYNC = 3,
    HF_TIMED_GAP_WALK = 4,
    HF_TIMED_KINDS = 5
};
HF_API int hf_ctx_kernel_ms(hf_ctx *ctx, int kind, double *sum_ms, uint64_t *launches);
/* Name of the kernel the context's last encode call launched over its complete
 * batches ("enc_pack_stream_kernel", "enc_pack_kernel", "enc_bytes_kernel"; ""
 * before the first such call): lets a benchmark label the duration above.  A
 * measuring aid without counterpart in the reference. */
HF_API const char *hf_ctx_encode_kernel(const hf_ctx *ctx);
/* The same for the context's last hf_decode / hf_decode_async call
 * ("dec_kernel_pair2", "dec_kernel_reg (pairs)", "dec_kernel_reg", "dec_kernel"). */
HF_API const char *hf_ctx_decode_kernel(const hf_ctx *ctx);

/* -------------------------------------------------------- L0: src/stats.rs */

/* generate_histogram (src/stats.rs:13-28) and the counting loop of
 * Huffman::from_slice (src/huffman/mod.rs:59-65): 256-bin byte counts of
 * d_in[0..n).  d_hist is 256 x uint64_t in device memory (overwritten). */
HF_API int hf_histogram_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist);
/* Same, then copies the counts to host memory. */
HF_API int hf_histogram(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t hist[256]);
/* The same counting pass (src/huffman/mod.rs:59-65) that keeps what it learns
 * about where the symbols are: besides d_hist it fills d_batch_hist with one
 * histogram per HF_PLAN_BATCH consecutive symbols, ceil(n / HF_PLAN_BATCH) x
 * 256 x uint32_t (hf_batch_hist_bytes(n) bytes, 1.6 % of n).  Given to
 * hf_encode_planned_async together with the same input, they place every
 * batch of the stream in advance, whatever codebook the histogram leads to. */
HF_API uint64_t hf_batch_hist_bytes(uint64_t n);
HF_API int hf_histogram_batches_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n,
                                      uint64_t *d_hist, uint32_t *d_batch_hist);

/* ------------------------------------- L1: src/huffman/mod.rs (host side) */

/* calculate_codeword_length_inplace (src/huffman/mod.rs:71-125): weights in
 * non-increasing order are replaced by code lengths. */
HF_API int hf_codeword_lengths_inplace(uint64_t *weights, uint64_t n);
/* sort_by_value (src/huffman/mod.rs:127-141): (index, count) pairs with
 * count > 0 in descending count order; equal counts keep ascending index order.
 * idx/cnt hold up to 256 entries; *m receives the pair count. */
HF_API int hf_sort_by_value(const uint64_t *store, uint64_t n, uint64_t *idx, uint64_t *cnt,
                            uint64_t *m);
/* calculate_codewords_based_on_length (src/huffman/mod.rs:155-171). */
HF_API int hf_codewords_from_lengths(const uint64_t *lengths, uint64_t n, uint64_t *small_codes,
                                     uint64_t *big_codes);
/* generate_extended_codewords (src/huffman/mod.rs:181-206) == what
 * Huffman::from_histogram (src/huffman/mod.rs:48-51) stores. */
HF_API int hf_build_codebook(const uint64_t *hist, uint64_t n, uint64_t codes[256],
                             uint64_t lens[256]);
/* Model::sentinel (src/huffman/mod.rs:20-26). */
HF_API uint64_t hf_model_sentinel(const uint64_t lens[256]);
/* Model::to_btreemap (src/huffman/mod.rs:27-38) flattened: entries sorted by
 * key = code << (sentinel - len); *m receives the entry count (<= 256). */
HF_API int hf_model_to_table(const uint64_t codes[256], const uint64_t lens[256],
                             uint64_t keys[256], uint8_t syms[256], uint8_t klens[256],
                             uint64_t *m);
/* calculate_length (src/huffman/encode.rs:140-149). */
HF_API uint64_t hf_calculate_length(uint64_t val);

/* Huffman::new (src/huffman/mod.rs:45-47): validates the codebook and stages
 * it on the device (flat symbol -> (code,len) table for encoding, multi-level
 * lookup table + sorted keys for decoding). */
HF_API int hf_model_create(hf_ctx *ctx, const uint64_t codes[256], const uint64_t lens[256],
                           hf_model **model);
/* Decode-only model from a key table, as Decoder::from_header builds it from
 * Header.btree (src/huffman/decode/mod.rs:48-66).  keys must be ascending. */
HF_API int hf_model_from_table(hf_ctx *ctx, const uint64_t *keys, const uint8_t *syms,
                               const uint8_t *klens, uint64_t m, uint64_t sentinel,
                               hf_model **model);
HF_API int hf_model_destroy(hf_model *model);

/* -------------------------------------- L2: src/huffman/encode.rs (device) */

/* Upper bound, in bytes, of the stream hf_encode can produce for n symbols
 * with this model and lead_bits leading pad bits. */
HF_API uint64_t hf_encode_bound(const hf_model *model, uint64_t n, uint32_t lead_bits);
/* Sizes of the side metadata for n symbols: n_chunks = ceil(n / chunk_syms)
 * entries of chunk_bits, n_segs = ceil(n_chunks / HF_SEG_CHUNKS) entries of
 * seg_off.  chunk_syms must be 64, 128 or 256 (a decoding workgroup keeps the
 * output of 256 chunks in LDS). */
HF_API int hf_gap_layout(uint64_t n, uint32_t chunk_syms, uint64_t *n_chunks, uint64_t *n_segs);

/* Encoder::write over the whole slice followed by Encoder::flush
 * (src/huffman/encode.rs:91-136): the codes of d_in[0..n) concatenated MSB
 * first and zero-padded to a whole byte, written to d_out.
 *   lead_bits   0..7: that many zero bits precede the first code (lets a shard
 *               be placed at a global bit offset; 0 for a whole stream).
 *   d_out       4-byte aligned, out_cap bytes.
 *   d_chunk_bits, d_seg_off   optional (both or neither): the gap array.
 *               chunk_bits[c] = bit length of chunk c (chunk_syms symbols);
 *               seg_off[s]    = bit offset (lead_bits included) of the first
 *               symbol of chunk s * HF_SEG_CHUNKS.
 *   d_result    HF_RES_WORDS x uint64_t: total bits (lead_bits excluded) and flags.
 * The stream occupies ceil((lead_bits + total_bits) / 8) bytes. */
HF_API int hf_encode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                           uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap,
                           uint32_t chunk_syms, uint16_t *d_chunk_bits, uint64_t *d_seg_off,
                           uint64_t *d_result);
/* hf_encode_async (src/huffman/encode.rs:91-136) for an input whose batch
 * histograms are at hand (hf_histogram_batches_async over the same d_in, n):
 * same stream, same gap array, but the start bit of every batch comes from
 * the histograms and the code lengths, so the workgroups neither take tickets
 * nor wait for each other.  Histograms that do not belong to the input are
 * detected (a batch's planned and actual bit totals differ): the stream is
 * then invalid and the flags carry HF_FLAG_PLAN (HF_ERR_BAD_ARG from the
 * synchronous callers). */
HF_API int hf_encode_planned_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                                   uint64_t n, uint32_t lead_bits,
                                   const uint32_t *d_batch_hist, uint8_t *d_out,
                                   uint64_t out_cap, uint32_t chunk_syms,
                                   uint16_t *d_chunk_bits, uint64_t *d_seg_off,
                                   uint64_t *d_result);
/* hf_encode_async (d_batch_hist == NULL) or hf_encode_planned_async (d_batch_hist
 * given) for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments, same stream, same gap array,
 * same result block -- but the look-back descriptors / plan arrays live in a
 * fixed block of HF_NOALLOC_SCRATCH_BYTES that is part of the library's code
 * object on every GPU, so the call reaches neither hipMalloc nor hipFree nor
 * any other allocating routine, and it records no kernel timing.
 *   - n above HF_NOALLOC_MAX_SYMBOLS is refused with HF_ERR_UNSUPPORTED.
 *   - Every call on one GPU, from any context of the process, uses that GPU's
 *     one block.  The library orders them: a call on another stream than the
 *     block's last user first makes its stream wait for the event that user
 *     recorded behind its kernels (hipStreamWaitEvent / hipEventRecord on one
 *     event per GPU; neither allocates), so calls on several streams run one
 *     after the other instead of corrupting each other.  Calls on one stream
 *     record the event only.  Inside a stream capture keep to one stream: a
 *     captured stream cannot wait for an event recorded outside the capture.
 *     Two PROCESSES on one GPU each have their own block.  The block needs no
 *     clearing by the caller. */
#define HF_NOALLOC_MAX_SYMBOLS (1ull << 32)
#define HF_NOALLOC_SCRATCH_BYTES ((HF_NOALLOC_MAX_SYMBOLS / 16384 + 4) * 8 + 224)
HF_API int hf_encode_noalloc_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                                   uint64_t n, uint32_t lead_bits, const uint32_t *d_batch_hist,
                                   uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
                                   uint16_t *d_chunk_bits, uint64_t *d_seg_off,
                                   uint64_t *d_result);
/* Synchronous form; *total_bits receives the bit count (lead_bits excluded). */
HF_API int hf_encode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                     uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
                     uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint64_t *total_bits);

/* Return value and counters of one Encoder::write call
 * (src/huffman/encode.rs:91-116).  That value is not the number of input bytes
 * consumed but `bytes emitted during the call + 8 - trailing_zeros(buffer)/8`,
 * which depends on the reference's byte-wise `put` / 5-byte `cleanup` schedule
 * (:57-83).  One device lane replays that schedule over d_in[0..n).
 *   state  in/out {buffer, remaining_bits, writeout}; a fresh encoder is
 *          {0, 64, 0} (src/huffman/encode.rs:39-49).
 *   ret    the value Encoder::write returns.
 * Sequential by nature: limited to HF_WRITE_VALUE_MAX symbols per call. */
#define HF_WRITE_VALUE_MAX (1ull << 20)
HF_API int hf_encoder_write_value(hf_ctx *ctx, const uint8_t *d_in, uint64_t n,
                                  const uint64_t codes[256], const uint64_t lens[256],
                                  uint64_t state[3], uint64_t *ret);

/* -------------------------------- L2: src/huffman/decode/mod.rs (device) */

/* decode::read (src/huffman/decode/mod.rs:167-231) / Decoder::read (:70-137)
 * for chunks [first_chunk, first_chunk + ceil(n_out / chunk_syms)): n_out
 * symbols are written to d_out.  Lookup is the reference's predecessor search
 * over the model's keys, so models that are not prefix free decode as they do
 * in the reference.
 *   d_in, in_bytes   compressed bytes; bit `bit_base` of the stream is bit 0
 *                    of d_in[0] (0 for a whole stream); d_in 4-byte aligned.
 *   d_chunk_bits     entry of chunk first_chunk first.
 *   d_seg_off        entry of segment first_chunk / HF_SEG_CHUNKS first;
 *                    first_chunk must be a multiple of HF_SEG_CHUNKS.
 * Bits past in_bytes read as zero, as the reference's drain loop sees them. */
HF_API int hf_decode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                           uint64_t in_bytes, uint64_t bit_base, uint64_t n_out,
                           uint32_t chunk_syms, const uint16_t *d_chunk_bits,
                           const uint64_t *d_seg_off, uint8_t *d_out, uint64_t *d_result);
HF_API int hf_decode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                     uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
                     const uint16_t *d_chunk_bits, const uint64_t *d_seg_off, uint8_t *d_out);

/* The same decode for a stream that carries no gap array (e.g. one written by
 * the reference): a single sequential walk on the device.  Slow by
 * construction; meant for small inputs and for rebuilding a gap array. */
HF_API int hf_decode_serial(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                            uint64_t in_bytes, uint64_t bit_base, uint64_t n_out,
                            uint8_t *d_out);

/* Gap array of a stream that came without one, e.g. a file the reference wrote
 * (src/lib.rs:31-80 stores no side metadata; its decoders, src/huffman/decode/
 * mod.rs:70-231, walk the stream sequentially).  The device decodes
 * sub-sequences of the stream speculatively in parallel and iterates until
 * every one starts where its predecessor ends -- the sequential decode's own
 * code boundaries, for any model -- then records the bit position of every
 * chunk_syms-th symbol.  Synchronous.  d_chunk_bits / d_seg_off are sized by
 * hf_gap_layout(n_out, chunk_syms) and then drive hf_decode.  HF_ERR_DECODE
 * when the stream holds fewer than n_out symbols (hf_decode_serial continues
 * such a stream with zero bits as the reference does) or a bit window lies
 * below the smallest key. */
HF_API int hf_build_gap_array(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                              uint64_t in_bytes, uint64_t n_out, uint32_t chunk_syms,
                              uint16_t *d_chunk_bits, uint64_t *d_seg_off);

/* decode::read (src/huffman/decode/mod.rs:167-231) of a stream that came
 * without gap array, device buffers: the walks that hf_build_gap_array makes
 * write the symbols themselves in their last pass, so neither a gap array nor
 * a second kernel is needed.  What every decoder of a reference-written file
 * runs (Decoder::from_header, src/huffman/decode/mod.rs:48-66, and the CLI's
 * `d`; hf_decode_host without gap arrays calls the same code).  Synchronous.
 * d_in 4-byte aligned.  Errors as hf_build_gap_array. */
HF_API int hf_decode_bare(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                          uint64_t in_bytes, uint64_t n_out, uint8_t *d_out);

/* ------------------------------------------------ one stream over several GPUs
 *
 * The reference encodes one slice in one process (Huffman::from_slice,
 * src/huffman/mod.rs:59-65, then Encoder::write + flush, src/huffman/encode.rs:
 * 91-136).  Cut into `world` contiguous shards, one per GPU, the same stream
 * comes out when (1) the shards' histograms are summed before the codebook is
 * built and (2) shard r is encoded with lead_bits[r] = start_bit[r] % 8, where
 * start_bit is the exclusive scan of the shards' bit lengths: the shards then
 * concatenate bytewise and only a shared boundary byte is OR-merged.
 *
 * hf_shard_plan: host arithmetic of step (2), identical on every rank.
 *   hists      world x 256 counts, row r = histogram of shard r (as gathered
 *              from the ranks; their column sums are the histogram the
 *              codebook `lens` was built from)
 *   bits[r]        = sum_s hists[r][s] * lens[s]
 *   start_bit[r]   = bits[0] + ... + bits[r-1]
 *   lead_bits[r]   = start_bit[r] % 8        (what rank r passes to hf_encode*)
 *   byte_first[r], byte_last[r]  = [first, last) bytes of the whole stream that
 *              shard r's local stream (lead bits included) covers
 *   *total_bits    = bit length of the whole stream
 * What a binding's Encoder::flush on rank r calls between the histogram
 * exchange and its hf_encode*_async. */
HF_API int hf_shard_plan(const uint64_t *hists, const uint64_t lens[256], uint32_t world,
                         uint64_t *bits, uint64_t *start_bit, uint32_t *lead_bits,
                         uint64_t *byte_first, uint64_t *byte_last, uint64_t *total_bits);
/* Puts shard r's local stream (host memory, lead bits included, as hf_encode*
 * produced it) at its place in the whole stream `whole` (whole_cap bytes, zero
 * initialised, possibly already holding other shards): the bytes it may share
 * with its predecessor and its successor are OR-merged, the rest copied.
 * Shards may be placed in any order. */
HF_API int hf_place_shard(uint8_t *whole, uint64_t whole_cap, const uint8_t *shard,
                          uint64_t start_bit, uint64_t shard_bits);

/* ------------------------------------------------------- host-buffer forms
 *
 * The same operations for callers whose data lives in host memory (what a
 * `&[u8]` / `Vec<u8>` binding passes): the library stages the buffers through
 * a device arena owned by the context (grow only, reused between calls),
 * copies over PCIe, runs the device entry point on the context's stream and
 * synchronises.  hf_encode_host, and hf_decode_host when it is given gap arrays,
 * move buffers above "host_piece_mib" in pieces whose upload, encode / decode
 * and download overlap (results are those of a single call).  Gap arrays are
 * optional; without them hf_decode_host rebuilds them on the device
 * (hf_build_gap_array) and walks sequentially only through streams that end
 * early. */
HF_API int hf_histogram_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t hist[256]);
HF_API int hf_encode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t n,
                          uint32_t lead_bits, uint8_t *h_out, uint64_t out_cap,
                          uint32_t chunk_syms, uint16_t *h_chunk_bits, uint64_t *h_seg_off,
                          uint64_t *total_bits);
/* Huffman::from_reader followed by Encoder::write over the same bytes and
 * Encoder::flush -- what stream_compress_with_header_information does with a
 * file between the header and the end (src/lib.rs:36-37 and 68-79) -- for a
 * host slice, with a single upload: counting pass with batch histograms,
 * codebook on the host (returned in codes/lens, src/huffman/mod.rs:181-206),
 * planned encode, download.  out_cap >= n always suffices (a Huffman code
 * needs at most the 8 bits per symbol of the plain bytes); n == 0 is
 * HF_ERR_EMPTY as in the reference (mod.rs:72-73). */
HF_API int hf_compress_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t codes[256],
                            uint64_t lens[256], uint8_t *h_out, uint64_t out_cap,
                            uint32_t chunk_syms, uint16_t *h_chunk_bits, uint64_t *h_seg_off,
                            uint64_t *total_bits);
HF_API int hf_decode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in,
                          uint64_t in_bytes, uint64_t n_out, uint32_t chunk_syms,
                          const uint16_t *h_chunk_bits, const uint64_t *h_seg_off,
                          uint8_t *h_out);
/* Decoder::read (src/huffman/decode/mod.rs:70-137) for a reader that is drained
 * piece by piece, so that the host never holds more than a window of the
 * stream and a piece of the output.
 *
 * hf_decode_range_host: a range of whole segments of a stream whose gap array
 * is at hand.  h_win holds the stream from bit `bit_base` (a multiple of 32;
 * bit 0 of h_win[0]) at least up to the end of the range plus 16 bytes of
 * run-ahead (or to the stream's end); h_chunk_bits / h_seg_off start at the
 * range's first chunk / segment, seg_off values stay absolute.
 *
 * hf_decode_window_host: the same without a gap array (a file the reference
 * wrote): the window's first code starts at bit `start_bit` of h_win[0].  With
 * last == 0 the device lays out and decodes as many whole segments of 64 x
 * HF_SEG_CHUNKS symbols as lie safely inside the window (codes that start in
 * its last 2048 bits are left for the next window), at most n_max symbols;
 * with last != 0 the window reaches the stream's end and exactly n_max symbols
 * are decoded (a stream that ends early is continued with zero bits as in
 * :219-229, for up to HF_WRITE_VALUE_MAX symbols).  *n_decoded symbols land in
 * h_out; *end_bit is the bit of the window, counted from bit 0 of h_win[0], at
 * which the next symbol starts. */
HF_API int hf_decode_range_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_win,
                                uint64_t win_bytes, uint64_t bit_base, uint64_t n_out,
                                uint32_t chunk_syms, const uint16_t *h_chunk_bits,
                                const uint64_t *h_seg_off, uint8_t *h_out);
HF_API int hf_decode_window_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_win,
                                 uint64_t win_bytes, uint32_t start_bit, uint64_t n_max, int last,
                                 uint8_t *h_out, uint64_t *n_decoded, uint64_t *end_bit);
HF_API int hf_encoder_write_value_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n,
                                       const uint64_t codes[256], const uint64_t lens[256],
                                       uint64_t state[3], uint64_t *ret);

#ifdef __cplusplus
}
#endif
#endif /* HFGPU_H */
