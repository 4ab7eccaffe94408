// Decoder kernels (src/huffman/decode/mod.rs:70-231).
// Part of the single translation unit huffman_amd/csrc/hfgpu.hip (device code
// for gfx950, wave64); included there, not compiled on its own.
#ifndef HFGPU_KERN_DECODE_HPP
#define HFGPU_KERN_DECODE_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "hfgpu.h"
#include "dev_common.hpp"
#include "hfgpu_internal.h"

namespace {

// ------------------------------------------------------------------- decode
//
// One workgroup decodes one segment = 256 chunks; lane l decodes chunk l
// (chunk_syms symbols) sequentially.  The chunk start bits come from the gap
// array: the segment's absolute start plus a workgroup scan of the 16-bit
// chunk lengths.  The segment's compressed bytes are staged once, with
// coalesced 16-byte loads, byte-swapped into an LDS image of big-endian
// words, and the decoded bytes are collected in a padded LDS image and
// written out with coalesced 16-byte stores, so global memory only sees
// whole-line traffic although every lane walks its own bit stream.
// Symbols are resolved through the LDS-resident multi-level table; the rare
// slots that the table cannot resolve fall back to a predecessor search over
// the sorted keys with the full sentinel-wide window.

constexpr int DEC_THREADS = 256;

struct DecParams {
    const uint8_t *in;
    uint64_t in_bytes;
    uint64_t bit_base;
    uint64_t n_out;
    uint64_t n_chunks;
    uint32_t chunk_syms;
    const uint16_t *chunk_bits;
    const uint64_t *seg_off;
    uint8_t *out;
    uint64_t *result;
    const uint16_t *lut;
    uint32_t lut_entries;
    uint32_t w1;
    uint32_t sentinel;
    const uint64_t *keys;
    const uint8_t *ksyms;
    const uint8_t *klens;
    uint32_t m;
    const uint32_t *pair;  // two-symbol table over the first-level index (PAIR kernels), or null
    uint32_t pair_entries; // (1 << w1) + second-level tables (HF_PAIR_W2 bits each) when pair is set
    uint32_t istg_words;   // LDS words reserved for the stream image (padding slots included)
    uint32_t img_cap;      // dec_kernel_reg: stream words the image holds
    uint32_t in_aligned16;
    uint32_t out_aligned16;
    uint64_t *trace; // HF_DEC_TRACE builds: 16 stamps per segment, or null
};

// Big-endian 32-bit word `w` of the compressed buffer; bytes past the end are 0.
__device__ __forceinline__ uint32_t dec_load_word(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                  uint64_t w)
{
    const uint64_t b = w << 2;
    if (b + 4 <= in_bytes)
        return bswap32(*reinterpret_cast<const uint32_t *>(in + b));
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (b + i < in_bytes)
            v |= (uint32_t)in[b + i] << (24 - 8 * i);
    return v;
}

struct LdsWords {
    const uint32_t *w;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return w[i]; }
};
// LDS image with one unused slot after every 32 words, for streams whose
// chunks all have (nearly) the same size -- 16 words for 8-bit codes: without
// the padding all lanes of a wave would start on the same two banks.  Costs two
// instructions per word read, so irregular streams keep the plain layout.
struct SwzWords {
    const uint32_t *w;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return w[i + (i >> 5)]; }
};
struct GlobalWords {
    const uint8_t *in;
    uint64_t in_bytes;
    uint64_t wbase;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const
    {
        return dec_load_word(in, in_bytes, wbase + i);
    }
};

// Largest key <= v among the sorted keys, or -1.
__device__ __forceinline__ int dec_predecessor(const uint64_t *keys, uint32_t m, uint64_t v)
{
    int lo = 0, hi = (int)m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] <= v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo - 1;
}

template <typename Src>
__device__ __forceinline__ uint64_t dec_peek64(const Src &src, uint32_t pos)
{
    const uint32_t w = pos >> 5, o = pos & 31u;
    const uint64_t hi = ((uint64_t)src(w) << 32) | src(w + 1);
    return o ? (hi << o) | ((uint64_t)src(w + 2) >> (32 - o)) : hi;
}

// MSB-first bit reader over 32-bit words.  The word after the ones already
// merged into `buf` is kept in a register (`nxt`), so a refill never waits for
// memory: it merges the register and issues the load of the following word.
template <typename Src>
struct BitReader {
    const Src &src;
    uint64_t buf; // unread bits, left aligned, zeros below
    int32_t bits; // number of valid bits in buf
    uint32_t wp;  // index of the word after nxt
    uint32_t nxt; // prefetched word wp - 1
    __device__ __forceinline__ BitReader(const Src &s, uint32_t pos) : src(s) { seek(pos); }
    __device__ __forceinline__ void seek(uint32_t pos)
    {
        const uint32_t w = pos >> 5, o = pos & 31u;
        buf = (((uint64_t)src(w) << 32) | src(w + 1)) << o;
        bits = 64 - (int32_t)o;
        nxt = src(w + 2);
        wp = w + 3;
    }
    __device__ __forceinline__ uint32_t pos() const { return (wp - 1) * 32u - (uint32_t)bits; }
    __device__ __forceinline__ void refill()
    {
        if (bits <= 32) {
            buf |= (uint64_t)nxt << (32 - bits);
            bits += 32;
            nxt = src(wp);
            ++wp;
        }
    }
};

// Resolves a table entry with the escape bit set: walks the sub-tables, then
// falls back to the predecessor search.  Consumes the symbol's bits itself.
template <typename Src>
__device__ inline uint32_t dec_escape(BitReader<Src> &br, uint32_t e, const uint16_t *s_lut,
                                            const DecParams &p, bool &bad)
{
    const uint32_t hi = (uint32_t)(br.buf >> 32);
    uint32_t used = p.w1;
    while ((e & HF_LUT_ESC) && e < HF_LUT_INVALID) {
        const uint32_t idx = (hi << used) >> (32 - HF_LUT_W2);
        e = s_lut[(1u << p.w1) + ((e >> 8) << HF_LUT_W2) + idx];
        used += HF_LUT_W2;
    }
    if (e < HF_LUT_INVALID) {
        const uint32_t len = HF_LUT_LEN(e);
        br.buf <<= len;
        br.bits -= (int32_t)len;
        return HF_LUT_SYM(e);
    }
    if (e == HF_LUT_SEARCH) {
        const uint32_t pos = br.pos();
        const uint64_t window = dec_peek64(br.src, pos) >> (64 - p.sentinel);
        const int k = dec_predecessor(p.keys, p.m, window);
        uint32_t sym = 0, len = 0;
        if (k >= 0) {
            sym = p.ksyms[k];
            len = p.klens[k];
        } else {
            bad = true;
        }
        br.seek(pos + len);
        br.refill();
        return sym;
    }
    bad = true; // window below the smallest key
    return 0;
}

// One symbol.  Needs >= 32 valid bits (or >= sentinel bits when the model's
// sentinel is <= 16) in the reader.  ESC = the table contains escape entries.
template <typename Src, bool ESC>
__device__ __forceinline__ uint32_t dec_symbol(BitReader<Src> &br, const uint16_t *s_lut,
                                               uint32_t shift, const DecParams &p, bool &bad)
{
    const uint32_t e = s_lut[(uint32_t)(br.buf >> 32) >> shift];
    if (ESC && __builtin_expect(e & HF_LUT_ESC, 0))
        return dec_escape(br, e, s_lut, p, bad);
    const uint32_t len = HF_LUT_LEN(e);
    br.buf <<= len;
    br.bits -= (int32_t)len;
    return HF_LUT_SYM(e);
}

template <typename Src, int SPR, bool ESC>
__device__ __forceinline__ void dec_lane(const Src &src, uint32_t startbit, uint32_t cnt,
                                         const uint16_t *s_lut, const DecParams &p,
                                         uint32_t *orow, bool &bad)
{
    BitReader<Src> br(src, startbit);
    const uint32_t shift = 32 - p.w1;
    uint32_t i = 0;
    for (; i + 4 <= cnt; i += 4) {
        uint32_t pack = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((k % SPR) == 0)
                br.refill();
            pack |= dec_symbol<Src, ESC>(br, s_lut, shift, p, bad) << (8 * k);
        }
        orow[i >> 2] = pack;
    }
    if (i < cnt) {
        uint32_t pack = 0;
        for (uint32_t k = 0; i + k < cnt; ++k) {
            br.refill();
            pack |= dec_symbol<Src, ESC>(br, s_lut, shift, p, bad) << (8 * k);
        }
        orow[i >> 2] = pack;
    }
}

template <int SPR, bool ESC>
__global__ __launch_bounds__(DEC_THREADS) void dec_kernel(DecParams p)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (p.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *istg = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    uint32_t *ostg = istg + p.istg_words;
    const uint32_t cs = p.chunk_syms;
    const uint32_t orow_words = cs / 4 + 1; // +1: lanes start on different banks
    uint32_t *s_wave = ostg + HF_SEG_CHUNKS * orow_words;

    const uint64_t seg = blockIdx.x;
    const uint64_t c = seg * HF_SEG_CHUNKS + threadIdx.x;
    const bool valid = c < p.n_chunks;
    const uint32_t cb = valid ? p.chunk_bits[c] : 0u;
    // table: 16-byte loads (the table starts 16-byte aligned in device memory)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.lut);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < lut_bytes / 16; i += DEC_THREADS)
            dst[i] = src[i];
    }
    uint32_t total;
    const uint32_t excl = block_excl_scan256(cb, s_wave, total);
    const uint64_t seg_start = p.seg_off[seg] - p.bit_base;
    // the image starts at the 16-byte unit that holds the segment's first bit
    const uint64_t ubase = seg_start >> 7;
    const uint32_t s_lo = (uint32_t)(seg_start & 127u);
    const uint32_t nunits = (s_lo + total + 127u) / 128u + 1u; // +1: readers run ahead
    const bool use_lds = nunits * 4u <= p.istg_words;
    if (use_lds) {
        uint4 *dst = reinterpret_cast<uint4 *>(istg);
        const uint64_t whole = p.in_bytes >> 4; // units that lie fully inside the buffer
        if (p.in_aligned16) {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.in);
            for (uint32_t i = threadIdx.x; i < nunits; i += DEC_THREADS) {
                const uint64_t u = ubase + i;
                uint4 v;
                if (u < whole) {
                    v = src[u];
                    v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
                } else {
                    v.x = dec_load_word(p.in, p.in_bytes, u * 4 + 0);
                    v.y = dec_load_word(p.in, p.in_bytes, u * 4 + 1);
                    v.z = dec_load_word(p.in, p.in_bytes, u * 4 + 2);
                    v.w = dec_load_word(p.in, p.in_bytes, u * 4 + 3);
                }
                dst[i] = v;
            }
        } else {
            for (uint32_t i = threadIdx.x; i < nunits * 4u; i += DEC_THREADS)
                istg[i] = dec_load_word(p.in, p.in_bytes, ubase * 4 + i);
        }
    }
    __syncthreads();

    const uint64_t sym0 = c * cs;
    const uint32_t cnt = valid ? (uint32_t)min((uint64_t)cs, p.n_out - sym0) : 0u;
    bool bad = false;
    if (cnt) {
        uint32_t *orow = ostg + threadIdx.x * orow_words;
        if (use_lds) {
            LdsWords src{istg};
            dec_lane<LdsWords, SPR, ESC>(src, s_lo + excl, cnt, s_lut, p, orow, bad);
        } else {
            GlobalWords src{p.in, p.in_bytes, ubase * 4};
            dec_lane<GlobalWords, 1, true>(src, s_lo + excl, cnt, s_lut, p, orow, bad);
        }
    }
    if (bad)
        atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
    __syncthreads();

    const uint64_t obase = seg * HF_SEG_CHUNKS * (uint64_t)cs;
    const uint64_t oleft = p.n_out - obase;
    const uint32_t seg_len = (uint32_t)min(oleft, (uint64_t)HF_SEG_CHUNKS * cs);
    for (uint32_t o = threadIdx.x * 16u; o < seg_len; o += DEC_THREADS * 16u) {
        const uint32_t ch = o / cs, within = o % cs;
        const uint32_t *srcw = ostg + ch * orow_words + within / 4;
        uint4 v;
        v.x = srcw[0];
        v.y = srcw[1];
        v.z = srcw[2];
        v.w = srcw[3];
        if (o + 16u <= seg_len && p.out_aligned16) {
            *reinterpret_cast<uint4 *>(p.out + obase + o) = v;
        } else {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t i = 0; i < 16u && o + i < seg_len; ++i)
                p.out[obase + o + i] = (uint8_t)(vv[i >> 2] >> (8 * (i & 3)));
        }
    }
}

// Escape entry `e` of the first level resolved from a 64-bit window whose
// top `sentinel` bits are valid: walks the sub-tables, then falls back to the
// predecessor search.  Returns a plain entry (sym << 8 | len); len 0 and
// `bad` set when the window lies below the smallest key.
__device__ inline uint32_t dec_escape_window(uint64_t win, uint32_t e, const uint16_t *s_lut,
                                             const DecParams &p, bool &bad)
{
    const uint32_t hi = (uint32_t)(win >> 32);
    uint32_t used = p.w1;
    while ((e & HF_LUT_ESC) && e < HF_LUT_INVALID) {
        const uint32_t idx = (hi << used) >> (32 - HF_LUT_W2);
        e = s_lut[(1u << p.w1) + ((e >> 8) << HF_LUT_W2) + idx];
        used += HF_LUT_W2;
    }
    if (e < HF_LUT_INVALID)
        return e;
    if (e == HF_LUT_SEARCH) {
        const int k = dec_predecessor(p.keys, p.m, win >> (64 - p.sentinel));
        if (k >= 0)
            return HF_LUT_ENTRY(p.ksyms[k], p.klens[k]);
    }
    bad = true;
    return 0;
}

// Completes a two-symbol table entry that is marked slow: the first symbol
// comes from the entry, or from a single lookup (escapes included) when the
// entry could not hold it; the second always from a single lookup behind it.
// Returns the entry the table could not hold: sym0 << 8 | sym1 << 16 | (len0 +
// len1).  Two codes that do not fit the 64-bit window together mark the lane
// dirty instead (their length is reported as 0).
__device__ inline uint32_t dec_pair_slow(uint64_t win, uint32_t e, const uint16_t *s_lut,
                                         const uint32_t *s_pair, const DecParams &p, bool &bad,
                                         bool &dirty)
{
    const uint32_t shift = 32 - p.w1;
    if (e & HF_PAIR_SUB) {
        // the index's second-level table over the next HF_PAIR_W2 bits: nearly always
        // a complete pair (of up to w1 + HF_PAIR_W2 bits)
        const uint32_t next = ((uint32_t)(win >> 32) << p.w1) >> (32 - HF_PAIR_W2);
        e = s_pair[(1u << p.w1) + (((e >> 8) & 0xffu) << HF_PAIR_W2) + next];
        if ((int32_t)e >= 0)
            return e;
    }
    uint32_t sym0 = (e >> 8) & 0xffu, l0 = e & 0x7fu;
    if (e & HF_PAIR_UNRESOLVED) {
        uint32_t e0 = s_lut[(uint32_t)(win >> 32) >> shift];
        if (e0 & HF_LUT_ESC)
            e0 = dec_escape_window(win, e0, s_lut, p, bad);
        sym0 = HF_LUT_SYM(e0), l0 = HF_LUT_LEN(e0);
    }
    win <<= (l0 & 63u);
    uint32_t e1 = s_lut[(uint32_t)(win >> 32) >> shift];
    if (e1 & HF_LUT_ESC)
        e1 = dec_escape_window(win, e1, s_lut, p, bad);
    uint32_t len = l0 + HF_LUT_LEN(e1);
    if (len > 64u) {
        dirty = true;
        len = 0;
    }
    return (sym0 << 8) | (HF_LUT_SYM(e1) << 16) | len;
}

// Register-output form of the decode kernel for 64-symbol chunks.  ESC = the
// model's table has escape entries (codes longer than the first-level index).
// A lane keeps its CS decoded bytes in CS/4 registers (the symbol loop is
// fully unrolled), so the LDS image of the compressed words can be reused for
// the output once every lane has finished reading it: a workgroup needs
// max(input, output) bytes of LDS instead of their sum, which nearly doubles
// the number of resident workgroups -- and decoding is bound by the latency of
// its dependent table lookups, i.e. by the number of chunks in flight.
// Lanes whose chunk is ragged (end of the stream) and segments too dense for
// the image take a rolled loop that stores bytes straight to global memory.
// Built with -DHF_DEC_TRACE (HF_DEC_TRACE=1 python huffman_amd/build.py) and
// given a buffer through the context option "dec_trace", threads 0 and 192 of
// every work-group stamp the wall clock at the phase boundaries of the decode
// kernel (8 stamps each per segment); tools/trace_dec.py prints the timeline.
#ifdef HF_DEC_TRACE
#define DEC_TRACE(i)                                                                          \
    do {                                                                                      \
        if (trp)                                                                              \
            trp[(i)] = wall_clock64();                                                        \
    } while (0)
#else
#define DEC_TRACE(i)                                                                          \
    do {                                                                                      \
    } while (0)
#endif
// PAIR = strongly skewed complete prefix codes (short codes dominate): a second
// table over the same first-level index holds, for almost every index, the TWO
// symbols whose codes start there and their combined length, so one lookup,
// one shift and one byte permute serve two symbols; the few indices whose
// second code does not end inside the index are marked and resolved by two
// single lookups.  Every iteration still yields exactly two symbols, so the
// output registers keep their static indices.
template <int CS, int K, bool ESC, bool SWZ, bool SPEC, bool PAIR = false>
__global__ __launch_bounds__(DEC_THREADS) void dec_kernel_reg(DecParams p)
{
    static_assert(!PAIR || (K % 2 == 0 && SPEC && CS == 64), "pair decoding: even K, checked windows");
    extern __shared__ __align__(16) uint8_t smem[];
    using Img = typename std::conditional<SWZ, SwzWords, LdsWords>::type;
    const uint32_t lut_bytes = (p.lut_entries * 2u + 15u) & ~15u;
    const uint32_t pair_bytes = PAIR ? p.pair_entries * 4u : 0u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    const uint32_t *s_pair = reinterpret_cast<const uint32_t *>(smem + lut_bytes);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes + pair_bytes); // stream image
    uint32_t *s_wave = img + p.istg_words;

    const uint64_t seg = blockIdx.x;
#ifdef HF_DEC_TRACE
    uint64_t *trp = (p.trace && (threadIdx.x == 0 || threadIdx.x == 192))
                        ? p.trace + seg * 16 + (threadIdx.x ? 8 : 0)
                        : nullptr;
#endif
    DEC_TRACE(0);
    // The segment's span of the stream is known from the segment offsets
    // alone (this one's and the next one's), so its first 16 KiB are requested
    // before anything else: one memory latency covers them, the gap array and
    // the table, instead of two in a row.
    const uint64_t c = seg * HF_SEG_CHUNKS + threadIdx.x;
    const bool valid = c < p.n_chunks;
    const uint32_t cb = valid ? p.chunk_bits[c] : 0u;
    const uint4 *lut16 = reinterpret_cast<const uint4 *>(p.lut);
    const uint32_t lut_units = lut_bytes / 16, pair_units = pair_bytes / 16;
    // two units of the table per lane are requested here, unconditionally (what
    // lies beyond 8 KiB of table follows in a loop below); the pair table likewise
    const uint4 lut0 = lut16[min(threadIdx.x, lut_units - 1u)];
    const uint4 lut1 = lut16[min(threadIdx.x + DEC_THREADS, lut_units - 1u)];
    const uint4 *pair16 = reinterpret_cast<const uint4 *>(PAIR ? (const void *)p.pair : (const void *)p.lut);
    const uint32_t pair_top = PAIR ? pair_units - 1u : 0u;
    uint4 pr0 = make_uint4(0, 0, 0, 0), pr1 = pr0;
    if (PAIR) {
        pr0 = pair16[min(threadIdx.x, pair_top)];
        pr1 = pair16[min(threadIdx.x + DEC_THREADS, pair_top)];
    }
    // (straight-line on purpose: a load behind a branch is waited for at once)
    const uint64_t last_seg = (p.n_chunks - 1) / HF_SEG_CHUNKS;
    const uint64_t so0 = p.seg_off[seg], so1 = p.seg_off[min(seg + 1, last_seg)];
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t seg_start = so0 - p.bit_base;
    const uint64_t ubase = seg_start >> 7;
    const uint32_t s_lo = (uint32_t)(seg_start & 127u);
    const uint32_t IMG_CAP = p.img_cap; // words; sized by the host for the stream's density
    constexpr int EARLY = 4;                                  // 16-byte units per lane
    const uint64_t whole = p.in_bytes >> 4;
    const uint4 *in16 = reinterpret_cast<const uint4 *>(p.in);
    uint32_t nunits_e = 0;
    if (seg < last_seg && p.in_aligned16 && so1 - so0 <= (uint64_t)IMG_CAP * 32u) {
        nunits_e = (uint32_t)((s_lo + (so1 - so0) + 127u) / 128u) + 1u;
        if (nunits_e * 4u > IMG_CAP || ubase + nunits_e > whole)
            nunits_e = 0;
    }
    uint4 st[EARLY];
    {
        // unconditional: without a span to fetch, the loads re-read the table
        const uint4 *from = nunits_e ? in16 + ubase : lut16;
        const uint32_t top = nunits_e ? nunits_e - 1u : 0u;
#pragma unroll
        for (int k = 0; k < EARLY; ++k)
            st[k] = from[min(threadIdx.x + k * DEC_THREADS, top)];
    }
    __builtin_amdgcn_sched_barrier(0); // loads return in order: the small ones were issued first
    {
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        if (threadIdx.x < lut_units)
            dst[threadIdx.x] = lut0;
        if (threadIdx.x + DEC_THREADS < lut_units)
            dst[threadIdx.x + DEC_THREADS] = lut1;
        for (uint32_t i = threadIdx.x + 2 * DEC_THREADS; i < lut_units; i += DEC_THREADS)
            dst[i] = lut16[i];
        if (PAIR) {
            uint4 *pdst = reinterpret_cast<uint4 *>(smem + lut_bytes);
            if (threadIdx.x < pair_units)
                pdst[threadIdx.x] = pr0;
            if (threadIdx.x + DEC_THREADS < pair_units)
                pdst[threadIdx.x + DEC_THREADS] = pr1;
            for (uint32_t i = threadIdx.x + 2 * DEC_THREADS; i < pair_units; i += DEC_THREADS)
                pdst[i] = pair16[i];
        }
    }
    uint32_t total;
    const uint32_t excl = block_excl_scan256(cb, s_wave, total);
    DEC_TRACE(1); // gap array and table in, chunk offsets scanned
    const uint32_t nunits = (s_lo + total + 127u) / 128u + 1u; // +1: readers run ahead
    const bool use_lds = nunits * 4u <= IMG_CAP && p.in_aligned16;
    if (use_lds) {
        auto put = [&](uint32_t i, uint4 v) {
            if (SWZ) { // word w of the image lives in slot w + w / 32 (see SwzWords)
                uint32_t *dst = img + 4 * i + ((4 * i) >> 5);
                dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
            } else {
                reinterpret_cast<uint4 *>(img)[i] = v;
            }
        };
        // (a gap array whose chunk lengths exceed the segment's span is
        // inconsistent: everything is then fetched here, as for the last segment)
        const bool early_ok = nunits <= nunits_e;
        if (early_ok) {
#pragma unroll
            for (int k = 0; k < EARLY; ++k) {
                const uint32_t i = threadIdx.x + k * DEC_THREADS;
                if (i < nunits) {
                    uint4 v = st[k];
                    v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
                    put(i, v);
                }
            }
        }
        for (uint32_t i = threadIdx.x + (early_ok ? EARLY * DEC_THREADS : 0); i < nunits;
             i += DEC_THREADS) {
            const uint64_t u = ubase + i;
            uint4 v;
            if (u < whole) {
                v = in16[u];
                v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
            } else {
                v.x = dec_load_word(p.in, p.in_bytes, u * 4 + 0);
                v.y = dec_load_word(p.in, p.in_bytes, u * 4 + 1);
                v.z = dec_load_word(p.in, p.in_bytes, u * 4 + 2);
                v.w = dec_load_word(p.in, p.in_bytes, u * 4 + 3);
            }
            put(i, v);
        }
    }
    DEC_TRACE(2); // stream image staged (this wave's share)
    __syncthreads();
    DEC_TRACE(3);

    const uint64_t sym0 = c * CS;
    const uint32_t cnt = valid ? (uint32_t)min((uint64_t)CS, p.n_out - sym0) : 0u;
    const bool fast = use_lds && cnt == (uint32_t)CS;
    uint32_t outw[CS / 4];
#pragma unroll
    for (int i = 0; i < CS / 4; ++i)
        outw[i] = 0; // (lanes without a whole chunk still take part in the exchange below)
    bool bad = false;
    bool dirty = false; // SPEC: some group of K codes ran past its 64-bit window
    if (fast && K >= 8 && !ESC && !SPEC) {
        // codes of at most 8 bits: a streaming reader that merges one word per
        // 32 consumed bits touches LDS least (one word per four symbols)
        Img src{img};
        BitReader<Img> br(src, s_lo + excl);
        const uint32_t shift = 32 - p.w1;
#pragma unroll
        for (int i = 0; i < CS / 4; ++i) {
            uint32_t pack = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 0)
                    br.refill();
                pack |= dec_symbol<Img, false>(br, s_lut, shift, p, bad) << (8 * k);
            }
            outw[i] = pack;
        }
    } else if (fast && PAIR) {
        // as below, two symbols per lookup
        uint32_t pos = s_lo + excl;
        const uint32_t shift = 32 - p.w1;
#pragma unroll
        for (int i = 0; i < CS / 4; ++i)
            outw[i] = 0;
#pragma unroll
        for (int g = 0; g < CS; g += K) {
            const Img src{img};
            const uint32_t wi = pos >> 5;
            const uint32_t w0 = src(wi), w1 = src(wi + 1), w2 = src(wi + 2);
            const uint32_t sh = 32u - (pos & 31u); // 1..32
            const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
            const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
            uint64_t win = ((uint64_t)hi << 32) | lo;
            // The low byte of the sum of the entries is the sum of their lengths
            // as long as that stays below 256: a pair from the table has at most
            // w1 bits, from a second-level table w1 + 6 (8 x 17), a completed one
            // up to 64.  Completed pairs that take the window past its 64 bits
            // mark the lane dirty on the spot, so the byte is exact (at most
            // 64 + 136) whenever it is used.
            uint32_t esum = 0;
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                const int j = g + k;
                if (j < CS) {
                    uint32_t e = s_pair[(uint32_t)(win >> 32) >> shift];
                    if (__builtin_expect((int32_t)e < 0, 0)) {
                        e = dec_pair_slow(win, e, s_lut, s_pair, p, bad, dirty);
                        dirty = dirty || (esum & 0xffu) + (e & 0x7fu) > 64u;
                    }
                    win <<= (e & 63u); // the entry is its own shift amount
                    esum += e;
                    // bytes j % 4 and j % 4 + 1 of the output word take bytes 1 and 2 of the entry
                    const uint32_t sel = (j & 2) ? 0x06050100u : 0x03020605u;
                    outw[j >> 2] = __builtin_amdgcn_perm(e, outw[j >> 2], sel);
                }
            }
            const uint32_t used = esum & 0xffu;
            pos += used;
            dirty = dirty || used > 64u;
        }
    } else if (fast) {
        // No reader state besides the bit position: every K symbols the lane
        // rebuilds a 64-bit window from the three image words at its position;
        // no refill branches.  Either K * sentinel <= 64, so the window cannot
        // run dry, or (SPEC, complete prefix codes only) the host chose K so that
        // K codes practically never exceed 64 bits; a lane to which it happens
        // anyway is marked dirty and decodes its chunk again at the end.
        uint32_t pos = s_lo + excl;
        const uint32_t shift = 32 - p.w1;
#pragma unroll
        for (int i = 0; i < CS / 4; ++i)
            outw[i] = 0;
#pragma unroll
        for (int g = 0; g < CS; g += K) {
            const Img src{img};
            const uint32_t wi = pos >> 5;
            const uint32_t w0 = src(wi), w1 = src(wi + 1), w2 = src(wi + 2);
            const uint32_t sh = 32u - (pos & 31u); // 1..32
            const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
            const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
            uint64_t win = ((uint64_t)hi << 32) | lo;
            // Entries of tables without escapes have lengths of at most 32 bits:
            // up to seven of them are summed whole and the low byte of the sum is
            // the sum of their lengths.
            constexpr bool BYTESUM = !ESC && K <= 7;
            uint32_t used = 0, esum = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = g + k;
                if (j < CS) {
                    uint32_t e = s_lut[(uint32_t)(win >> 32) >> shift];
                    if (ESC && __builtin_expect((e & HF_LUT_ESC) != 0, 0))
                        e = dec_escape_window(win, e, s_lut, p, bad);
                    win <<= (e & 63u); // the entry is its own shift amount
                    if (BYTESUM)
                        esum += e;
                    else
                        used += HF_LUT_LEN(e);
                    // byte j % 4 of the output word takes byte 1 of the entry
                    constexpr uint32_t keep = 0x03020100u;
                    const uint32_t sel = (keep & ~(0xffu << (8 * (j & 3)))) | (0x05u << (8 * (j & 3)));
                    outw[j >> 2] = __builtin_amdgcn_perm(e, outw[j >> 2], sel);
                }
            }
            if (BYTESUM)
                used = esum & 0xffu;
            pos += used;
            if (SPEC)
                dirty = dirty || used > 64u;
            // One group is scheduled at a time: left free, the scheduler spreads the
            // byte permutes of a group's symbols over the following groups and their
            // entries live through dozens of look-ups.  (A build scheduled that way
            // is how the hardware's trouble with a 64-bit shift whose amount sits in
            // the wave's last allocated VGPR was met -- tools/check_isa.py describes it
            // and checks every build for it; the fence itself costs nothing measurable.)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (cnt) {
        // ragged chunk or oversized segment: bytes go straight to global memory
        GlobalWords src{p.in, p.in_bytes, ubase * 4};
        BitReader<GlobalWords> br(src, s_lo + excl);
        const uint32_t shift = 32 - p.w1;
        for (uint32_t i = 0; i < cnt; ++i) {
            br.refill();
            p.out[sym0 + i] = (uint8_t)dec_symbol<GlobalWords, true>(br, s_lut, shift, p, bad);
        }
    }
    DEC_TRACE(4); // decoded
    if (bad && !dirty) // a dirty lane's symbols (and errors) are not real: it decodes again below
        atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
    if (!use_lds)
        return; // (uniform over the workgroup)
    // Output.  A lane holds the CS bytes of its chunk; stored as they are, one
    // store instruction would touch 64 different lines with 16 bytes each.  The
    // four lanes of a quad exchange their 16-byte units instead (a 4 x 4
    // transpose through DPP quad permutes, no LDS and no barrier): afterwards
    // lane r holds unit r of each of the quad's four chunks, and every store
    // instruction writes 64 contiguous bytes per quad.
    const bool whole_chunk = fast && !dirty;
    if (fast && !p.out_aligned16) {
        // (buffers at odd addresses: the lane stores its own bytes)
        if (!dirty) {
#pragma unroll
            for (int i = 0; i < CS / 4; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p.out[sym0 + 4 * i + q] = (uint8_t)(outw[i] >> (8 * q));
            }
        }
    }
    const unsigned long long whole_mask = __ballot(whole_chunk && p.out_aligned16);
    {
        const uint32_t r = threadIdx.x & 3u;
        const bool odd1 = (r & 1u) != 0, odd2 = (r & 2u) != 0;
        const uint32_t quad0 = threadIdx.x & 60u & 63u; // first lane of the quad within the wave
#pragma unroll
        for (int blk = 0; blk < CS / 64; ++blk) {
            uint32_t *U = outw + 16 * blk; // four units of four dwords
#pragma unroll
            for (int stage = 0; stage < 2; ++stage) {
                const bool odd = stage == 0 ? odd1 : odd2;
                const int step = stage == 0 ? 1 : 2; // partner lane = lane ^ step, partner unit = unit ^ step
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u & step)
                        continue;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        // even lane: keeps a, takes the partner's a; odd lane: takes the
                        // partner's b, keeps b (one permute and one select per result)
                        const uint32_t a = U[4 * u + d], b = U[4 * (u + step) + d];
                        const uint32_t pa = (uint32_t)(stage == 0
                            ? __builtin_amdgcn_update_dpp((int)a, (int)a, 0xB1, 0xf, 0xf, true)   // quad_perm [1,0,3,2]
                            : __builtin_amdgcn_update_dpp((int)a, (int)a, 0x4E, 0xf, 0xf, true)); // quad_perm [2,3,0,1]
                        const uint32_t pb = (uint32_t)(stage == 0
                            ? __builtin_amdgcn_update_dpp((int)b, (int)b, 0xB1, 0xf, 0xf, true)
                            : __builtin_amdgcn_update_dpp((int)b, (int)b, 0x4E, 0xf, 0xf, true));
                        U[4 * u + d] = odd ? pb : a;
                        U[4 * (u + step) + d] = odd ? b : pa;
                    }
                }
            }
            // slot i now holds unit (4 blk + r) of the chunk of quad lane i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((whole_mask >> (quad0 + i)) & 1ull) {
                    const uint64_t ci = c - r + i; // that lane's chunk
                    uint4 v = make_uint4(U[4 * i + 0], U[4 * i + 1], U[4 * i + 2], U[4 * i + 3]);
                    *reinterpret_cast<uint4 *>(p.out + ci * CS + (4u * blk + r) * 16u) = v;
                }
            }
        }
    }
    DEC_TRACE(7); // stored
    if (SPEC) {
        if (dirty) {
            GlobalWords src{p.in, p.in_bytes, ubase * 4};
            BitReader<GlobalWords> br(src, s_lo + excl);
            const uint32_t shift = 32 - p.w1;
            bool bad2 = false;
            for (uint32_t i = 0; i < cnt; ++i) {
                br.refill();
                p.out[sym0 + i] = (uint8_t)dec_symbol<GlobalWords, true>(br, s_lut, shift, p, bad2);
            }
            if (bad2)
                atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS],
                         (unsigned long long)HF_FLAG_DECODE);
        }
    }
}

// Two segments per work-group, two look-up chains per lane (pair tables, 64-symbol
// chunks, plain image, 16-byte aligned buffers): lane l decodes chunk l of both
// segments in one interleaved loop.  The pair decoder is bound by the length of
// its chain of dependent LDS look-ups with all eight wave slots of a SIMD taken;
// a second, independent chain per lane is the parallelism that is left, and the
// tables are staged once for two segments.  Everything else -- window rebuild
// every K codes, completion of pairs the table cannot hold, dirty lanes, quad
// transpose before the stores -- is dec_kernel_reg's, per chain.
__device__ __forceinline__ void dec_store_chunk64(uint32_t (&U)[16], uint64_t c, bool whole, const DecParams &p)
{
    const unsigned long long whole_mask = __ballot(whole);
    const uint32_t r = threadIdx.x & 3u;
    const bool odd1 = (r & 1u) != 0, odd2 = (r & 2u) != 0;
    const uint32_t quad0 = threadIdx.x & 60u;
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        const bool odd = stage == 0 ? odd1 : odd2;
        const int step = stage == 0 ? 1 : 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u & step)
                continue;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // even lane: keeps a, takes the partner's a; odd lane: takes the partner's b,
                // keeps b -- a select each, whose one operand comes through the quad permute
                const uint32_t a = U[4 * u + d], b = U[4 * (u + step) + d];
                const uint32_t pa = (uint32_t)(stage == 0
                    ? __builtin_amdgcn_update_dpp((int)a, (int)a, 0xB1, 0xf, 0xf, true)
                    : __builtin_amdgcn_update_dpp((int)a, (int)a, 0x4E, 0xf, 0xf, true));
                const uint32_t pb = (uint32_t)(stage == 0
                    ? __builtin_amdgcn_update_dpp((int)b, (int)b, 0xB1, 0xf, 0xf, true)
                    : __builtin_amdgcn_update_dpp((int)b, (int)b, 0x4E, 0xf, 0xf, true));
                U[4 * u + d] = odd ? pb : a;
                U[4 * (u + step) + d] = odd ? b : pa;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((whole_mask >> (quad0 + i)) & 1ull) {
            const uint64_t ci = c - r + i;
            *reinterpret_cast<uint4 *>(p.out + ci * 64u + r * 16u) =
                make_uint4(U[4 * i + 0], U[4 * i + 1], U[4 * i + 2], U[4 * i + 3]);
        }
    }
}

template <int K>
__global__ __launch_bounds__(DEC_THREADS) void dec_kernel_pair2(DecParams p)
{
    constexpr int CS = 64;
    static_assert(K % 2 == 0, "pairs");
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (p.lut_entries * 2u + 15u) & ~15u;
    const uint32_t pair_bytes = p.pair_entries * 4u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    const uint32_t *s_pair = reinterpret_cast<const uint32_t *>(smem + lut_bytes);
    uint32_t *img0 = reinterpret_cast<uint32_t *>(smem + lut_bytes + pair_bytes);
    uint32_t *img1 = img0 + p.istg_words;
    uint32_t *s_wave = img1 + p.istg_words; // 8 words

    const uint64_t last_seg = (p.n_chunks - 1) / HF_SEG_CHUNKS;
    const uint64_t segA = 2ull * blockIdx.x;
    const bool haveB = segA + 1 <= last_seg;
    const uint64_t segB = haveB ? segA + 1 : segA;
    const uint64_t cA = segA * HF_SEG_CHUNKS + threadIdx.x, cB = segB * HF_SEG_CHUNKS + threadIdx.x;
    const bool validA = cA < p.n_chunks, validB = haveB && cB < p.n_chunks;
    const uint32_t cbA = validA ? p.chunk_bits[cA] : 0u;
    const uint32_t cbB = validB ? p.chunk_bits[cB] : 0u;
    const uint4 *lut16 = reinterpret_cast<const uint4 *>(p.lut);
    const uint4 *pair16 = reinterpret_cast<const uint4 *>(p.pair);
    const uint32_t lut_units = lut_bytes / 16, pair_units = pair_bytes / 16;
    const uint4 lut0 = lut16[min(threadIdx.x, lut_units - 1u)];
    const uint4 pr0 = pair16[min(threadIdx.x, pair_units - 1u)];
    const uint4 pr1 = pair16[min(threadIdx.x + DEC_THREADS, pair_units - 1u)];
    const uint64_t so0 = p.seg_off[segA], so1 = p.seg_off[segB];
    const uint64_t so2 = p.seg_off[min(segB + 1, last_seg)];
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t IMG_CAP = p.img_cap;
    const uint64_t whole = p.in_bytes >> 4;
    const uint4 *in16 = reinterpret_cast<const uint4 *>(p.in);
    const uint64_t startA = so0 - p.bit_base, startB = so1 - p.bit_base;
    const uint64_t ubA = startA >> 7, ubB = startB >> 7;
    const uint32_t sloA = (uint32_t)(startA & 127u), sloB = (uint32_t)(startB & 127u);
    // the spans are known from consecutive segment offsets: their first units are
    // requested before the scans (two 16-byte units per lane and segment)
    constexpr int EARLY = 2;
    uint32_t neA = 0, neB = 0;
    if (haveB && so1 - so0 <= (uint64_t)IMG_CAP * 32u) {
        neA = (uint32_t)((sloA + (so1 - so0) + 127u) / 128u) + 1u;
        if (neA * 4u > IMG_CAP || ubA + neA > whole)
            neA = 0;
    }
    if (haveB && segB < last_seg && so2 - so1 <= (uint64_t)IMG_CAP * 32u) {
        neB = (uint32_t)((sloB + (so2 - so1) + 127u) / 128u) + 1u;
        if (neB * 4u > IMG_CAP || ubB + neB > whole)
            neB = 0;
    }
    uint4 stA[EARLY], stB[EARLY];
    {
        const uint4 *fromA = neA ? in16 + ubA : lut16, *fromB = neB ? in16 + ubB : lut16;
        const uint32_t topA = neA ? neA - 1u : 0u, topB = neB ? neB - 1u : 0u;
#pragma unroll
        for (int k = 0; k < EARLY; ++k) {
            stA[k] = fromA[min(threadIdx.x + k * DEC_THREADS, topA)];
            stB[k] = fromB[min(threadIdx.x + k * DEC_THREADS, topB)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        if (threadIdx.x < lut_units)
            dst[threadIdx.x] = lut0;
        for (uint32_t i = threadIdx.x + DEC_THREADS; i < lut_units; i += DEC_THREADS)
            dst[i] = lut16[i];
        uint4 *pdst = reinterpret_cast<uint4 *>(smem + lut_bytes);
        if (threadIdx.x < pair_units)
            pdst[threadIdx.x] = pr0;
        if (threadIdx.x + DEC_THREADS < pair_units)
            pdst[threadIdx.x + DEC_THREADS] = pr1;
        for (uint32_t i = threadIdx.x + 2 * DEC_THREADS; i < pair_units; i += DEC_THREADS)
            pdst[i] = pair16[i];
    }
    uint32_t totalA, totalB;
    const uint32_t exclA = block_excl_scan256(cbA, s_wave, totalA);
    const uint32_t exclB = block_excl_scan256(cbB, s_wave + 4, totalB);
    const uint32_t nuA = (sloA + totalA + 127u) / 128u + 1u, nuB = (sloB + totalB + 127u) / 128u + 1u;
    const bool ldsA = nuA * 4u <= IMG_CAP, ldsB = haveB && nuB * 4u <= IMG_CAP;
    auto stage = [&](uint32_t *img, uint64_t ubase, uint32_t nunits, uint32_t n_early, const uint4 (&st)[EARLY]) {
        const bool early_ok = nunits <= n_early;
        if (early_ok) {
#pragma unroll
            for (int k = 0; k < EARLY; ++k) {
                const uint32_t i = threadIdx.x + k * DEC_THREADS;
                if (i < nunits) {
                    uint4 v = st[k];
                    v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
                    reinterpret_cast<uint4 *>(img)[i] = v;
                }
            }
        }
        for (uint32_t i = threadIdx.x + (early_ok ? EARLY * DEC_THREADS : 0); i < nunits; i += DEC_THREADS) {
            const uint64_t u = ubase + i;
            uint4 v;
            if (u < whole) {
                v = in16[u];
                v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
            } else {
                v.x = dec_load_word(p.in, p.in_bytes, u * 4 + 0);
                v.y = dec_load_word(p.in, p.in_bytes, u * 4 + 1);
                v.z = dec_load_word(p.in, p.in_bytes, u * 4 + 2);
                v.w = dec_load_word(p.in, p.in_bytes, u * 4 + 3);
            }
            reinterpret_cast<uint4 *>(img)[i] = v;
        }
    };
    if (ldsA)
        stage(img0, ubA, nuA, neA, stA);
    if (ldsB)
        stage(img1, ubB, nuB, neB, stB);
    __syncthreads();

    const uint64_t symA = cA * CS, symB = cB * CS;
    const uint32_t cntA = validA ? (uint32_t)min((uint64_t)CS, p.n_out - symA) : 0u;
    const uint32_t cntB = validB ? (uint32_t)min((uint64_t)CS, p.n_out - symB) : 0u;
    const bool fastA = ldsA && cntA == (uint32_t)CS, fastB = ldsB && cntB == (uint32_t)CS;
    uint32_t outA[CS / 4], outB[CS / 4];
#pragma unroll
    for (int i = 0; i < CS / 4; ++i)
        outA[i] = 0, outB[i] = 0;
    bool badA = false, badB = false, dirtyA = false, dirtyB = false;
    {
        // (a chain without a whole chunk in the image walks from bit 0 of its image:
        // whatever it reads there, its look-ups stay inside the tables and its
        // results are dropped)
        uint32_t posA = fastA ? sloA + exclA : 0u, posB = fastB ? sloB + exclB : 0u;
        const uint32_t shift = 32 - p.w1;
#pragma unroll
        for (int g = 0; g < CS; g += K) {
            uint64_t winA, winB;
            {
                const uint32_t *w = img0 + (posA >> 5);
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
                const uint32_t sh = 32u - (posA & 31u);
                const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
                const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
                winA = ((uint64_t)hi << 32) | lo;
            }
            {
                const uint32_t *w = img1 + (posB >> 5);
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
                const uint32_t sh = 32u - (posB & 31u);
                const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
                const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
                winB = ((uint64_t)hi << 32) | lo;
            }
            uint32_t esA = 0, esB = 0;
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                const int j = g + k;
                if (j < CS) {
                    uint32_t eA = s_pair[(uint32_t)(winA >> 32) >> shift];
                    uint32_t eB = s_pair[(uint32_t)(winB >> 32) >> shift];
                    if (__builtin_expect((int32_t)eA < 0, 0)) {
                        eA = dec_pair_slow(winA, eA, s_lut, s_pair, p, badA, dirtyA);
                        dirtyA = dirtyA || (esA & 0xffu) + (eA & 0x7fu) > 64u;
                    }
                    if (__builtin_expect((int32_t)eB < 0, 0)) {
                        eB = dec_pair_slow(winB, eB, s_lut, s_pair, p, badB, dirtyB);
                        dirtyB = dirtyB || (esB & 0xffu) + (eB & 0x7fu) > 64u;
                    }
                    winA <<= (eA & 63u);
                    winB <<= (eB & 63u);
                    esA += eA;
                    esB += eB;
                    const uint32_t sel = (j & 2) ? 0x06050100u : 0x03020605u;
                    outA[j >> 2] = __builtin_amdgcn_perm(eA, outA[j >> 2], sel);
                    outB[j >> 2] = __builtin_amdgcn_perm(eB, outB[j >> 2], sel);
                }
            }
            const uint32_t usedA = esA & 0xffu, usedB = esB & 0xffu;
            // (a chain that is not decoding stays inside its image: it starts over)
            posA = fastA ? posA + usedA : 0u;
            posB = fastB ? posB + usedB : 0u;
            dirtyA = dirtyA || usedA > 64u;
            dirtyB = dirtyB || usedB > 64u;
        }
    }
    dirtyA = dirtyA && fastA;
    dirtyB = dirtyB && fastB;
    // ragged chunks and segments too dense for the image: bytes straight to global memory
    auto rolled = [&](uint64_t ubase, uint32_t start, uint64_t sym0, uint32_t cnt, bool &bad) {
        GlobalWords src{p.in, p.in_bytes, ubase * 4};
        BitReader<GlobalWords> br(src, start);
        const uint32_t sh = 32 - p.w1;
        for (uint32_t i = 0; i < cnt; ++i) {
            br.refill();
            p.out[sym0 + i] = (uint8_t)dec_symbol<GlobalWords, true>(br, s_lut, sh, p, bad);
        }
    };
    if (cntA && !fastA) {
        badA = false;
        rolled(ubA, sloA + exclA, symA, cntA, badA);
    }
    if (cntB && !fastB) {
        badB = false;
        rolled(ubB, sloB + exclB, symB, cntB, badB);
    }
    if ((badA && cntA && !dirtyA) || (badB && cntB && !dirtyB))
        atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
    dec_store_chunk64(outA, cA, fastA && !dirtyA, p);
    dec_store_chunk64(outB, cB, fastB && !dirtyB, p);
    if (dirtyA || dirtyB) {
        bool bad2 = false;
        if (dirtyA)
            rolled(ubA, sloA + exclA, symA, cntA, bad2);
        if (dirtyB)
            rolled(ubB, sloB + exclB, symB, cntB, bad2);
        if (bad2)
            atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
    }
}

// Gap-free decode: one lane walks the whole stream with predecessor search.
__global__ void dec_serial_kernel(DecParams p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    uint64_t pos = 0; // may exceed 32 bits: peek relative to the current word
    bool bad = false;
    for (uint64_t i = 0; i < p.n_out; ++i) {
        GlobalWords s2{p.in, p.in_bytes, pos >> 5};
        const uint64_t window = dec_peek64(s2, (uint32_t)(pos & 31u)) >> (64 - p.sentinel);
        const int k = dec_predecessor(p.keys, p.m, window);
        uint32_t len = 0;
        uint8_t sym = 0;
        if (k >= 0) {
            sym = p.ksyms[k];
            len = p.klens[k];
        } else {
            bad = true;
        }
        p.out[i] = sym;
        pos += len;
    }
    if (bad)
        atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
}

} // namespace

#endif // HFGPU_KERN_DECODE_HPP
