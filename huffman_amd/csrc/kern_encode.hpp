// Encoder kernels (src/huffman/encode.rs:57-136): pack, plan, scan, and the write() automaton.
// Part of the single translation unit huffman_amd/csrc/hfgpu.hip (device code
// for gfx950, wave64); included there, not compiled on its own.
#ifndef HFGPU_KERN_ENCODE_HPP
#define HFGPU_KERN_ENCODE_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "hfgpu.h"
#include "dev_common.hpp"

namespace {

// ------------------------------------------------------------------- encode
//
// Two pack kernels share the tile geometry, the image copy-out and the rule of
// word ownership below: enc_pack_kernel (described first; chained, planned and
// three-kernel mode, any model, ragged ends) and, further down,
// enc_pack_stream_kernel (planned mode, complete batches: one lookup per
// symbol) with enc_fix_tiles_kernel behind its speculative form.
//
// A tile is 16384 consecutive symbols: 512 threads x a run of 32 consecutive
// symbols (two 16-byte loads) each.  A workgroup handles a batch of four tiles.
//   1. the symbols stay in registers; a byte table of code lengths in LDS
//      gives every thread's bit total, and a workgroup scan turns the totals
//      into bit offsets inside the tile (and into the chunk lengths of the
//      gap array);
//   2. start bit of the batch in the stream: either read from a scanned array
//      (three-kernel mode) or obtained inside the same launch by a chained
//      scan with decoupled look-back over 8-byte {status,value} descriptors;
//   3. every thread looks its symbols up in an LDS table of {code, len} pairs,
//      merges G neighbouring
//      codes, streams them MSB-first through a 64-bit accumulator and ORs
//      finished 32-bit words into an LDS image of the tile's part of the
//      stream (the image is aligned to 16-byte units of the stream);
//   4. the image is copied out with coalesced, byte-swapped 16-byte stores.
// Stream words are owned by the tile that contains their first bit, so every
// output word is written exactly once with a plain store: a tile completes
// its last word by encoding up to 31 bits of the symbols that follow it and
// skips the leading word it shares with its predecessor.
// Codes longer than 32 bits do not fit a table entry: their entry carries a
// flag and the length only, and a thread that meets one re-encodes its run
// from the 64-bit table in global memory (rare, wave-divergent).

constexpr int ENC_THREADS = 512;
constexpr int ENC_WAVES = ENC_THREADS / 64;
constexpr int ENC_SPT = 32;
constexpr int ENC_TILE = ENC_THREADS * ENC_SPT; // 16384 symbols
constexpr int ENC_WINW = 5120;                  // words of stream image per pass (20 KiB)
constexpr int ENC_REPL = 1;                     // replicas of the {code, len} table in LDS
constexpr int ENC_BATCH = 4;                    // consecutive tiles per workgroup
constexpr int ENC_MIN_WAVES = 6;                // waves per SIMD the register budget allows
static_assert(ENC_TILE * ENC_BATCH == HF_PLAN_BATCH, "a planned batch is the encoder's batch");
constexpr uint32_t ENC_LONG_FLAG = 1u << 24;    // len word of a code above 32 bits: flag | len (four cannot wrap)

constexpr unsigned long long DESC_AGG = 1ull << 62;  // batch total published
constexpr unsigned long long DESC_INCL = 2ull << 62; // inclusive prefix published
constexpr unsigned long long DESC_VAL = (1ull << 62) - 1;

struct EncParams {
    const uint8_t *in;
    uint64_t n;
    const uint64_t *tab;   // 256 x (code << 6 | len), any length up to 57
    const uint2 *tab2;     // 256 x {code, len}; {0, ENC_LONG_FLAG | len} when len > 32
    const uint8_t *len8;   // 256 x len
    uint8_t *out;
    uint64_t out_cap;
    uint32_t lead_bits;
    uint32_t aligned16; // input base is 16-byte aligned
    uint32_t out_aligned16;
    uint32_t chunk_syms;
    uint16_t *chunk_bits;
    uint64_t *seg_off;
    uint64_t n_chunks;
    uint64_t *result;
    uint64_t *tile_off;       // three-kernel mode: exclusive bit offsets per tile
    // planned mode: start bit of every group of ENC_WAVES batches (scanned, [n_groups] = total),
    // start of every batch inside its group, and every batch's bit total
    const uint64_t *plan_off;
    const uint32_t *plan_local;
    const uint32_t *plan_tot;
    unsigned long long *desc; // chained mode: per-tile descriptors (zeroed)
    uint32_t *ticket;         // chained mode: tile counter (zeroed)
    uint32_t *abort_flag;     // chained mode: set when a wait expired
    uint64_t timeout_ticks;   // in 100 MHz wall-clock ticks
    uint32_t n_tiles;
    uint32_t first_batch; // batch index of this launch's first workgroup
    uint32_t wg_batches;   // enc_pack_stream_kernel: consecutive batches per workgroup
    uint32_t n_wg_batches; //   and batches of the launch
    // enc_pack_stream_kernel on a model with longer codes: tiles it could not
    // finish, {tile, start bit low, high, -} each, for enc_fix_tiles_kernel
    uint32_t *fail_list;
    uint32_t *fail_count;
    uint32_t debug;       // timing experiments only: bit mask of phases to skip
};

__device__ __forceinline__ uint32_t enc_sym(const uint32_t raw[ENC_SPT / 4], int k)
{
    return (raw[k >> 2] >> (8 * (k & 3))) & 0xffu;
}

// The thread's 32 symbols as 8 dwords.  FULL = all 32 exist and the input is
// 16-byte aligned (two 16-byte loads); otherwise byte loads, zeros past the end.
template <bool FULL>
__device__ __forceinline__ void enc_load_raw(const EncParams &p, uint64_t base, uint32_t vcnt,
                                             uint32_t raw[ENC_SPT / 4])
{
    if (FULL) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.in + base);
        const uint4 a = src[0], b = src[1];
        raw[0] = a.x, raw[1] = a.y, raw[2] = a.z, raw[3] = a.w;
        raw[4] = b.x, raw[5] = b.y, raw[6] = b.z, raw[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < ENC_SPT / 4; ++k)
            raw[k] = 0;
#pragma unroll
        for (int k = 0; k < ENC_SPT; ++k) // static register indices: no scratch
            if ((uint32_t)k < vcnt)
                raw[k >> 2] |= (uint32_t)p.in[base + k] << (8 * (k & 3));
    }
}

// Bit total of the thread's first vcnt symbols from the byte table of code
// lengths (one extract, one LDS byte read and a third of an add per symbol).
// Eight lookups are in flight at a time: the fences keep the scheduler from
// hoisting all 32 (and their registers) to the front.
template <bool FULL>
__device__ __forceinline__ uint32_t enc_sum_lens(const uint32_t raw[ENC_SPT / 4], uint32_t vcnt,
                                                 const uint8_t *s_len)
{
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < ENC_SPT / 8; ++q) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            e[j] = s_len[enc_sym(raw, 8 * q + j)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            bits += (FULL || (uint32_t)(8 * q + j) < vcnt) ? e[j] : 0u;
        __builtin_amdgcn_sched_barrier(0);
    }
    return bits;
}

// OR `len` bits of `code` into the stream image at image bit `bitpos`.
// LDS slot of image word w: one slot is left unused after every 32 words.
// Threads that produce the same number of words each (eight for 8-bit codes)
// would otherwise OR into the same few banks at the same time; irregular
// streams gain a little as well.
// (PAD = false: streams of short, irregular codes -- the kernels that use the
// carry sink -- keep the plain layout: the slot arithmetic costs them more
// than the conflicts it avoids.)
template <bool PAD>
__device__ __forceinline__ uint32_t enc_slot(uint32_t w) { return PAD ? w + (w >> 5) : w; }
constexpr int ENC_IMG_SLOTS = ENC_WINW + ENC_WINW / 32 + 4;
// The slot of image word w as a pointer, in two instructions: left to itself
// the compiler distributes the scaling over the sum and needs three.
template <bool PAD>
__device__ __forceinline__ uint32_t *enc_slot_ptr(uint32_t *stg, uint32_t w)
{
    if (!PAD)
        return stg + w;
    uint32_t byte_off;
    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(byte_off) : "v"(w), "v"(w >> 5));
    return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(stg) + byte_off);
}

template <bool PAD>
__device__ __forceinline__ void enc_or_bits(uint32_t *stg, uint32_t bitpos,
                                            uint64_t code, uint32_t len, uint32_t wlo,
                                            uint32_t whi, uint32_t clip)
{
    if (bitpos + len > clip) { // keep only the bits in front of image bit `clip`
        const uint32_t keep = clip > bitpos ? clip - bitpos : 0u;
        code >>= (len - keep);
        len = keep;
    }
    while (len > 0) {
        const uint32_t w = bitpos >> 5, o = bitpos & 31u;
        const uint32_t take = min(32u - o, len);
        const uint32_t part =
            (uint32_t)(code >> (len - take)) & (take == 32 ? ~0u : ((1u << take) - 1u));
        if (w >= wlo && w < whi)
            __hip_atomic_fetch_or(stg + enc_slot<PAD>(w - wlo), part << (32u - o - take), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        bitpos += take;
        len -= take;
    }
}

// Bit sink of one thread: codes are appended MSB-first to a 64-bit accumulator
// and every completed 32-bit word is ORed into the image.  Only image words
// [wlo, wlo + ENC_WINW) exist in a pass; others are dropped.
template <bool PAD>
struct EncSink {
    uint32_t *stg;
    uint32_t w;   // image word being filled, relative to wlo (wraps below the window)
    uint32_t nb;  // bits of it that are already spoken for
    uint64_t acc; // pending bits, right aligned; bits above nb are stale
    __device__ __forceinline__ EncSink(uint32_t *s, uint32_t bitpos, uint32_t wlo)
        : stg(s), w((bitpos >> 5) - wlo), nb(bitpos & 31u), acc(0)
    {
    }
    __device__ __forceinline__ void emit(uint32_t val)
    {
        if (w < (uint32_t)ENC_WINW)
            __hip_atomic_fetch_or(stg + enc_slot<PAD>(w), val, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        ++w;
    }
    __device__ __forceinline__ void append(uint32_t code, uint32_t len) // len <= 32
    {
        acc = (acc << len) | code;
        nb += len;
        if (nb >= 32) {
            nb -= 32;
            emit((uint32_t)(acc >> nb));
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nb > 0)
            emit((uint32_t)(acc << (32 - nb)));
    }
};

// The sink of the single-pass fast path: straight-line code in which only the
// LDS OR itself is predicated (lanes whose word is not complete yet sit the
// instruction out), and the image slot is computed from a word index.
template <bool PAD>
struct EncSinkFlat {
    uint32_t *stg;
    uint32_t w; // image word being filled
    uint32_t nb;
    uint64_t acc;
    bool alive; // once cleared, nothing is written any more
    __device__ __forceinline__ EncSinkFlat(uint32_t *s, uint32_t bitpos)
        : stg(s), w(bitpos >> 5), nb(bitpos & 31u), acc(0), alive(true)
    {
    }
    __device__ __forceinline__ void append(uint32_t code, uint32_t len) // len <= 32
    {
        acc = (acc << (len & 63u)) | code;
        nb += len;
        const bool full = nb >= 32 && alive;
        nb &= 31u;
        if (full)
            __hip_atomic_fetch_or(enc_slot_ptr<PAD>(stg, w), (uint32_t)(acc >> nb), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        w += full ? 1 : 0;
    }
    __device__ __forceinline__ void finish()
    {
        if (nb && alive)
            __hip_atomic_fetch_or(enc_slot_ptr<PAD>(stg, w), (uint32_t)(acc << (32 - nb)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// The same sink for merged groups of at most 31 bits.  The bit count of the
// word being filled lives in the top five bits of a register and the table
// supplies len | len << 27, so that one add advances it, wraps it and leaves
// "word complete" in the carry; the low bits only collect the lengths
// (at most 32 x 31, far from bit 27).
template <bool PAD>
struct EncSinkCarry {
    uint32_t *stg;
    uint32_t w;
    uint32_t nbl; // nb << 27 | junk
    uint64_t acc;
    static constexpr bool alive = true;
    __device__ __forceinline__ EncSinkCarry(uint32_t *s, uint32_t bitpos)
        : stg(s), w(bitpos >> 5), nbl(bitpos << 27), acc(0)
    {
    }
    __device__ __forceinline__ void append(uint32_t code, uint32_t y) // y = len | len << 27
    {
        acc = (acc << (y & 63u)) | code;
        uint32_t next;
        const bool full = __builtin_add_overflow(nbl, y, &next);
        nbl = next;
        if (full)
            __hip_atomic_fetch_or(enc_slot_ptr<PAD>(stg, w), (uint32_t)(acc >> (nbl >> 27)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        w += full ? 1 : 0;
    }
    __device__ __forceinline__ void finish()
    {
        const uint32_t nb = nbl >> 27;
        if (nb)
            __hip_atomic_fetch_or(enc_slot_ptr<PAD>(stg, w), (uint32_t)(acc << (32 - nb)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Generic form: walks the thread's run again through the 64-bit table with a
// rolled loop.  Used when a code is longer than a packed entry holds and when
// a tile needs more than one image pass (both rare).
template <bool PAD>
__device__ inline void enc_pack_thread_rolled(const EncParams &p, uint64_t sym0,
                                                    uint32_t vcnt, uint32_t *stg, uint32_t bitpos,
                                                    uint32_t nbits, uint32_t wlo)
{
    if (nbits == 0)
        return;
    if (((bitpos + nbits - 1) >> 5) < wlo || (bitpos >> 5) >= wlo + ENC_WINW)
        return;
    EncSink<PAD> sink(stg, bitpos, wlo);
    for (uint32_t k = 0; k < vcnt; ++k) {
        const uint64_t e = p.tab[p.in[sym0 + k]];
        const uint32_t len = (uint32_t)e & 63u;
        const uint64_t code = e >> 6;
        if (len > 32) {
            sink.append((uint32_t)(code >> 32), len - 32);
            sink.append((uint32_t)code, 32);
        } else {
            sink.append((uint32_t)code, len);
        }
    }
    sink.finish();
}

// Fast form for a tile that fits one image pass.  The {code, len} entries are
// looked up four at a time (64-bit LDS reads, no field extraction), the next
// four while the current four are merged, so only eight of them are live;
// G codes are merged per append.  SPEC = a merged group can exceed 32 bits
// (or the table holds codes longer than 32 bits, whose entries carry
// ENC_LONG_FLAG in the length).  The loop has no branches: a lane that meets
// such a group merely stops writing to the image -- what it wrote so far is
// correct -- and the function returns false for it.  The caller then repeats
// the lane's run with enc_pack_thread_rolled (ORing bits twice is harmless).
// Dword q of tile t's symbols, picked out of the batch's registers by masking
// (q is a compile-time constant at every use).  Written with AND/OR on
// purpose: a chain of selects on t is recognised as raw[t][q] and turns the
// whole array into a scratch-memory object.
__device__ __forceinline__ uint32_t enc_pick(const uint32_t (&raw)[ENC_BATCH][ENC_SPT / 4],
                                             uint32_t t, int q)
{
    uint32_t v = 0;
#pragma unroll
    for (int u = 0; u < ENC_BATCH; ++u)
        v |= raw[u][q] & (0u - (uint32_t)(t == (uint32_t)u));
    return v;
}

template <int G, bool SPEC, bool FULL, bool CARRY, bool PAD>
__device__ __forceinline__ bool enc_pack_thread(const uint32_t (&raw)[ENC_BATCH][ENC_SPT / 4],
                                                uint32_t t, uint32_t vcnt, const uint2 *s_tab2,
                                                uint32_t *stg, uint32_t bitpos)
{
    static_assert(!(CARRY && SPEC), "the carry sink needs groups of at most 31 bits");
    constexpr int LG = 4; // lookups per group = the four symbols of one input dword
    const uint2 *tab = s_tab2 + (threadIdx.x & (ENC_REPL - 1));
    std::conditional_t<CARRY, EncSinkCarry<PAD>, EncSinkFlat<PAD>> sink(stg, bitpos);
    uint2 cur[LG], nxt[LG];
    auto fetch = [&](uint2 e[LG], int q) {
        const uint32_t dw = enc_pick(raw, t, q); // the four symbols of this group
#pragma unroll
        for (int j = 0; j < LG; ++j) {
            const uint2 v = tab[((dw >> (8 * j)) & 0xffu) * ENC_REPL];
            e[j] = (FULL || (uint32_t)(LG * q + j) < vcnt) ? v : make_uint2(0, 0);
        }
    };
    fetch(cur, 0);
#pragma unroll
    for (int q = 0; q < ENC_SPT / LG; ++q) {
        if (q + 1 < ENC_SPT / LG)
            fetch(nxt, q + 1);
#pragma unroll
        for (int k = 0; k < LG; k += G) {
            uint32_t code = cur[k].x, len = cur[k].y;
#pragma unroll
            for (int j = 1; j < G; ++j) {
                code = (code << (cur[k + j].y & 31u)) | cur[k + j].x;
                len += cur[k + j].y;
            }
            if constexpr (SPEC)
                sink.alive = sink.alive && len <= 32u;
            sink.append(code, len);
        }
#pragma unroll
        for (int j = 0; j < LG; ++j)
            cur[j] = nxt[j];
        __builtin_amdgcn_sched_barrier(0);
    }
    sink.finish();
    return sink.alive;
}

// The first 32 stream bits that follow a tile, left aligned in *s_la (ORed in:
// the word starts as zero), and in *s_have how many of them exist (capped at
// 32): what the tile needs to complete its last stream word.  They do not
// depend on where the tile starts, so a wave that would otherwise idle
// computes them while wave 0 is looking back.  `la_sym` is the symbol at
// tile_end + lane, loaded together with the tile's own symbols so that no
// global-memory latency is added here.
__device__ __forceinline__ void enc_lookahead_bits(const EncParams &p, uint64_t tile_end,
                                                   uint32_t la_sym, const uint2 *s_tab2,
                                                   uint32_t *s_la, uint32_t *s_have)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t idx = tile_end + lane;
    const uint2 e2x = idx < p.n ? s_tab2[la_sym * ENC_REPL] : make_uint2(0, 0);
    uint64_t code = e2x.x;
    const uint32_t len = e2x.y & 63u;
    if (__builtin_expect((e2x.y & ENC_LONG_FLAG) != 0, 0))
        code = p.tab[la_sym] >> 6;
    auto contribute = [&](uint64_t c, uint32_t l, uint32_t off) { // off < 32, l > 0
        const int sh = 32 - (int)off - (int)l;
        const uint32_t part = sh >= 0 ? (uint32_t)(c << sh) : (uint32_t)(c >> -sh);
        __hip_atomic_fetch_or(s_la, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const uint32_t incl = wave_incl_scan(len);
    const uint32_t off = incl - len;
    if (len && off < 32u)
        contribute(code, len, off);
    uint32_t have = __shfl(incl, 63, 64);
    if (have < 32u && tile_end + 64 < p.n) {
        // only reachable when the data holds symbols whose code length is 0
        if (lane == 0) {
            for (uint64_t i = tile_end + 64; i < p.n && have < 32u; ++i) {
                const uint64_t e2 = p.tab[p.in[i]];
                const uint32_t l2 = (uint32_t)e2 & 63u;
                if (l2)
                    contribute(e2 >> 6, l2, have);
                have += l2;
            }
        }
        have = __shfl(have, 0, 64);
    }
    if (lane == 0)
        *s_have = min(have, 32u);
}

// CARRY: the length word becomes len | len << 27 (see EncSinkCarry).
template <bool CARRY>
__device__ __forceinline__ void enc_stage_table(const EncParams &p, uint2 *s_tab2, uint8_t *s_len)
{
    for (uint32_t i = threadIdx.x; i < 256 * ENC_REPL; i += ENC_THREADS) {
        uint2 e = p.tab2[i / ENC_REPL];
        if (CARRY)
            e.y |= e.y << 27;
        s_tab2[i] = e;
    }
    if (threadIdx.x < 64)
        reinterpret_cast<uint32_t *>(s_len)[threadIdx.x] =
            reinterpret_cast<const uint32_t *>(p.len8)[threadIdx.x];
}

// Three-kernel mode, kernel 1: bit total of every tile.
__global__ __launch_bounds__(ENC_THREADS) void enc_tile_bits_kernel(EncParams p)
{
    __shared__ __align__(16) uint8_t s_len[256];
    __shared__ uint32_t s_wave[ENC_WAVES];
    if (threadIdx.x < 64)
        reinterpret_cast<uint32_t *>(s_len)[threadIdx.x] =
            reinterpret_cast<const uint32_t *>(p.len8)[threadIdx.x];
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint64_t base = (uint64_t)tile * ENC_TILE;
    const uint64_t left = p.n - base;
    const uint32_t nsym = left < ENC_TILE ? (uint32_t)left : ENC_TILE;
    const uint32_t t0 = threadIdx.x * ENC_SPT;
    const uint32_t vcnt = nsym > t0 ? min(nsym - t0, (uint32_t)ENC_SPT) : 0u;
    uint32_t raw[ENC_SPT / 4];
    if (nsym == ENC_TILE && p.aligned16)
        enc_load_raw<true>(p, base + t0, vcnt, raw);
    else
        enc_load_raw<false>(p, base + t0, vcnt, raw);
    uint32_t sum = enc_sum_lens<false>(raw, vcnt, s_len);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63u) == 0)
        s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int i = 0; i < ENC_WAVES; ++i)
            total += s_wave[i];
        p.tile_off[tile] = total;
    }
}

// Planned mode, kernel 1: bit total of every batch from its histogram and the
// code lengths (one wave per batch, four bins per lane), the batch's start
// inside its group of ENC_WAVES batches (= one workgroup here) and the group's
// total, which kernel 2 scans: 8 times fewer entries for its single workgroup.
__global__ __launch_bounds__(ENC_THREADS) void enc_plan_bits_kernel(
    const uint32_t *__restrict__ batch_hist, const uint8_t *__restrict__ len8,
    uint64_t *group_off, uint32_t *local, uint32_t *tot, uint32_t n_batches, uint64_t *result,
    uint32_t *fail_count)
{
    __shared__ uint32_t s_tot[ENC_WAVES];
    if (blockIdx.x == 0 && threadIdx.x < HF_RES_WORDS)
        result[threadIdx.x] = 0; // first kernel of a planned encode: no separate memset
    if (blockIdx.x == 0 && threadIdx.x == 0 && fail_count)
        *fail_count = 0;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * ENC_WAVES + wv;
    uint32_t sum = 0; // at most 65536 symbols of at most 57 bits
    if (b < n_batches) {
        const uint4 h = reinterpret_cast<const uint4 *>(batch_hist + (uint64_t)b * 256)[lane];
        const uint32_t l4 = reinterpret_cast<const uint32_t *>(len8)[lane];
        sum = h.x * (l4 & 0xffu) + h.y * ((l4 >> 8) & 0xffu) + h.z * ((l4 >> 16) & 0xffu) +
              h.w * (l4 >> 24);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        sum += __shfl_xor(sum, d, 64);
    if (lane == 0)
        s_tot[wv] = sum;
    __syncthreads();
    if (threadIdx.x < ENC_WAVES) {
        uint32_t before = 0, all = 0;
        for (uint32_t i = 0; i < ENC_WAVES; ++i) {
            before += i < threadIdx.x ? s_tot[i] : 0u;
            all += s_tot[i];
        }
        const uint32_t mine = blockIdx.x * ENC_WAVES + threadIdx.x;
        if (mine < n_batches) {
            local[mine] = before;
            tot[mine] = s_tot[threadIdx.x];
        }
        if (threadIdx.x == 0)
            group_off[blockIdx.x] = all;
    }
}

// Three-kernel and planned mode, kernel 2: in-place exclusive scan of the
// tile (batch) totals by one workgroup of 1024 lanes, 16 entries per lane and
// round: wave scans by shuffles, one LDS hand-over between the 16 waves.  The
// grand total lands in tile_off[n] and result[0].
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void enc_scan_tiles_kernel(uint64_t *tile_off,
                                                                      uint32_t n,
                                                                      uint64_t *result)
{
    __shared__ uint64_t s_wave[SCAN_THREADS / 64];
    constexpr uint32_t PER = 16;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint64_t carry = 0; // total of all earlier rounds (every lane keeps its own copy)
    for (uint32_t base = 0; base < n; base += SCAN_THREADS * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) // unconditional loads: all 16 in flight together
            v[k] = tile_off[min(i0 + k, n - 1)];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            v[k] = (i0 + k < n) ? v[k] : 0;
            sum += v[k];
        }
        uint64_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d)
                incl += t;
        }
        if (lane == 63)
            s_wave[wv] = incl;
        __syncthreads();
        uint64_t before = 0, round_total = 0;
#pragma unroll
        for (uint32_t i = 0; i < SCAN_THREADS / 64; ++i) {
            const uint64_t x = s_wave[i];
            before += i < wv ? x : 0;
            round_total += x;
        }
        uint64_t run = carry + before + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (i0 + k < n)
                tile_off[i0 + k] = run;
            run += v[k];
        }
        carry += round_total;
        __syncthreads(); // s_wave is rewritten in the next round
    }
    if (threadIdx.x == 0) {
        tile_off[n] = carry;
        result[HF_RES_TOTAL_BITS] = carry;
    }
}

// Decoupled look-back, run by wave 0: exclusive bit prefix of batch `tile`.
// Descriptors are single 8-byte words written and read with agent-scope
// atomics (write-through / L1-bypassing), so the value is its own flag and no
// other data is handed between workgroups.  Tiles are numbered by a ticket
// taken at workgroup start, so every predecessor of a waiting tile belongs to
// a workgroup that is already running; waits are bounded by a wall clock.
__device__ __forceinline__ bool enc_lookback(const EncParams &p, uint32_t tile, uint64_t &excl,
                                             uint64_t *dbg)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t sum = 0;
    int64_t j = (int64_t)tile - 1;
    const uint64_t t_start = wall_clock64();
    uint32_t polls = 0, steps = 0;
    while (true) {
        const int64_t idx = j - (int64_t)lane;
        unsigned long long d;
        uint32_t spins = 0;
        ++steps;
        // One read at a time: a descriptor read crosses the fabric (about
        // 1.4 us under load) and the descriptors of all running workgroups
        // share a few lines, so keeping several reads in flight, or reading a
        // wider window, slows every workgroup's reads down (both measured).
        while (true) {
            ++polls;
            d = idx >= 0 ? __hip_atomic_load(p.desc + idx, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)
                         : DESC_INCL; // virtual tile -1: inclusive prefix 0
            // ready once an inclusive prefix is in sight and every descriptor
            // between it and this batch carries its total
            const unsigned long long valid = __ballot((d >> 62) != 0);
            const unsigned long long incl = __ballot((d >> 62) == 2);
            const unsigned long long below = incl ? ((incl & (0 - incl)) - 1) : ~0ull;
            if ((valid & below) == below)
                break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                const bool expired = wall_clock64() - t_start > p.timeout_ticks;
                const bool aborted = __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (expired || aborted) {
                    if (lane == 0) {
                        __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS],
                                 (unsigned long long)HF_FLAG_TIMEOUT);
                    }
                    excl = 0;
                    return false;
                }
            }
        }
        const unsigned long long incl_mask = __ballot((d >> 62) == 2);
        const uint32_t first = incl_mask ? (uint32_t)__ffsll((long long)incl_mask) - 1u : 64u;
        uint64_t v = lane <= first ? (d & DESC_VAL) : 0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1)
            v += __shfl_xor(v, s, 64);
        sum += v;
        if (incl_mask)
            break;
        j -= 64;
    }
    if (dbg && lane == 0)
        *dbg = polls | ((uint64_t)steps << 32);
    excl = sum;
    return true;
}

// Second half of an image pass: complete the tile's last word from the
// symbols that follow it (last pass only) and copy the pass's words out with
// coalesced byte-swapped 16-byte stores, clearing the image on the way.
template <bool PAD>
__device__ __forceinline__ void enc_finish_pass(const EncParams &p, uint32_t *stg,
                                                uint32_t la_bits, uint32_t la_have,
                                                uint64_t E, uint64_t gbase,
                                                uint64_t g0, uint64_t g1, uint32_t endbit,
                                                uint32_t need, uint32_t nwords, uint32_t wlo,
                                                bool reuse)
{
    const uint32_t whi = wlo + ENC_WINW;
    const bool last_pass = nwords - 1 < whi;
    // the word that holds the tile's last bit is completed by the first `need`
    // bits of what follows the tile
    if (threadIdx.x == 0 && last_pass && need)
        __hip_atomic_fetch_or(stg + enc_slot<PAD>((endbit >> 5) - wlo), la_bits >> (32u - need),
                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    // the stream ends inside this tile's last word when fewer than `need`
    // bits follow the tile: only its leading bytes exist then
    const uint64_t limit =
        (need && last_pass && la_have < need) ? (E + la_have + 7) >> 3 : ~0ull;
    const uint64_t ga = max(g0, gbase + wlo);
    const uint64_t gb = min(g1, gbase + whi);
    // 16-byte units [ua, ub) of the stream intersect [ga, gb); the sweep starts
    // at the first unit of the image so that the tile's leading bits, which sit
    // in a word owned by the previous tile, are cleared as well
    const uint64_t ua = (gbase + wlo) >> 2, ub = (gb + 3) >> 2;
    for (uint64_t u = ua + threadIdx.x; u < ub; u += ENC_THREADS) {
        const uint32_t j = (uint32_t)((u << 2) - gbase) - wlo; // image word of the unit
        // the thread that copies a unit out also clears it: the image is all
        // zero again when the pass ends, without a separate sweep
        // (the unit's four words share a 32-word row, so their slots are adjacent)
        uint32_t *src = stg + enc_slot<PAD>(j);
        uint4 v;
        if (PAD) {
            v = make_uint4(src[0], src[1], src[2], src[3]);
            src[0] = 0, src[1] = 0, src[2] = 0, src[3] = 0;
        } else { // j is a multiple of four and the image is 16-byte aligned
            uint4 *src4 = reinterpret_cast<uint4 *>(__builtin_assume_aligned(src, 16));
            v = *src4;
            *src4 = make_uint4(0, 0, 0, 0);
        }
        const uint32_t vals[4] = {bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)};
        const uint64_t b = u << 4;
        if ((u << 2) >= ga && (u << 2) + 4 <= gb && b + 16 <= limit && b + 16 <= p.out_cap &&
            p.out_aligned16) {
            *reinterpret_cast<uint4 *>(p.out + b) = make_uint4(vals[0], vals[1], vals[2], vals[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t g = (u << 2) + q;
                if (g < ga || g >= gb)
                    continue;
                const uint64_t bb = g << 2;
                if (bb + 4 <= limit && bb + 4 <= p.out_cap) {
                    *reinterpret_cast<uint32_t *>(p.out + bb) = vals[q];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (bb + i < limit) {
                            if (bb + i < p.out_cap)
                                p.out[bb + i] = (uint8_t)(vals[q] >> (8 * i));
                            else
                                atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS],
                                         (unsigned long long)HF_FLAG_CAPACITY);
                        }
                    }
                }
            }
        }
    }
    if (reuse)
        __syncthreads(); // the next pass ORs into units that were just read and cleared
}

// A workgroup encodes a batch of ENC_BATCH consecutive tiles in two stages.
// Stage A, all tiles of the batch at once: load the symbols (8 x 16 bytes in
// flight per thread), sum the code lengths, scan them into per-thread bit
// offsets (kept in LDS) and, in chained mode, publish the batch's bit total
// straight away so that later batches never wait for this one's packing.
// Stage B, tile by tile: pack, complete the last word, copy out.
// The batch index is a ticket taken at workgroup start (one device-scope
// atomic per 64 Ki symbols keeps that counter far below its saturation rate);
// it also orders the look-back: every earlier batch belongs to a workgroup
// that is already running.
//
// FULL = every tile of every batch of this launch has all its symbols and the
// input is 16-byte aligned; the host launches the (at most one) ragged batch
// at the end of the input, and unaligned inputs, with FULL = false so that the
// hot instantiation carries no tail handling.  first_batch offsets the batch
// index of such a follow-up launch.
//
// Built with -DHF_ENC_TRACE (HF_ENC_TRACE=1 python huffman_amd/build.py) and run
// with the context option debug = 16, threads 0 and 448 of every workgroup
// stamp the wall clock at the phase boundaries below into the unused tail of
// the output buffer (40 x 8 bytes per batch); tools/trace_enc.py prints the
// resulting timeline (the stream itself must end before that tail, or it is
// overwritten: a diagnostic build, not for production use).
#ifdef HF_ENC_TRACE
#define ENC_TRACE(i)                                                                          \
    do {                                                                                      \
        if (trp)                                                                              \
            trp[(i)] = wall_clock64();                                                        \
    } while (0)
#define ENC_TRACE_TILE(t, i) ENC_TRACE(7 + 3 * (t) + (i))
constexpr uint64_t ENC_TRACE_BYTES = 320; // per batch
#else
#define ENC_TRACE(i)                                                                          \
    do {                                                                                      \
    } while (0)
#define ENC_TRACE_TILE(t, i) ENC_TRACE(0)
#endif
template <int G, bool SPEC, bool FULL, bool CARRY>
__global__ __launch_bounds__(ENC_THREADS, ENC_MIN_WAVES) void enc_pack_kernel(EncParams p)
{
#ifdef HF_ENC_TRACE
    const uint64_t trace_t0 = wall_clock64();
    uint64_t *trp = nullptr;
#endif
    constexpr bool PAD = !CARRY; // image layout, see enc_slot
    const bool CHAINED = p.desc != nullptr; // else: tile offsets come from the scan kernel
    __shared__ __align__(16) uint32_t stg2[2][ENC_IMG_SLOTS];
    __shared__ uint2 s_tab[256 * ENC_REPL];
    __shared__ __align__(16) uint8_t s_len[256];
    __shared__ uint32_t s_toff[ENC_BATCH][ENC_THREADS + 1];
    __shared__ uint32_t s_wave[ENC_BATCH][ENC_WAVES];
    __shared__ uint64_t s_start;
    __shared__ uint32_t s_ctl[2];
    __shared__ uint32_t s_la[ENC_BATCH], s_have[ENC_BATCH]; // look-ahead bits per tile
    __shared__ uint32_t s_batch;

    // the ticket first: its round trip overlaps with the staging of the tables
    if (CHAINED) {
        if (threadIdx.x == 0)
            s_batch = p.first_batch + ((p.debug & 4u) ? blockIdx.x : atomicAdd(p.ticket, 1u));
    }
    enc_stage_table<CARRY>(p, s_tab, s_len);
    if (threadIdx.x < ENC_BATCH)
        s_la[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < sizeof(stg2) / 16; i += ENC_THREADS)
        reinterpret_cast<uint4 *>(&stg2[0][0])[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t batch = CHAINED ? s_batch : p.first_batch + blockIdx.x;
#ifdef HF_ENC_TRACE
    if ((p.debug & 16u) && (threadIdx.x == 0 || threadIdx.x == 448)) {
        const uint64_t tail = ((uint64_t)p.n_tiles / ENC_BATCH + 2) * ENC_TRACE_BYTES;
        trp = reinterpret_cast<uint64_t *>(p.out + ((p.out_cap - tail) & ~15ull)) +
              (uint64_t)batch * 40 + (threadIdx.x ? 20 : 0);
        trp[0] = trace_t0;
    }
#endif
    ENC_TRACE(1);
    const uint32_t tile0 = batch * ENC_BATCH;
    if (tile0 >= p.n_tiles)
        return;
    const uint32_t t0 = threadIdx.x * ENC_SPT;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;

    // ---- stage A
    uint32_t raw[ENC_BATCH][ENC_SPT / 4];
    uint32_t vcnt[ENC_BATCH];
    uint32_t la_sym = 0; // wave t + 1: the symbol `lane` places behind tile t
    uint64_t la_end = 0; //             and where that tile ends
#pragma unroll
    for (int t = 0; t < ENC_BATCH; ++t) {
        const uint64_t base = (uint64_t)(tile0 + t) * ENC_TILE;
        const uint64_t left = base < p.n ? p.n - base : 0;
        const uint32_t nsym = left < ENC_TILE ? (uint32_t)left : ENC_TILE;
        if (wv == (uint32_t)t + 1) {
            la_end = base + nsym;
            if (la_end + lane < p.n)
                la_sym = p.in[la_end + lane];
        }
        if (FULL)
            __builtin_assume(nsym == ENC_TILE);
        vcnt[t] = FULL ? (uint32_t)ENC_SPT : (nsym > t0 ? min(nsym - t0, (uint32_t)ENC_SPT) : 0u);
        enc_load_raw<FULL>(p, base + t0, vcnt[t], raw[t]);
    }
    __builtin_amdgcn_sched_barrier(0); // all global loads of the batch are issued first
    uint32_t nbits[ENC_BATCH], incl[ENC_BATCH];
#pragma unroll
    for (int t = 0; t < ENC_BATCH; ++t) {
        nbits[t] = enc_sum_lens<FULL>(raw[t], vcnt[t], s_len);
        incl[t] = wave_incl_scan(nbits[t]);
        if (lane == 63)
            s_wave[t][wv] = incl[t];
    }
    ENC_TRACE(2);
    __syncthreads();
    ENC_TRACE(3);
    uint32_t batch_bits = 0;
#pragma unroll
    for (int t = 0; t < ENC_BATCH; ++t) {
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (uint32_t i = 0; i < ENC_WAVES; ++i) {
            const uint32_t x = s_wave[t][i];
            woff += i < wv ? x : 0u;
            tot += x;
        }
        s_toff[t][threadIdx.x] = woff + incl[t] - nbits[t];
        if (threadIdx.x == ENC_THREADS - 1)
            s_toff[t][ENC_THREADS] = tot;
        batch_bits += tot;
    }

    ENC_TRACE(4);
    // waves 1.. complete the tiles' last words in advance, one tile each
    if (wv >= 1 && wv <= (uint32_t)ENC_BATCH && tile0 + wv - 1 < p.n_tiles)
        enc_lookahead_bits(p, la_end, la_sym, s_tab, &s_la[wv - 1], &s_have[wv - 1]);
    // start bit of the batch in the stream (lead bits included)
    if (CHAINED) {
        if (threadIdx.x < 64) {
            uint64_t excl = 0;
            bool ok = true;
            if (batch == 0) {
                if (threadIdx.x == 0)
                    __hip_atomic_store(p.desc, DESC_INCL | (unsigned long long)batch_bits,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (threadIdx.x == 0)
                    __hip_atomic_store(p.desc + batch, DESC_AGG | (unsigned long long)batch_bits,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(p.debug & 8u))
                    {
#ifdef HF_ENC_TRACE
                    ok = enc_lookback(p, batch, excl, trp ? trp + 19 : nullptr);
#else
                    ok = enc_lookback(p, batch, excl, nullptr);
#endif
                }
                if (ok && threadIdx.x == 0)
                    __hip_atomic_store(p.desc + batch,
                                       DESC_INCL | (unsigned long long)(excl + batch_bits),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x == 0) {
                s_start = excl + p.lead_bits;
                s_ctl[0] = ok ? 1u : 0u;
            }
        }
    } else {
        if (threadIdx.x == 0) {
            if (p.plan_off) {
                // the batch's place was worked out from its histogram: if the
                // totals disagree, the histograms belong to other data
                const uint64_t at = p.plan_off[batch / ENC_WAVES] + p.plan_local[batch];
                const bool ok = p.plan_tot[batch] == batch_bits;
                if (!ok)
                    atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS],
                             (unsigned long long)HF_FLAG_PLAN);
                s_start = at + p.lead_bits;
                s_ctl[0] = ok ? 1u : 0u;
            } else {
                s_start = p.tile_off[tile0] + p.lead_bits;
                s_ctl[0] = 1u;
            }
        }
    }
    ENC_TRACE(5);
    __syncthreads();
    ENC_TRACE(6);
    if (!s_ctl[0])
        return; // a wait expired, or the plan is not this input's: the host reports it
    uint64_t S = s_start;
    if (p.debug & 1u)
        return;

    // ---- stage B: a rolled loop (one copy of the code) that always works on
    // register set 0; the sets are rotated down after every tile
#pragma unroll 1
    for (uint32_t t = 0; t < ENC_BATCH; ++t) {
        const uint32_t tile = tile0 + t;
        if (tile >= p.n_tiles)
            break;
        // the table is read again here on purpose: without this fence the
        // compiler keeps all of stage A's lookups alive (in scratch) instead
        asm volatile("" ::: "memory");
        const uint32_t vc = vcnt[0];
        // consecutive tiles alternate between two images: a wave that has
        // copied its share of tile t out packs tile t + 1 straight away, and
        // the barrier after that packing also tells that the image of tile t
        // is clear again for tile t + 2
        uint32_t *stg = stg2[t & 1u];
        const uint64_t base = (uint64_t)tile * ENC_TILE;
        const uint64_t left = p.n - base;
        const uint64_t tile_end = base + (left < ENC_TILE ? left : (uint64_t)ENC_TILE);
        const uint32_t tile_bits = s_toff[t][ENC_THREADS];
        const uint32_t toff = s_toff[t][threadIdx.x];
        const uint32_t my_bits = s_toff[t][threadIdx.x + 1] - toff;
        const uint64_t E = S + tile_bits;

        // gap array: bit length of every chunk, start bit of every segment
        if (p.chunk_bits) {
            const uint32_t cpt = p.chunk_syms / ENC_SPT; // threads per chunk
            const uint32_t cit = ENC_THREADS / cpt;      // chunks per tile
            if (threadIdx.x < cit) {
                const uint64_t c = (uint64_t)tile * cit + threadIdx.x;
                if (c < p.n_chunks) {
                    const uint32_t b0 = s_toff[t][threadIdx.x * cpt];
                    const uint32_t b1 = s_toff[t][(threadIdx.x + 1) * cpt];
                    p.chunk_bits[c] = (uint16_t)(b1 - b0);
                    if ((c % HF_SEG_CHUNKS) == 0)
                        p.seg_off[c / HF_SEG_CHUNKS] = S + b0;
                }
            }
        }
        if (tile_end == p.n && threadIdx.x == 0)
            p.result[HF_RES_TOTAL_BITS] = E - p.lead_bits;

        // stream words [g0, g1) belong to this tile; the image starts at the
        // 16-byte unit of the stream that holds bit S
        const uint64_t gbase = (S >> 7) << 2;
        const uint64_t g0 = tile == 0 ? 0 : (S + 31) >> 5;
        const uint64_t g1 = (E + 31) >> 5;
        if (g1 > g0) {
            const uint32_t s_lo = (uint32_t)(S - (gbase << 5)); // 0..127
            const uint32_t endbit = s_lo + tile_bits;
            const uint32_t need = (32u - (endbit & 31u)) & 31u;
            const uint32_t nwords = (uint32_t)(g1 - gbase);
            if (nwords <= ENC_WINW) {
                // common case, straight-line: the whole tile fits one image pass
                if (my_bits && !(p.debug & 2u)) {
                    const bool done =
                        enc_pack_thread<G, SPEC, FULL, CARRY, PAD>(raw, 0, vc, s_tab, stg, s_lo + toff);
                    if (SPEC && __builtin_expect(!done, 0)) // rare by the host's choice of G
                        enc_pack_thread_rolled<PAD>(p, base + t0, vc, stg, s_lo + toff, my_bits, 0);
                }
                ENC_TRACE_TILE(t, 0); // packed
                enc_finish_pass<PAD>(p, stg, s_la[t], s_have[t], E, gbase, g0, g1, endbit, need, nwords,
                                0, false);
            } else {
                for (uint32_t wlo = 0; wlo < nwords; wlo += ENC_WINW) {
                    enc_pack_thread_rolled<PAD>(p, base + t0, vc, stg, s_lo + toff, my_bits, wlo);
                    enc_finish_pass<PAD>(p, stg, s_la[t], s_have[t], E, gbase, g0, g1, endbit, need,
                                    nwords, wlo, true);
                }
            }
        }
        ENC_TRACE_TILE(t, 2); // copied out
        S = E;
#pragma unroll
        for (int u = 0; u + 1 < ENC_BATCH; ++u) {
            vcnt[u] = vcnt[u + 1];
#pragma unroll
            for (int q = 0; q < ENC_SPT / 4; ++q)
                raw[u][q] = raw[u + 1][q];
        }
    }
}

// Planned encode of complete batches with a model whose merged groups of G
// codes stay below 32 bits (the carry sink's condition): every symbol is
// looked up ONCE and the input is requested one tile ahead all the time.
// enc_pack_kernel reads the length table for the scan and the {code, len}
// table again for the packing, because a chained batch must publish its bit
// total before it packs anything, and it requests a whole batch of symbols in
// one burst and then computes for several times as long.  A planned batch
// knows its start bit beforehand, so a workgroup takes its tiles (those of
// p.wg_batches consecutive batches: one or two, see the launch) one after the
// other:
//   1. look the tile's symbols up, merge G codes at a time and keep the merged
//      groups in registers; the length words also add up to the thread's bit
//      total;
//   2. wave scan, one LDS hand-over between the waves (one barrier): the
//      thread's bit offset inside the tile and the tile's total; the gap array
//      comes from the neighbours' offsets by shuffles;
//   3. the groups go from the registers through the carry sink into the image;
//   4. last word, copy-out (enc_finish_pass, one more barrier).
// The next tile's symbols are requested before step 2's barrier, so about a
// tile per workgroup is always on its way.  The bits that complete a tile's
// last word are worked out by one wave per tile (wave t mod 8) from 64 symbols
// that it requested a tile earlier.  Every batch's bit total is compared with
// the planned one: batch histograms that belong to other data set
// HF_FLAG_PLAN (the stream is then void; every store is still checked against
// the buffer's capacity).
// PACKED: a group is one register, code | len << 27 (merged codes of at most
// 22 bits): added to the sink's counter as it is -- the code bits of a tile's
// groups stay below bit 27 in sum -- at the price of one shift per group to
// get the length alone; otherwise two registers, code and len | len << 27.
// The carry sink (EncSinkCarry) for the plain image layout with a pointer to
// the word being filled instead of its index: no address arithmetic at the OR.
struct EncSinkPtr {
    uint32_t *wp;
    uint32_t nbl; // nb << 27 | junk
    uint64_t acc;
    __device__ __forceinline__ EncSinkPtr(uint32_t *stg, uint32_t bitpos)
        : wp(stg + (bitpos >> 5)), nbl(bitpos << 27), acc(0)
    {
    }
    __device__ __forceinline__ void step(uint32_t y)
    {
        uint32_t next;
        const bool full = __builtin_add_overflow(nbl, y, &next);
        nbl = next;
        if (full)
            __hip_atomic_fetch_or(wp, (uint32_t)(acc >> (nbl >> 27)), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        wp += full ? 1 : 0;
    }
    __device__ __forceinline__ void append(uint32_t code, uint32_t y) // y = len | len << 27
    {
        acc = (acc << (y & 63u)) | code;
        step(y);
    }
    // x = code | len << 27: the code bits only add to the junk below bit 27
    __device__ __forceinline__ void append_packed(uint32_t x)
    {
        acc = (acc << (x >> 27)) | (x & 0x07ffffffu);
        step(x);
    }
    __device__ __forceinline__ void finish()
    {
        const uint32_t nb = nbl >> 27;
        if (nb)
            __hip_atomic_fetch_or(wp, (uint32_t)(acc << (32 - nb)), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// enc_finish_pass for a tile that fits the (plain) image in one pass: the
// units that lie wholly inside the tile's words, the stream and the buffer --
// all but the first and the last, as a rule -- are told apart by two 32-bit
// comparisons with bounds worked out once per tile.
__device__ __forceinline__ void enc_finish_stream(const EncParams &p, uint32_t *stg, uint32_t la_bits,
                                                  uint32_t la_have, uint64_t E, uint64_t gbase,
                                                  uint64_t g0, uint64_t g1, uint32_t endbit,
                                                  uint32_t need)
{
    if (threadIdx.x == 0 && need)
        __hip_atomic_fetch_or(stg + (endbit >> 5), la_bits >> (32u - need), __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    // the stream ends inside this tile's last word when fewer than `need`
    // bits follow the tile: only its leading bytes exist then
    const uint64_t limit = (need && la_have < need) ? (E + la_have + 7) >> 3 : ~0ull;
    const uint64_t ga = max(g0, gbase);
    const uint64_t ua = gbase >> 2, ub = (g1 + 3) >> 2;
    const uint32_t nunits = (uint32_t)(ub - ua);
    const uint32_t jlo = (uint32_t)((ga - gbase + 3) >> 2);
    uint32_t jhi = (uint32_t)((g1 - gbase) >> 2);
    const uint64_t whole = min(limit, p.out_cap) >> 4; // 16-byte units that exist in full
    if (whole < ub)
        jhi = whole > ua ? min(jhi, (uint32_t)(whole - ua)) : 0u;
    if (!p.out_aligned16)
        jhi = 0;
    uint4 *dst = reinterpret_cast<uint4 *>(p.out) + ua;
    uint4 *img = reinterpret_cast<uint4 *>(__builtin_assume_aligned(stg, 16));
    for (uint32_t j = threadIdx.x; j < nunits; j += ENC_THREADS) {
        // the thread that copies a unit out also clears it
        const uint4 v = img[j];
        img[j] = make_uint4(0, 0, 0, 0);
        const uint32_t vals[4] = {bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)};
        if (j >= jlo && j < jhi) {
            dst[j] = make_uint4(vals[0], vals[1], vals[2], vals[3]);
        } else {
            const uint64_t u = ua + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t g = (u << 2) + q;
                if (g < ga || g >= g1)
                    continue;
                const uint64_t bb = g << 2;
                if (bb + 4 <= limit && bb + 4 <= p.out_cap) {
                    *reinterpret_cast<uint32_t *>(p.out + bb) = vals[q];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (bb + i < limit) {
                            if (bb + i < p.out_cap)
                                p.out[bb + i] = (uint8_t)(vals[q] >> (8 * i));
                            else
                                atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS],
                                         (unsigned long long)HF_FLAG_CAPACITY);
                        }
                    }
                }
            }
        }
    }
}

// Codes of at most ENC_STREAM_MAXLEN bits: a tile's part of the stream always
// fits the image, so this kernel has no second image pass.
constexpr uint32_t ENC_STREAM_MAXLEN = 11;
constexpr uint32_t ENC_STREAM_WINW = ENC_TILE * ENC_STREAM_MAXLEN / 32 + 8;
// One image is enough: the barrier of the next tile's scan lies between a
// tile's copy-out and the next packing.  25 KB of LDS and 64 registers: four
// workgroups per CU.
constexpr int ENC_STREAM_WAVES = 8;  // waves per SIMD
// (Two codes per group in two registers each -- the speculative form for models
// that merge two codes at a time -- need 74 registers: three workgroups per CU.)
template <int G, bool PACKED>
constexpr int enc_stream_waves = (G == 2 && !PACKED) ? ENC_MIN_WAVES : ENC_STREAM_WAVES;
template <int G, bool PACKED, bool SPEC>
__global__ __launch_bounds__(ENC_THREADS, (enc_stream_waves<G, PACKED>)) void enc_pack_stream_kernel(EncParams p)
{
    static_assert(!(SPEC && PACKED), "groups that may overflow keep their length word");
    constexpr int LG = 4;            // lookups per group = the four symbols of one input dword
    constexpr int NG = ENC_SPT / G;  // merged groups per thread and tile
    __shared__ __align__(16) uint32_t stg[ENC_STREAM_WINW + 4];
    __shared__ uint2 s_tab[256 * ENC_REPL];
    __shared__ __align__(16) uint8_t s_len[256];
    __shared__ uint32_t s_wave[2][ENC_WAVES];
    __shared__ uint64_t s_start;
    __shared__ uint32_t s_la[2], s_have[2];

    const uint32_t batch_lo = p.first_batch + blockIdx.x * p.wg_batches;
    const uint32_t batch_hi = min(batch_lo + p.wg_batches, p.first_batch + p.n_wg_batches);
    if (batch_lo >= batch_hi)
        return;
    const uint32_t tile_lo = batch_lo * ENC_BATCH, tile_hi = batch_hi * ENC_BATCH;
    const uint32_t lane = threadIdx.x & 63u;
    // (the wave's number in a scalar register: what depends on it branches, not masks)
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // A tile's symbols: uniform base + the thread's 32-bit offset.  Nothing
    // else is loaded from global memory between a request and its use one tile
    // later (loads complete in order: a younger load that is waited for would
    // wait for the symbols as well).
    uint4 ra, rb;
    auto request = [&](uint32_t tile) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.in + (uint64_t)tile * ENC_TILE);
        ra = src[threadIdx.x * 2];
        rb = src[threadIdx.x * 2 + 1];
    };
    // the 64 symbols behind a tile, for the wave that completes its last word
    // (clamped: past the end of the input they are not used)
    auto request_la = [&](uint32_t tile) -> uint32_t {
        const uint64_t at = (uint64_t)(tile + 1) * ENC_TILE + lane;
        return p.in[at < p.n ? at : p.n - 1];
    };
    request(tile_lo);
    uint32_t la_sym = 0, planned = 0;
    if (wv == (tile_lo & (ENC_WAVES - 1)))
        la_sym = request_la(tile_lo);
    enc_stage_table<true>(p, s_tab, s_len);
    if (threadIdx.x == 0)
        s_start = p.plan_off[batch_lo / ENC_WAVES] + p.plan_local[batch_lo] + p.lead_bits;
    for (uint32_t i = threadIdx.x; i < sizeof(stg) / 16; i += ENC_THREADS)
        reinterpret_cast<uint4 *>(&stg[0])[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // what is the same in every lane is kept in scalar registers: the vector
    // registers are needed for the merged groups
    uint64_t S;
    {
        const uint64_t s = s_start;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)s);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(s >> 32));
        S = (uint64_t)hi << 32 | lo;
    }
    uint32_t batch_bits = 0;

#pragma unroll 1
    for (uint32_t tile = tile_lo; tile < tile_hi; ++tile) {
        // ---- 1. one lookup per symbol; merged groups stay in registers
        uint32_t code[NG], y[PACKED ? 1 : NG], ysum = 0;
        uint32_t worst = 0; // SPEC: the largest group length, shifted up by five bits
        {
            const uint32_t raw[ENC_SPT / 4] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            uint2 cur[LG], nxt[LG];
            auto fetch = [&](uint2 e[LG], int q) {
#pragma unroll
                for (int j = 0; j < LG; ++j)
                    e[j] = s_tab[((raw[q] >> (8 * j)) & 0xffu) * ENC_REPL];
            };
            fetch(cur, 0);
#pragma unroll
            for (int q = 0; q < ENC_SPT / LG; ++q) {
                if (q + 1 < ENC_SPT / LG)
                    fetch(nxt, q + 1);
#pragma unroll
                for (int k = 0; k < LG; k += G) {
                    uint32_t c = cur[k].x, l = cur[k].y; // l = len | len << 27
#pragma unroll
                    for (int j = 1; j < G; ++j) {
                        c = (c << (cur[k + j].y & 31u)) | cur[k + j].x;
                        l += cur[k + j].y;
                    }
                    ysum += l;
                    if (SPEC) // lengths and long-code flags sit below bit 27
                        worst = max(worst, l << 5);
                    if (PACKED)
                        c |= l & 0xf8000000u;
                    // (pinned here: left alone, the compiler moves the merging down to
                    // the packing and keeps all 32 table entries alive until then)
                    if (PACKED) {
                        asm volatile("" : "+v"(c));
                    } else {
                        asm volatile("" : "+v"(c), "+v"(l));
                        y[(q * LG + k) / G] = l;
                    }
                    code[(q * LG + k) / G] = c;
                }
#pragma unroll
                for (int j = 0; j < LG; ++j)
                    cur[j] = nxt[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the low bits of the length words collect plain lengths (at most 32 x 57;
        // the flags of codes above 32 bits add up from bit 24 on)
        const uint32_t nbits = ysum & (SPEC ? 0x00ffffffu : 0x07ffffffu);
        // SPEC: a group of more than 31 bits, or one with a code above 32 bits,
        // cannot go through the carry sink: the lane writes nothing and the tile
        // is put on the list for enc_fix_tiles_kernel
        const bool bad = SPEC && worst > (31u << 5);
        // ---- 2. bit offsets
        const uint32_t incl = wave_incl_scan(nbits);
        const bool wave_bad = SPEC && __ballot(bad) != 0;
        if (lane == 63) // (a wave's total stays below 2^17)
            s_wave[tile & 1u][wv] = incl | (wave_bad ? 0x80000000u : 0u);
        // one wave completes the tile's last word in advance
        if (wv == (tile & (ENC_WAVES - 1))) {
            if (lane == 0)
                s_la[tile & 1u] = 0;
            enc_lookahead_bits(p, (uint64_t)(tile + 1) * ENC_TILE, la_sym, s_tab, &s_la[tile & 1u],
                               &s_have[tile & 1u]);
        }
        const uint32_t planned_now = planned;
        if (tile + 1 < tile_hi) { // the next tile's symbols travel while this one is packed
            request(tile + 1);
            if (wv == ((tile + 1) & (ENC_WAVES - 1)))
                la_sym = request_la(tile + 1);
            // (by one thread: a value that is the same in all lanes would be moved
            // to a scalar register at once, and that waits for the symbols)
            if (threadIdx.x == 0 && ((tile + 1) & (ENC_BATCH - 1)) == ENC_BATCH - 1)
                planned = p.plan_tot[(tile + 1) / ENC_BATCH];
        }
        __syncthreads();
        // the eight wave totals, scanned across lanes 0..7 of every wave
        static_assert(ENC_WAVES == 8, "the scan below covers eight lanes");
        uint32_t wsum = s_wave[tile & 1u][lane & (ENC_WAVES - 1)];
        const bool any_bad = SPEC && __ballot((wsum >> 31) != 0) != 0;
        if (SPEC)
            wsum &= 0x7fffffffu;
        wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x111, 0xf, 0xf, false); // row_shr:1
        wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x112, 0xf, 0xf, false); // row_shr:2
        wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x114, 0xf, 0xf, false); // row_shr:4
        const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)wsum, ENC_WAVES - 1);
        const uint32_t woff = wv ? (uint32_t)__builtin_amdgcn_readlane((int)wsum, wv - 1) : 0u;
        const uint32_t toff = woff + incl - nbits;
        batch_bits += tile_bits;
        const uint64_t E = S + tile_bits;

        // gap array: bit length of every chunk, start bit of every segment
        if (p.chunk_bits) {
            // threads per chunk: 2, 4 or 8 (as a shift: a division by a run-time
            // value is worked out before the loop and its result kept, in scratch
            // memory in the end, whose reload waits for the symbols)
            const uint32_t cshift = (uint32_t)__builtin_ctz(p.chunk_syms / ENC_SPT);
            const uint32_t cpt = 1u << cshift;
            const uint32_t wave_end = woff + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t nxt_off = __shfl_down(toff, cpt, 64);
            const uint32_t b1 = lane + cpt < 64u ? nxt_off : wave_end;
            if ((threadIdx.x & (cpt - 1)) == 0) {
                uint32_t tid = threadIdx.x; // (opaque: worked out here, per tile, not kept)
                asm volatile("" : "+v"(tid));
                const uint64_t c = ((uint64_t)tile << 9 >> cshift) + (tid >> cshift);
                static_assert(ENC_THREADS == 512, "chunks per tile = 512 >> cshift");
                if (c < p.n_chunks) {
                    p.chunk_bits[c] = (uint16_t)(b1 - toff);
                    if ((c % HF_SEG_CHUNKS) == 0)
                        p.seg_off[c / HF_SEG_CHUNKS] = S + toff;
                }
            }
        }
        if ((tile & (ENC_BATCH - 1)) == ENC_BATCH - 1) {
            // the batches were placed by their histograms: if a total disagrees,
            // the histograms belong to other data
            if (threadIdx.x == 0 && planned_now != batch_bits)
                atomicOr((unsigned long long *)&p.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_PLAN);
            batch_bits = 0;
        }
        if (threadIdx.x == 0 && (uint64_t)(tile + 1) * ENC_TILE == p.n)
            p.result[HF_RES_TOTAL_BITS] = E - p.lead_bits;

        // ---- 3. and 4.  stream words [g0, g1) belong to this tile; the image
        // starts at the 16-byte unit of the stream that holds bit S
        const uint64_t gbase = (S >> 7) << 2;
        const uint64_t g0 = tile == 0 ? 0 : (S + 31) >> 5;
        const uint64_t g1 = (E + 31) >> 5;
        // SPEC: more bits than the image holds; nothing of the tile is written here
        const bool overflow = SPEC && g1 - gbase > ENC_STREAM_WINW;
        if (SPEC && (any_bad || overflow) && threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(p.fail_count, 1u);
            p.fail_list[4 * slot + 0] = tile;
            p.fail_list[4 * slot + 1] = (uint32_t)S;
            p.fail_list[4 * slot + 2] = (uint32_t)(S >> 32);
        }
        if (g1 > g0 && !overflow) {
            const uint32_t s_lo = (uint32_t)(S - (gbase << 5)); // 0..127
            const uint32_t endbit = s_lo + tile_bits;
            const uint32_t need = (32u - (endbit & 31u)) & 31u;
            if (nbits && !bad) {
                EncSinkPtr sink(stg, s_lo + toff);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (PACKED)
                        sink.append_packed(code[g]);
                    else
                        sink.append(code[g], y[g]);
                }
                sink.finish();
            }
            // (g1 - gbase words, at most ENC_STREAM_WINW)
            const uint32_t la_bits = (uint32_t)__builtin_amdgcn_readfirstlane(s_la[tile & 1u]);
            const uint32_t la_have = (uint32_t)__builtin_amdgcn_readfirstlane(s_have[tile & 1u]);
            enc_finish_stream(p, stg, la_bits, la_have, E, gbase, g0, g1, endbit, need);
        }
        S = E;
    }
}


// Tiles that enc_pack_stream_kernel<.., SPEC> left out (a group of codes above
// 31 bits, a code above 32 bits, or more bits than its image holds: rare by the
// choice of that kernel for a model) are encoded again here by the general
// path: lengths and codes from the 64-bit table in global memory, as many image
// passes as the tile needs.  Every word the tile owns is written in full, so
// what the first kernel wrote of it does not matter.  Workgroups beyond the list's
// length leave at once; the list is empty for most inputs.
constexpr uint32_t ENC_FIX_GRID = 1024;
__global__ __launch_bounds__(ENC_THREADS) void enc_fix_tiles_kernel(EncParams p)
{
    constexpr bool PAD = false;
    __shared__ __align__(16) uint32_t stg[ENC_IMG_SLOTS];
    __shared__ uint2 s_tab[256 * ENC_REPL];
    __shared__ __align__(16) uint8_t s_len[256];
    __shared__ uint32_t s_wave[ENC_WAVES];
    __shared__ uint64_t s_tab64[256]; // code << 6 | len, any length
    __shared__ uint32_t s_la, s_have;
    const uint32_t count = *p.fail_count;
    if (blockIdx.x >= count)
        return;
    enc_stage_table<false>(p, s_tab, s_len);
    if (threadIdx.x < 256)
        s_tab64[threadIdx.x] = p.tab[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < sizeof(stg) / 16; i += ENC_THREADS)
        reinterpret_cast<uint4 *>(&stg[0])[i] = make_uint4(0, 0, 0, 0);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t tile = p.fail_list[4 * i];
        const uint64_t S = (uint64_t)p.fail_list[4 * i + 2] << 32 | p.fail_list[4 * i + 1];
        const uint64_t base = (uint64_t)tile * ENC_TILE + threadIdx.x * ENC_SPT;
        const uint64_t tile_end = (uint64_t)(tile + 1) * ENC_TILE;
        if (threadIdx.x == 0)
            s_la = 0;
        __syncthreads(); // tables, image and s_la are ready; the last round is over
        uint32_t raw[ENC_SPT / 4];
        enc_load_raw<true>(p, base, ENC_SPT, raw);
        uint32_t nbits = 0;
#pragma unroll
        for (int k = 0; k < ENC_SPT; ++k)
            nbits += s_len[enc_sym(raw, k)];
        const uint32_t incl = wave_incl_scan(nbits);
        if (lane == 63)
            s_wave[wv] = incl;
        if (wv == 0) {
            const uint64_t at = tile_end + lane;
            enc_lookahead_bits(p, tile_end, p.in[at < p.n ? at : p.n - 1], s_tab, &s_la, &s_have);
        }
        __syncthreads();
        uint32_t woff = 0, tile_bits = 0;
#pragma unroll
        for (uint32_t w = 0; w < ENC_WAVES; ++w) {
            const uint32_t x = s_wave[w];
            woff += w < wv ? x : 0u;
            tile_bits += x;
        }
        const uint32_t toff = woff + incl - nbits;
        const uint64_t E = S + tile_bits;
        const uint64_t gbase = (S >> 7) << 2;
        const uint64_t g0 = tile == 0 ? 0 : (S + 31) >> 5;
        const uint64_t g1 = (E + 31) >> 5;
        if (g1 > g0) {
            const uint32_t s_lo = (uint32_t)(S - (gbase << 5));
            const uint32_t endbit = s_lo + tile_bits;
            const uint32_t need = (32u - (endbit & 31u)) & 31u;
            const uint32_t nwords = (uint32_t)(g1 - gbase);
            const uint32_t bitpos = s_lo + toff;
            for (uint32_t wlo = 0; wlo < nwords; wlo += ENC_WINW) {
                // the lane's bits that fall into image words [wlo, wlo + ENC_WINW)
                if (nbits && ((bitpos + nbits - 1) >> 5) >= wlo && (bitpos >> 5) < wlo + ENC_WINW) {
                    EncSink<PAD> sink(stg, bitpos, wlo);
#pragma unroll
                    for (int k = 0; k < ENC_SPT; ++k) {
                        const uint64_t e = s_tab64[enc_sym(raw, k)];
                        const uint32_t len = (uint32_t)e & 63u;
                        const uint64_t code = e >> 6;
                        if (len > 32) {
                            sink.append((uint32_t)(code >> 32), len - 32);
                            sink.append((uint32_t)code, 32);
                        } else {
                            sink.append((uint32_t)code, len);
                        }
                    }
                    sink.finish();
                }
                enc_finish_pass<PAD>(p, stg, s_la, s_have, E, gbase, g0, g1, endbit, need, nwords, wlo,
                                     true);
            }
        }
    }
}


// A model in which every one of the 256 symbols has an 8-bit code (what the
// Huffman construction gives data that does not compress: encrypted or already
// compressed bytes) turns the encoder into a byte map: stream byte i is the
// code of symbol i, every chunk has 8 x chunk_syms bits, nothing needs a scan.
// Runs at the rate of a copy.  Eight 16-byte loads per lane are in flight (the
// register that was just mapped is refilled at once, as in the counting pass;
// non-temporal loads make this kernel 3 % faster and the decoder that reads
// the stream next 6 % slower, so the loads are plain).
constexpr int ENCB_THREADS = 512;
constexpr int ENCB_VECS = 8;

__device__ __forceinline__ uint32_t encb_map4(const uint8_t *s_code, uint32_t v)
{
    const uint32_t a = s_code[v & 0xffu], b = s_code[(v >> 8) & 0xffu];
    const uint32_t c = s_code[(v >> 16) & 0xffu], d = s_code[v >> 24];
    return a | (b << 8) | (c << 16) | (d << 24);
}

// STREAM: loads and stores both non-temporal, for inputs far larger than the caches
// (a plain copy kernel gains 10 % that way, this kernel 3 %: 1 GiB 0.424 -> 0.411 ms;
// either alone is slower than plain accesses).
typedef uint32_t encb_u32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ uint4 encb_load(const uint4 *p)
{
    if (STREAM) {
        const encb_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const encb_u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <bool STREAM>
__device__ __forceinline__ void encb_store(uint4 *p, uint4 y)
{
    if (STREAM) {
        encb_u32x4 w;
        w.x = y.x, w.y = y.y, w.z = y.z, w.w = y.w;
        __builtin_nontemporal_store(w, reinterpret_cast<encb_u32x4 *>(p));
    } else {
        *p = y;
    }
}
template <bool STREAM>
__global__ __launch_bounds__(ENCB_THREADS) void enc_bytes_kernel(
    const uint8_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ tab,
    uint8_t *__restrict__ out, uint64_t out_cap, uint32_t chunk_syms,
    uint16_t *__restrict__ chunk_bits, uint64_t *__restrict__ seg_off, uint64_t *__restrict__ result)
{
    __shared__ __align__(16) uint8_t s_code[256];
    if (threadIdx.x < 256)
        s_code[threadIdx.x] = (uint8_t)(tab[threadIdx.x] >> 6); // entries are code << 6 | len
    if (out_cap < n) { // (uniform) nothing is written into a buffer that cannot hold the stream
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr((unsigned long long *)&result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_CAPACITY);
        return;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0)
        result[HF_RES_TOTAL_BITS] = 8ull * n;
    // gap array: chunk c has 8 x (its symbols) bits, segment s starts at bit 8 x 256 x chunk_syms x s
    if (chunk_bits) {
        const uint64_t n_chunks = (n + chunk_syms - 1) / chunk_syms;
        for (uint64_t c = (uint64_t)blockIdx.x * ENCB_THREADS + threadIdx.x; c < n_chunks;
             c += (uint64_t)gridDim.x * ENCB_THREADS) {
            const uint64_t left = n - c * chunk_syms;
            chunk_bits[c] = (uint16_t)(8u * (uint32_t)(left < chunk_syms ? left : chunk_syms));
            if ((c % HF_SEG_CHUNKS) == 0)
                seg_off[c / HF_SEG_CHUNKS] = 8ull * c * chunk_syms;
        }
    }
    const bool vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    constexpr uint64_t SPAN = (uint64_t)ENCB_VECS * ENCB_THREADS; // 16-byte units per span
    const uint64_t n_spans = vec ? (n >> 4) / SPAN : 0;
    const uint4 *vin = reinterpret_cast<const uint4 *>(in);
    uint4 *vout = reinterpret_cast<uint4 *>(out);
    uint64_t sp = blockIdx.x;
    if (sp < n_spans) {
        uint4 x[ENCB_VECS];
        const uint4 *src = vin + sp * SPAN + threadIdx.x;
#pragma unroll
        for (int k = 0; k < ENCB_VECS; ++k) {
            x[k] = encb_load<STREAM>(src + k * ENCB_THREADS);
            __builtin_amdgcn_sched_barrier(0); // in the order in which the loop refills them
        }
#pragma nounroll
        for (; sp < n_spans; sp += gridDim.x) {
            // after the last span every refill reads that span's first unit again (unused)
            const bool more = sp + gridDim.x < n_spans;
            const uint32_t step = more ? ENCB_THREADS : 0;
            src = vin + (more ? (sp + gridDim.x) * SPAN + threadIdx.x : sp * SPAN);
            uint4 *dst = vout + sp * SPAN + threadIdx.x;
#pragma unroll
            for (int k = 0; k < ENCB_VECS; ++k) {
                uint4 y;
                y.x = encb_map4(s_code, x[k].x), y.y = encb_map4(s_code, x[k].y);
                y.z = encb_map4(s_code, x[k].z), y.w = encb_map4(s_code, x[k].w);
                encb_store<STREAM>(dst + k * ENCB_THREADS, y);
                __builtin_amdgcn_sched_barrier(0); // the refill goes into the registers just mapped
                x[k] = encb_load<STREAM>(src + k * step);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // what the spans leave over (everything, for buffers that are not 16-byte aligned)
    for (uint64_t i = n_spans * SPAN * 16 + (uint64_t)blockIdx.x * ENCB_THREADS + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * ENCB_THREADS)
        out[i] = s_code[in[i]];
}


// Exact emission schedule of Encoder::write (src/huffman/encode.rs:91-116)
// for one call: one lane replays put (:57-66) and cleanup (:67-83) over the
// symbols so that the call's return value and the counters match the
// reference for small writes.  state = {buffer, remaining_bits, writeout};
// out = {return value, error (1 = code length above 64)}.
__global__ void enc_automaton_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                     const uint64_t *__restrict__ codes,
                                     const uint64_t *__restrict__ lens, uint64_t *state,
                                     uint64_t *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    uint64_t buffer = state[0], remaining = state[1], total = state[2];
    uint64_t emitted = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t code = codes[in[i]], len = lens[in[i]];
        if (len > 64) {
            out[1] = 1;
            break;
        }
        while (len > remaining) { // put
            emitted += 1;
            buffer <<= 8;
            remaining += 8;
        }
        remaining -= len;
        buffer += code << (remaining & 63);
        if (buffer & 0x0000000000FF0000ull) { // cleanup
            emitted += 5;
            buffer <<= 40;
            remaining += 40;
        }
    }
    const uint64_t tz = buffer ? (uint64_t)__builtin_ctzll(buffer) : 64;
    state[0] = buffer;
    state[1] = remaining;
    state[2] = total + emitted;
    out[0] = emitted + 8 - tz / 8;
}

} // namespace

#endif // HFGPU_KERN_ENCODE_HPP
