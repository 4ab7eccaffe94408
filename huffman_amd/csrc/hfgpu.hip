// hfgpu: HIP kernels (gfx950 / CDNA4, wave64) and the C ABI around them.
//
// Everything here is HBM-bound byte/bit work: no matrix cores.  The design
// rules are coalesced 16-byte global accesses, conflict-free LDS layouts, one
// pass over the input where the data flow allows it, and enough workgroups in
// flight (>> 256 CUs) to cover HBM latency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "hfgpu.h"
#include "hfgpu_internal.h"

namespace {

// ----------------------------------------------------------------- helpers

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d)
            v += t;
    }
    return v;
}

// Exclusive scan over a 256-thread workgroup.  s_wave needs 4 words and must
// not be reused by the caller before the next __syncthreads().
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 63)
        s_wave[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t x = s_wave[i];
        woff += i < wv ? x : 0u;
        tot += x;
    }
    total = tot;
    return woff + incl - v;
}

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// ---------------------------------------------------------------- histogram
//
// One LDS histogram per workgroup laid out [bin][32 columns]: a lane only ever
// touches column (lane & 31), i.e. its own LDS bank, so a wave's 64 atomic adds
// are bank-conflict free whatever the data looks like (lanes l and l+32 share
// a column but are served in different halves of the instruction).  Skewed
// inputs, where one bin takes most of the increments, therefore run at the
// same rate as uniform ones.  Columns are summed and merged into the global
// bins with one 64-bit atomic per bin and workgroup.

constexpr int HIST_THREADS = 512;
constexpr int HIST_COLS = 32;
constexpr int HIST_UNROLL = 4; // 16-byte vectors in flight per lane

__device__ __forceinline__ void hist_add4(uint32_t *col, uint32_t v)
{
    __hip_atomic_fetch_add(col + ((v & 0xffu) << 5), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(col + (((v >> 8) & 0xffu) << 5), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(col + (((v >> 16) & 0xffu) << 5), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(col + ((v >> 24) << 5), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const uint8_t *__restrict__ in,
                                                            uint64_t n,
                                                            unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t bins[256 * HIST_COLS];
    for (uint32_t i = threadIdx.x; i < 256 * HIST_COLS; i += HIST_THREADS)
        bins[i] = 0;
    __syncthreads();
    uint32_t *col = bins + (threadIdx.x & (HIST_COLS - 1));

    // bytes before the first 16-byte boundary and after the last whole vector
    const uint64_t mis = (16 - ((uintptr_t)in & 15)) & 15;
    const uint64_t head = mis < n ? mis : n;
    const uint64_t nvec = (n - head) >> 4;
    const uint64_t tail = head + (nvec << 4);
    if (blockIdx.x == 0) {
        for (uint64_t i = threadIdx.x; i < head; i += HIST_THREADS)
            __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        for (uint64_t i = tail + threadIdx.x; i < n; i += HIST_THREADS)
            __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    const uint4 *vin = reinterpret_cast<const uint4 *>(in + head);
    const uint64_t stride = (uint64_t)gridDim.x * HIST_THREADS;
    uint64_t v = (uint64_t)blockIdx.x * HIST_THREADS + threadIdx.x;
    // software pipeline: the next HIST_UNROLL vectors are requested before the
    // current ones are counted, so loads stay in flight during the LDS phase
    uint4 x[HIST_UNROLL], y[HIST_UNROLL];
    bool have = v + (HIST_UNROLL - 1) * stride < nvec;
    if (have) {
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; ++u)
            x[u] = vin[v + u * stride];
    }
    while (have) {
        const uint64_t vn = v + HIST_UNROLL * stride;
        const bool more = vn + (HIST_UNROLL - 1) * stride < nvec;
        if (more) {
#pragma unroll
            for (int u = 0; u < HIST_UNROLL; ++u)
                y[u] = vin[vn + u * stride];
        }
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; ++u) {
            hist_add4(col, x[u].x);
            hist_add4(col, x[u].y);
            hist_add4(col, x[u].z);
            hist_add4(col, x[u].w);
        }
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; ++u)
            x[u] = y[u];
        v = vn;
        have = more;
    }
    for (; v < nvec; v += stride) {
        const uint4 x = vin[v];
        hist_add4(col, x.x);
        hist_add4(col, x.y);
        hist_add4(col, x.z);
        hist_add4(col, x.w);
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < HIST_COLS; ++i) // rotate so the 32 lanes of a group hit 32 banks
            s += bins[threadIdx.x * HIST_COLS + ((i + threadIdx.x) & (HIST_COLS - 1))];
        if (s)
            atomicAdd(&hist[threadIdx.x], (unsigned long long)s);
    }
}

// The same histogram, and in addition one histogram per batch of
// HF_PLAN_BATCH consecutive symbols (uint32_t[256] each).  Whatever codebook
// the global histogram leads to, a batch's histogram times the code lengths is
// that batch's bit total, so the encoder can place every batch without
// reading the input a second time and without any exchange between its
// workgroups.  A workgroup counts a batch (512 lanes x 8 x 16 bytes) while
// the loads of its next batch are in flight, then sums and clears the 32
// columns of every bin.
constexpr uint32_t PLAN_BATCH = HF_PLAN_BATCH;
constexpr int PLAN_VECS = PLAN_BATCH / 16 / HIST_THREADS; // 16-byte vectors per lane and batch

template <bool ALIGNED>
__device__ __forceinline__ void hist_batch_load(const uint8_t *__restrict__ in, uint64_t n,
                                                uint64_t b, uint4 (&x)[PLAN_VECS])
{
    const uint64_t base = b * PLAN_BATCH;
    if (ALIGNED && base + PLAN_BATCH <= n) {
        const uint4 *vin = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < PLAN_VECS; ++k)
            x[k] = vin[threadIdx.x + k * HIST_THREADS];
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(HIST_THREADS) void hist_batches_kernel(
    const uint8_t *__restrict__ in, uint64_t n, unsigned long long *__restrict__ hist,
    uint32_t *__restrict__ batch_hist, uint64_t n_batches)
{
    __shared__ __align__(16) uint32_t bins[256 * HIST_COLS];
    for (uint32_t i = threadIdx.x; i < 256 * HIST_COLS; i += HIST_THREADS)
        bins[i] = 0;
    __syncthreads();
    uint32_t *col = bins + (threadIdx.x & (HIST_COLS - 1));
    // column sums: lanes 2b and 2b + 1 take 16 columns of bin b each, as four
    // 16-byte pieces in an order that keeps 16 neighbouring lanes on 16
    // different bank groups
    const uint32_t bin = threadIdx.x >> 1, half = threadIdx.x & 1u;
    uint4 *mine = reinterpret_cast<uint4 *>(bins + bin * HIST_COLS + half * 16);
    unsigned long long running = 0;

    uint4 x[PLAN_VECS], y[PLAN_VECS];
    uint64_t b = blockIdx.x;
    if (b < n_batches)
        hist_batch_load<ALIGNED>(in, n, b, x);
    for (; b < n_batches; b += gridDim.x) {
        const uint64_t bn = b + gridDim.x;
        if (bn < n_batches)
            hist_batch_load<ALIGNED>(in, n, bn, y);
        const uint64_t base = b * PLAN_BATCH;
        if (ALIGNED && base + PLAN_BATCH <= n) {
#pragma unroll
            for (int k = 0; k < PLAN_VECS; ++k) {
                hist_add4(col, x[k].x);
                hist_add4(col, x[k].y);
                hist_add4(col, x[k].z);
                hist_add4(col, x[k].w);
            }
        } else { // ragged last batch, or an input that is not 16-byte aligned
            const uint64_t end = base + PLAN_BATCH < n ? base + PLAN_BATCH : n;
            for (uint64_t i = base + threadIdx.x; i < end; i += HIST_THREADS)
                __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t piece = (k + (threadIdx.x >> 3)) & 3u;
            const uint4 v = mine[piece];
            mine[piece] = make_uint4(0, 0, 0, 0);
            s += v.x + v.y + v.z + v.w;
        }
        s += __shfl_xor(s, 1, 64);
        if (half == 0) {
            batch_hist[b * 256 + bin] = s;
            running += s;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PLAN_VECS; ++k)
            x[k] = y[k];
    }
    if (half == 0 && running)
        atomicAdd(&hist[bin], running);
}

// ------------------------------------------------------------------- encode
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
//   3. every thread looks its symbols up in an LDS table of {code, len} pairs
//      (four replicas, a lane uses replica lane % 4), merges G neighbouring
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
__device__ __forceinline__ uint32_t enc_slot(uint32_t w) { return w + (w >> 5); }
constexpr int ENC_IMG_SLOTS = ENC_WINW + ENC_WINW / 32 + 4;
// The slot of image word w as a pointer, in two instructions: left to itself
// the compiler distributes the scaling over the sum and needs three.
__device__ __forceinline__ uint32_t *enc_slot_ptr(uint32_t *stg, uint32_t w)
{
    uint32_t byte_off;
    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(byte_off) : "v"(w), "v"(w >> 5));
    return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(stg) + byte_off);
}

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
            __hip_atomic_fetch_or(stg + enc_slot(w - wlo), part << (32u - o - take), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        bitpos += take;
        len -= take;
    }
}

// Bit sink of one thread: codes are appended MSB-first to a 64-bit accumulator
// and every completed 32-bit word is ORed into the image.  Only image words
// [wlo, wlo + ENC_WINW) exist in a pass; others are dropped.
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
            __hip_atomic_fetch_or(stg + enc_slot(w), val, __ATOMIC_RELAXED,
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
            __hip_atomic_fetch_or(enc_slot_ptr(stg, w), (uint32_t)(acc >> nb), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        w += full ? 1 : 0;
    }
    __device__ __forceinline__ void finish()
    {
        if (nb && alive)
            __hip_atomic_fetch_or(enc_slot_ptr(stg, w), (uint32_t)(acc << (32 - nb)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// The same sink for merged groups of at most 31 bits.  The bit count of the
// word being filled lives in the top five bits of a register and the table
// supplies len | len << 27, so that one add advances it, wraps it and leaves
// "word complete" in the carry; the low bits only collect the lengths
// (at most 32 x 31, far from bit 27).
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
            __hip_atomic_fetch_or(enc_slot_ptr(stg, w), (uint32_t)(acc >> (nbl >> 27)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        w += full ? 1 : 0;
    }
    __device__ __forceinline__ void finish()
    {
        const uint32_t nb = nbl >> 27;
        if (nb)
            __hip_atomic_fetch_or(enc_slot_ptr(stg, w), (uint32_t)(acc << (32 - nb)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Generic form: walks the thread's run again through the 64-bit table with a
// rolled loop.  Used when a code is longer than a packed entry holds and when
// a tile needs more than one image pass (both rare).
__device__ inline void enc_pack_thread_rolled(const EncParams &p, uint64_t sym0,
                                                    uint32_t vcnt, uint32_t *stg, uint32_t bitpos,
                                                    uint32_t nbits, uint32_t wlo)
{
    if (nbits == 0)
        return;
    if (((bitpos + nbits - 1) >> 5) < wlo || (bitpos >> 5) >= wlo + ENC_WINW)
        return;
    EncSink sink(stg, bitpos, wlo);
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

template <int G, bool SPEC, bool FULL, bool CARRY>
__device__ __forceinline__ bool enc_pack_thread(const uint32_t (&raw)[ENC_BATCH][ENC_SPT / 4],
                                                uint32_t t, uint32_t vcnt, const uint2 *s_tab2,
                                                uint32_t *stg, uint32_t bitpos)
{
    static_assert(!(CARRY && SPEC), "the carry sink needs groups of at most 31 bits");
    constexpr int LG = 4; // lookups per group = the four symbols of one input dword
    const uint2 *tab = s_tab2 + (threadIdx.x & (ENC_REPL - 1));
    std::conditional_t<CARRY, EncSinkCarry, EncSinkFlat> sink(stg, bitpos);
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
    uint64_t *group_off, uint32_t *local, uint32_t *tot, uint32_t n_batches, uint64_t *result)
{
    __shared__ uint32_t s_tot[ENC_WAVES];
    if (blockIdx.x == 0 && threadIdx.x < HF_RES_WORDS)
        result[threadIdx.x] = 0; // first kernel of a planned encode: no separate memset
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
        __hip_atomic_fetch_or(stg + enc_slot((endbit >> 5) - wlo), la_bits >> (32u - need),
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
        uint32_t *src = stg + enc_slot(j);
        const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
        src[0] = 0, src[1] = 0, src[2] = 0, src[3] = 0;
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
                        enc_pack_thread<G, SPEC, FULL, CARRY>(raw, 0, vc, s_tab, stg, s_lo + toff);
                    if (SPEC && __builtin_expect(!done, 0)) // rare by the host's choice of G
                        enc_pack_thread_rolled(p, base + t0, vc, stg, s_lo + toff, my_bits, 0);
                }
                ENC_TRACE_TILE(t, 0); // packed
                enc_finish_pass(p, stg, s_la[t], s_have[t], E, gbase, g0, g1, endbit, need, nwords,
                                0, false);
            } else {
                for (uint32_t wlo = 0; wlo < nwords; wlo += ENC_WINW) {
                    enc_pack_thread_rolled(p, base + t0, vc, stg, s_lo + toff, my_bits, wlo);
                    enc_finish_pass(p, stg, s_la[t], s_have[t], E, gbase, g0, g1, endbit, need,
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
    uint32_t istg_words;
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
    while ((e & 0x8000u) && e < HF_LUT_INVALID) {
        const uint32_t idx = (hi << used) >> (32 - HF_LUT_W2);
        e = s_lut[(1u << p.w1) + ((e & 0x7fffu) << HF_LUT_W2) + idx];
        used += HF_LUT_W2;
    }
    if (e < HF_LUT_INVALID) {
        const uint32_t len = e >> 8;
        br.buf <<= len;
        br.bits -= (int32_t)len;
        return e & 0xffu;
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
    if (ESC && __builtin_expect(e & 0x8000u, 0))
        return dec_escape(br, e, s_lut, p, bad);
    const uint32_t len = e >> 8;
    br.buf <<= len;
    br.bits -= (int32_t)len;
    return e & 0xffu;
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
// predecessor search.  Returns a plain entry (len << 8 | sym); len 0 and
// `bad` set when the window lies below the smallest key.
__device__ inline uint32_t dec_escape_window(uint64_t win, uint32_t e, const uint16_t *s_lut,
                                             const DecParams &p, bool &bad)
{
    const uint32_t hi = (uint32_t)(win >> 32);
    uint32_t used = p.w1;
    while ((e & 0x8000u) && e < HF_LUT_INVALID) {
        const uint32_t idx = (hi << used) >> (32 - HF_LUT_W2);
        e = s_lut[(1u << p.w1) + ((e & 0x7fffu) << HF_LUT_W2) + idx];
        used += HF_LUT_W2;
    }
    if (e < HF_LUT_INVALID)
        return e;
    if (e == HF_LUT_SEARCH) {
        const int k = dec_predecessor(p.keys, p.m, win >> (64 - p.sentinel));
        if (k >= 0)
            return ((uint32_t)p.klens[k] << 8) | p.ksyms[k];
    }
    bad = true;
    return 0;
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
template <int CS, int K, bool ESC, bool SWZ, bool SPEC>
__global__ __launch_bounds__(DEC_THREADS) void dec_kernel_reg(DecParams p)
{
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr uint32_t OROW = CS / 4 + 1; // +1: lanes start on different banks
    using Img = typename std::conditional<SWZ, SwzWords, LdsWords>::type;
    const uint32_t lut_bytes = (p.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes); // input, then output
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
    const uint4 lut0 = threadIdx.x < lut_bytes / 16 ? lut16[threadIdx.x] : make_uint4(0, 0, 0, 0);
    // (straight-line on purpose: a load behind a branch is waited for at once)
    const uint64_t last_seg = (p.n_chunks - 1) / HF_SEG_CHUNKS;
    const uint64_t so0 = p.seg_off[seg], so1 = p.seg_off[min(seg + 1, last_seg)];
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t seg_start = so0 - p.bit_base;
    const uint64_t ubase = seg_start >> 7;
    const uint32_t s_lo = (uint32_t)(seg_start & 127u);
    constexpr uint32_t IMG_CAP = HF_SEG_CHUNKS * CS / 4 + 16; // words
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
        if (threadIdx.x < lut_bytes / 16)
            dst[threadIdx.x] = lut0;
        for (uint32_t i = threadIdx.x + DEC_THREADS; i < lut_bytes / 16; i += DEC_THREADS)
            dst[i] = lut16[i];
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
    } else if (fast) {
        // No reader state besides the bit position: every K symbols the lane
        // rebuilds a 64-bit window from the three image words at its position;
        // no refill branches.  Either K * sentinel <= 64, so the window cannot
        // run dry, or (SPEC, prefix-free models only) the host chose K so that
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
            uint32_t used = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = g + k;
                if (j < CS) {
                    uint32_t e = s_lut[(uint32_t)(win >> 32) >> shift];
                    if (ESC && __builtin_expect((e & 0x8000u) != 0, 0))
                        e = dec_escape_window(win, e, s_lut, p, bad);
                    const uint32_t len = e >> 8;
                    win <<= len;
                    used += len;
                    // byte j % 4 of the output word takes the entry's low byte
                    constexpr uint32_t keep = 0x03020100u;
                    const uint32_t sel = (keep & ~(0xffu << (8 * (j & 3)))) | (0x04u << (8 * (j & 3)));
                    outw[j >> 2] = __builtin_amdgcn_perm(e, outw[j >> 2], sel);
                }
            }
            pos += used;
            if (SPEC)
                dirty = dirty || used > 64u;
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
    __syncthreads(); // every lane is done with the compressed image: reuse it for the output
    if (fast) {
        uint32_t *orow = img + threadIdx.x * OROW;
#pragma unroll
        for (int i = 0; i < CS / 4; ++i)
            orow[i] = outw[i];
    }
    DEC_TRACE(5); // output rows written
    __syncthreads();
    DEC_TRACE(6);
    if (!use_lds)
        return;
    // coalesced copy-out of the whole chunks of the segment
    const uint64_t obase = seg * HF_SEG_CHUNKS * (uint64_t)CS;
    const uint64_t oleft = p.n_out - obase;
    const uint32_t full_chunks = (uint32_t)min(oleft / CS, (uint64_t)HF_SEG_CHUNKS);
    const uint32_t seg_len = full_chunks * CS;
    for (uint32_t o = threadIdx.x * 16u; o < seg_len; o += DEC_THREADS * 16u) {
        const uint32_t ch = o / CS, within = o % CS;
        const uint32_t *srcw = img + ch * OROW + within / 4;
        uint4 v;
        v.x = srcw[0];
        v.y = srcw[1];
        v.z = srcw[2];
        v.w = srcw[3];
        if (p.out_aligned16) {
            *reinterpret_cast<uint4 *>(p.out + obase + o) = v;
        } else {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t i = 0; i < 16u; ++i)
                p.out[obase + o + i] = (uint8_t)(vv[i >> 2] >> (8 * (i & 3)));
        }
    }
    DEC_TRACE(7); // copied out
    if (SPEC) {
        __syncthreads(); // the chunk's (wrong) bytes from the image have been stored
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

// ---------------------------------------------------- gap array from a stream
//
// A stream that arrives without side metadata (a file the reference wrote) is
// cut into sub-sequences of GAP_SUB_BITS bits, one per lane.  Where the first
// code of a sub-sequence starts is only known once its predecessor has been
// decoded, so the lanes guess (offset 0), decode to the end of their
// sub-sequence, and hand the offset at which they leave it to their successor;
// a lane whose start changes decodes again.  Huffman codes resynchronise after
// a few symbols, so almost every lane already leaves its sub-sequence at the
// right offset after the first attempt.  The fixed point of this iteration --
// every lane starts where its predecessor ends, lane 0 at bit 0 -- is exactly
// the sequential decode, whatever the model (also one that is not prefix
// free); a stream that never resynchronises merely needs more rounds.
// Inside a workgroup the rounds run in a loop; across workgroups the host
// relaunches gap_sync_kernel until no workgroup's first start moved.
// Then the symbols of every sub-sequence are counted, scanned, and a last pass
// records the bit position of every chunk_syms-th symbol: the gap array.

constexpr int GAP_THREADS = 256;
constexpr uint32_t GAP_SUB_BITS = 1024;
constexpr uint32_t GAP_IMG_WORDS = GAP_THREADS * GAP_SUB_BITS / 32 + 8; // + look-ahead slack
// LDS slots of the image: one unused slot after every 32 words, so that the
// lanes, whose sub-sequences lie 32 words apart, start on 32 different banks
constexpr uint32_t GAP_IMG_SLOTS = GAP_IMG_WORDS + GAP_IMG_WORDS / 32 + 1;

struct GapWords {
    const uint32_t *w;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return w[i + (i >> 5)]; }
};

struct GapParams {
    DecParams d;         // stream and model (out / chunk arrays unused)
    uint64_t total_bits; // 8 * in_bytes
    uint64_t n_sub;      // sub-sequences
    uint64_t n_wg;       // workgroups = ceil(n_sub / GAP_THREADS)
    uint8_t *start_off;  // [n_sub] offset of the first code inside each sub-sequence
    uint8_t *exit_off;   // [n_sub] offset at which the lane leaves into the next one
    uint32_t *cnt;       // [n_sub] codes that start inside the sub-sequence
    uint32_t *wg_in;     // [n_wg] first start of each workgroup for this launch
    uint32_t *wg_used;   // [n_wg] first start its results were computed with
    uint32_t *changed;   // set by gap_snapshot_kernel when some wg_in != wg_used
    uint64_t *wg_base;   // [n_wg + 1] symbols before each workgroup (after the scan)
    uint32_t *start32;   // [n_chunks + 1] low 32 bits of the chunk start bits
    uint64_t n_chunks;
    uint32_t first;      // first launch: no earlier results to reuse
    uint32_t per_window; // symbols decoded per 64-bit window: per_window * sentinel <= 64
    uint16_t *chunk_bits;
    uint64_t *seg_off;
};

// Entry (len << 8 | sym) of the code at the head of a 64-bit window; 0 with
// `bad` set when the window lies below the smallest key.
__device__ __forceinline__ uint32_t gap_entry(uint64_t win, const uint16_t *s_lut,
                                              const DecParams &p, bool &bad)
{
    uint32_t e = s_lut[(uint32_t)(win >> 32) >> (32 - p.w1)];
    if (__builtin_expect((e & 0x8000u) != 0, 0))
        e = dec_escape_window(win, e, s_lut, p, bad);
    return e;
}

// 64-bit window at image bit `rel` (three words, see GapWords).
__device__ __forceinline__ uint64_t gap_window(const GapWords &src, uint32_t rel)
{
    const uint32_t w = rel >> 5;
    const uint32_t w0 = src(w), w1 = src(w + 1), w2 = src(w + 2);
    const uint32_t sh = 32u - (rel & 31u); // 1..32
    const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
    const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
    return ((uint64_t)hi << 32) | lo;
}

// Decodes from bit `start` of sub-sequence `i` to its end; returns the exit
// offset and the number of codes.  The window is rebuilt every `per_window`
// symbols (per_window * sentinel <= 64).
__device__ __forceinline__ void gap_walk(const GapWords &src, const GapParams &g, uint64_t i,
                                         uint32_t start, const uint16_t *s_lut, uint32_t &exit_off,
                                         uint32_t &count)
{
    const uint64_t left = g.total_bits - i * GAP_SUB_BITS; // bits from the sub-sequence's begin
    const uint32_t end = left < GAP_SUB_BITS ? (uint32_t)left : GAP_SUB_BITS;
    const uint32_t rel0 = (uint32_t)(i % GAP_THREADS) * GAP_SUB_BITS;
    const uint32_t per_window = g.per_window;
    uint32_t pos = start, c = 0;
    bool bad = false;
    while (pos < end) {
        uint64_t win = gap_window(src, rel0 + pos);
#pragma unroll 1
        for (uint32_t k = 0; k < per_window && pos < end; ++k) {
            const uint32_t e = gap_entry(win, s_lut, g.d, bad);
            uint32_t len = e >> 8;
            len = len ? len : 1u; // a wrong guess may land anywhere: keep moving
            win <<= len;
            pos += len;
            ++c;
        }
    }
    exit_off = pos >= GAP_SUB_BITS ? pos - GAP_SUB_BITS : 0u;
    count = c;
}

__global__ __launch_bounds__(GAP_THREADS) void gap_snapshot_kernel(GapParams g)
{
    const uint64_t w = (uint64_t)blockIdx.x * GAP_THREADS + threadIdx.x;
    if (w >= g.n_wg)
        return;
    const uint32_t v = w ? g.exit_off[w * GAP_THREADS - 1] : 0u;
    g.wg_in[w] = v;
    if (v != g.wg_used[w])
        atomicOr(g.changed, 1u);
}

// Stages the 32-bit big-endian words of workgroup `wg`'s bit range into LDS
// (slot of word i: i + i / 32, see GapWords).
__device__ __forceinline__ void gap_stage(const GapParams &g, uint64_t wg, uint32_t *img)
{
    const uint64_t ubase = wg * (GAP_THREADS * GAP_SUB_BITS / 128);
    const uint64_t whole = g.d.in_bytes >> 4;
    const bool aligned = g.d.in_aligned16 != 0;
    for (uint32_t u = threadIdx.x; u < GAP_IMG_WORDS / 4; u += GAP_THREADS) {
        uint4 v;
        if (aligned && ubase + u < whole) {
            v = reinterpret_cast<const uint4 *>(g.d.in)[ubase + u];
            v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
        } else {
            v.x = dec_load_word(g.d.in, g.d.in_bytes, (ubase + u) * 4 + 0);
            v.y = dec_load_word(g.d.in, g.d.in_bytes, (ubase + u) * 4 + 1);
            v.z = dec_load_word(g.d.in, g.d.in_bytes, (ubase + u) * 4 + 2);
            v.w = dec_load_word(g.d.in, g.d.in_bytes, (ubase + u) * 4 + 3);
        }
        uint32_t *dst = img + 4 * u + ((4 * u) >> 5); // the four words share a 32-word row
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    }
}

__device__ __forceinline__ void gap_stage_lut(const DecParams &d, uint8_t *smem)
{
    const uint32_t lut_bytes = (d.lut_entries * 2u + 15u) & ~15u;
    const uint4 *src = reinterpret_cast<const uint4 *>(d.lut);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (uint32_t i = threadIdx.x; i < lut_bytes / 16; i += GAP_THREADS)
        dst[i] = src[i];
}

__global__ __launch_bounds__(GAP_THREADS) void gap_sync_kernel(GapParams g)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    uint32_t *s_exit = img + GAP_IMG_SLOTS;

    const uint64_t wg = blockIdx.x;
    const uint32_t st0 = g.wg_in[wg];
    if (!g.first && st0 == g.wg_used[wg])
        return; // same first start as last time: every result of this workgroup stands
    const uint64_t i = wg * GAP_THREADS + threadIdx.x;
    const bool valid = i < g.n_sub;
    gap_stage_lut(g.d, smem);
    gap_stage(g, wg, img);
    __syncthreads();

    uint32_t start = threadIdx.x == 0 ? st0 : (g.first || !valid ? 0u : g.start_off[i]);
    uint32_t ex = (g.first || !valid) ? 0u : g.exit_off[i];
    uint32_t c = (g.first || !valid) ? 0u : g.cnt[i];
    bool need = valid && (g.first || threadIdx.x == 0);
    while (true) {
        if (need) {
            GapWords src{img};
            gap_walk(src, g, i, start, s_lut, ex, c);
        }
        s_exit[threadIdx.x] = ex;
        __syncthreads();
        const uint32_t ns = threadIdx.x == 0 ? start : s_exit[threadIdx.x - 1];
        need = valid && ns != start;
        start = ns;
        if (!__syncthreads_or(need))
            break;
    }
    if (valid) {
        g.start_off[i] = (uint8_t)start;
        g.exit_off[i] = (uint8_t)ex;
        g.cnt[i] = c;
    }
    if (threadIdx.x == 0)
        g.wg_used[wg] = st0;
}

// Symbols per workgroup, then (one workgroup) their exclusive scan.
__global__ __launch_bounds__(GAP_THREADS) void gap_wgsum_kernel(GapParams g)
{
    __shared__ uint32_t s_wave[GAP_THREADS / 64];
    const uint64_t i = (uint64_t)blockIdx.x * GAP_THREADS + threadIdx.x;
    uint32_t total;
    block_excl_scan256(i < g.n_sub ? g.cnt[i] : 0u, s_wave, total);
    if (threadIdx.x == 0)
        g.wg_base[blockIdx.x] = total;
}

__global__ __launch_bounds__(GAP_THREADS) void gap_scan_kernel(GapParams g)
{
    __shared__ uint64_t s_part[GAP_THREADS];
    // every thread owns a contiguous run of workgroups
    const uint64_t per = (g.n_wg + GAP_THREADS - 1) / GAP_THREADS;
    const uint64_t lo = min(g.n_wg, per * threadIdx.x), hi = min(g.n_wg, lo + per);
    uint64_t sum = 0;
    for (uint64_t w = lo; w < hi; ++w)
        sum += g.wg_base[w];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint32_t k = 0; k < GAP_THREADS; ++k) {
            const uint64_t v = s_part[k];
            s_part[k] = run;
            run += v;
        }
        g.wg_base[g.n_wg] = run;
    }
    __syncthreads();
    uint64_t run = s_part[threadIdx.x];
    for (uint64_t w = lo; w < hi; ++w) {
        const uint64_t v = g.wg_base[w];
        g.wg_base[w] = run;
        run += v;
    }
}

// Last pass: every lane walks its sub-sequence from its final start and
// records the bit position of the symbols that open a chunk.
__global__ __launch_bounds__(GAP_THREADS) void gap_emit_kernel(GapParams g)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    uint32_t *s_wave = img + GAP_IMG_SLOTS;

    const uint64_t wg = blockIdx.x;
    const uint64_t i = wg * GAP_THREADS + threadIdx.x;
    const bool valid = i < g.n_sub;
    gap_stage_lut(g.d, smem);
    gap_stage(g, wg, img);
    uint32_t total;
    const uint32_t excl = block_excl_scan256(valid ? g.cnt[i] : 0u, s_wave, total);
    __syncthreads();
    if (!valid)
        return;
    uint64_t s = g.wg_base[wg] + excl; // index of the lane's first symbol
    if (s >= g.d.n_out)
        return;
    const uint64_t bit0 = i * GAP_SUB_BITS;
    const uint64_t left = g.total_bits - bit0;
    const uint32_t end = left < GAP_SUB_BITS ? (uint32_t)left : GAP_SUB_BITS;
    const uint32_t rel0 = threadIdx.x * GAP_SUB_BITS;
    const uint32_t cs = g.d.chunk_syms;
    uint32_t pos = g.start_off[i];
    uint32_t until = (uint32_t)(s % cs ? cs - s % cs : 0); // symbols until the next chunk opens
    bool bad = false;
    GapWords src{img};
    const uint32_t per_window = g.per_window;
    while (pos < end && s < g.d.n_out) {
        uint64_t win = gap_window(src, rel0 + pos);
#pragma unroll 1
        for (uint32_t k = 0; k < per_window && pos < end && s < g.d.n_out; ++k) {
            if (until == 0) {
                const uint64_t c = s / cs;
                g.start32[c] = (uint32_t)(bit0 + pos);
                if ((c % HF_SEG_CHUNKS) == 0)
                    g.seg_off[c / HF_SEG_CHUNKS] = bit0 + pos;
                until = cs;
            }
            const uint32_t e = gap_entry(win, s_lut, g.d, bad);
            uint32_t len = e >> 8;
            len = len ? len : 1u;
            win <<= len;
            pos += len;
            ++s;
            --until;
            if (s == g.d.n_out)
                g.start32[g.n_chunks] = (uint32_t)(bit0 + pos); // end of the last chunk
        }
    }
    if (bad)
        atomicOr((unsigned long long *)&g.d.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
}

__global__ __launch_bounds__(GAP_THREADS) void gap_diff_kernel(GapParams g)
{
    const uint64_t c = (uint64_t)blockIdx.x * GAP_THREADS + threadIdx.x;
    if (c < g.n_chunks)
        g.chunk_bits[c] = (uint16_t)(g.start32[c + 1] - g.start32[c]);
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

// ===================================================================== host

struct hf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    uint64_t *d_result = nullptr; // result block of the synchronous entry points
    int num_cus = 256;
    int enc_mode = 1;
    int hist_grid = 0;
    int timeout_ms = 2000;
    int debug = 0;
    uint64_t *dec_trace = nullptr; // device buffer for the decode kernel's phase stamps
    // device blocks of destroyed models, kept for the next hf_model_create: a
    // codec that builds a model per stream would otherwise pay hipMalloc/hipFree each time
    struct Blob {
        void *d;
        size_t bytes;
        hipStream_t last_stream; // kernels of the model that owned it ran there
    };
    std::vector<Blob> blob_cache;
    // pinned staging of the table uploads, used in turn: the upload is an
    // asynchronous copy on the context's stream, `done` tells when a buffer
    // may be filled again
    struct Pinned {
        void *h = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    Pinned pinned[4];
    unsigned pinned_next = 0;
    void *arena = nullptr; // device staging of the *_host entry points (grow only)
    size_t arena_bytes = 0;
    // option "kernel_timing": HIP events around the dominant kernel of every
    // histogram / encode / decode call (on the stream it is launched on), kept
    // in a ring until hf_ctx_kernel_ms collects them
    struct Timed {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ring;
        size_t pending = 0; // pairs recorded since the last collection
    };
    Timed timed[3];
    bool kernel_timing = false;
};

struct hf_model {
    hf_ctx *ctx = nullptr;
    void *d_blob = nullptr;
    size_t blob_bytes = 0;
    uint64_t *d_enc = nullptr;   // 256 x (code << 6 | len)
    uint2 *d_enc2 = nullptr;     // 256 x {code, len} ({0, ENC_LONG_FLAG | len} above 32 bits)
    uint8_t *d_len8 = nullptr;   // 256 x len
    uint16_t *d_lut = nullptr;
    uint64_t *d_keys = nullptr;
    uint8_t *d_ksyms = nullptr;
    uint8_t *d_klens = nullptr;
    uint32_t lut_entries = 0;
    uint32_t m = 0;
    uint32_t sentinel = 0;
    uint32_t w1 = 0;
    uint32_t maxlen = 0;
    int enc_group = 1; // codes the encoder merges per append: 4, 2 or 1
    bool enc_spec = false; // a merged group can exceed 32 bits
    bool has_encode = false;
    bool lut_escapes = false; // the decode table holds entries with the escape bit
    bool regular_chunks = false; // 64 symbols are expected to differ by < 8 bits in length
    uint32_t dec_spec_k = 0; // 16, 8, 7 or 6: codes per 64-bit window the decoder may assume (else 0)
    hipStream_t use_stream = nullptr; // the tables were uploaded on this stream ...
    hipEvent_t ready = nullptr;       // ... and are complete once this event has passed
    mutable bool foreign = false;     // kernels on other streams have used the tables
};

namespace {

int fail(hf_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

#define HF_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((ctx), HF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

int ensure_scratch(hf_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes)
        return HF_OK;
    if (ctx->scratch) {
        HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HF_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    size_t want = std::max(bytes, (size_t)1 << 20);
    HF_HIP(ctx, hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return HF_OK;
}

int ensure_arena(hf_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->arena_bytes)
        return HF_OK;
    if (ctx->arena) {
        HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HF_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HF_HIP(ctx, hipMalloc(&ctx->arena, want));
    ctx->arena_bytes = want;
    return HF_OK;
}

int flags_to_status(hf_ctx *ctx, uint64_t flags)
{
    if (flags & HF_FLAG_TIMEOUT)
        return fail(ctx, HF_ERR_INTERNAL, "a device-side inter-workgroup wait expired");
    if (flags & HF_FLAG_PLAN)
        return fail(ctx, HF_ERR_BAD_ARG, "the batch histograms do not belong to this input");
    if (flags & HF_FLAG_CAPACITY)
        return fail(ctx, HF_ERR_CAPACITY, "output buffer too small");
    if (flags & HF_FLAG_DECODE)
        return fail(ctx, HF_ERR_DECODE, "bit window below the smallest key of the model");
    return HF_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int upload_model(hf_ctx *ctx, hf_model *mod, const uint64_t *enc_tab /*256 or null*/,
                 const uint32_t *enc_tab2 /*512*/, const hf::DecodeLut &lut, const uint64_t *keys,
                 const uint8_t *syms, const uint8_t *klens, uint64_t m)
{
    const size_t off_enc = 0;
    const size_t off_enc2 = off_enc + 256 * sizeof(uint64_t);
    const size_t off_len8 = off_enc2 + 512 * sizeof(uint32_t);
    const size_t off_lut = align_up(off_len8 + 256, 16);
    const size_t off_keys = align_up(off_lut + lut.pool.size() * 2, 16);
    const size_t off_ksyms = off_keys + 256 * sizeof(uint64_t);
    const size_t off_klens = off_ksyms + 256;
    const size_t total = off_klens + 256;
    HF_HIP(ctx, hipSetDevice(ctx->device));
    hf_ctx::Pinned &ps = ctx->pinned[ctx->pinned_next++ % 4];
    if (ps.used)
        HF_HIP(ctx, hipEventSynchronize(ps.done)); // the copy that last read this buffer is over
    if (ps.bytes < total) {
        if (ps.h)
            HF_HIP(ctx, hipHostFree(ps.h));
        ps.h = nullptr, ps.bytes = 0;
        HF_HIP(ctx, hipHostMalloc(&ps.h, align_up(total, 4096), hipHostMallocDefault));
        ps.bytes = align_up(total, 4096);
    }
    if (!ps.done)
        HF_HIP(ctx, hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
    uint8_t *blob = static_cast<uint8_t *>(ps.h);
    std::memset(blob, 0, total);
    if (enc_tab) {
        std::memcpy(blob + off_enc, enc_tab, 256 * sizeof(uint64_t));
        std::memcpy(blob + off_enc2, enc_tab2, 512 * sizeof(uint32_t));
        for (int i = 0; i < 256; ++i)
            blob[off_len8 + i] = (uint8_t)(enc_tab[i] & 63u);
    }
    std::memcpy(blob + off_lut, lut.pool.data(), lut.pool.size() * 2);
    std::memcpy(blob + off_keys, keys, m * sizeof(uint64_t));
    std::memcpy(blob + off_ksyms, syms, m);
    std::memcpy(blob + off_klens, klens, m);
    for (size_t i = 0; i < ctx->blob_cache.size(); ++i) {
        if (ctx->blob_cache[i].bytes >= total) {
            // kernels of the block's previous owner may still be running: on
            // this stream the copy queues up behind them, another one is waited for
            if (ctx->blob_cache[i].last_stream != ctx->stream)
                HF_HIP(ctx, hipStreamSynchronize(ctx->blob_cache[i].last_stream));
            mod->d_blob = ctx->blob_cache[i].d;
            mod->blob_bytes = ctx->blob_cache[i].bytes;
            ctx->blob_cache.erase(ctx->blob_cache.begin() + (long)i);
            break;
        }
    }
    if (!mod->d_blob) {
        mod->blob_bytes = align_up(total, 4096);
        HF_HIP(ctx, hipMalloc(&mod->d_blob, mod->blob_bytes));
    }
    HF_HIP(ctx, hipMemcpyAsync(mod->d_blob, blob, total, hipMemcpyHostToDevice, ctx->stream));
    HF_HIP(ctx, hipEventRecord(ps.done, ctx->stream));
    ps.used = true;
    mod->use_stream = ctx->stream;
    mod->ready = ps.done;
    uint8_t *b = static_cast<uint8_t *>(mod->d_blob);
    mod->d_enc = reinterpret_cast<uint64_t *>(b + off_enc);
    mod->d_enc2 = reinterpret_cast<uint2 *>(b + off_enc2);
    mod->d_len8 = b + off_len8;
    mod->d_lut = reinterpret_cast<uint16_t *>(b + off_lut);
    mod->d_keys = reinterpret_cast<uint64_t *>(b + off_keys);
    mod->d_ksyms = b + off_ksyms;
    mod->d_klens = b + off_klens;
    mod->lut_entries = (uint32_t)lut.pool.size();
    mod->m = (uint32_t)m;
    mod->sentinel = lut.sentinel;
    mod->w1 = lut.w1;
    mod->lut_escapes = lut.has_escape;
    // spread of a 64-symbol chunk's bit length when symbols occur with the
    // probabilities 2^-len that the lengths imply
    double kraft = 0, m1 = 0, m2 = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const double w = std::ldexp(1.0, -(int)klens[i]);
        kraft += w, m1 += w * klens[i], m2 += w * klens[i] * klens[i];
    }
    if (kraft > 0) {
        m1 /= kraft, m2 /= kraft;
        mod->regular_chunks = 64.0 * (m2 - m1 * m1) < 64.0; // standard deviation below 8 bits
    }
    mod->dec_spec_k = hf::pick_decode_window(keys, klens, m, lut.sentinel);
    return HF_OK;
}

bool valid_chunk_syms(uint32_t cs)
{
    return cs == 64 || cs == 128 || cs == 256;
}

// Kernel timing (option "kernel_timing"): the next free pair of the ring, or
// nothing when the option is off or the ring is full.
constexpr size_t TIMED_RING = 4096;
std::pair<hipEvent_t, hipEvent_t> *timed_slot(hf_ctx *ctx, int kind)
{
    if (!ctx->kernel_timing)
        return nullptr;
    hf_ctx::Timed &t = ctx->timed[kind];
    if (t.pending >= TIMED_RING)
        return nullptr;
    if (t.pending >= t.ring.size()) {
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess)
            return nullptr;
        t.ring.push_back(ev);
    }
    return &t.ring[t.pending++];
}
#define HF_TIMED_LAUNCH(ctx, kind, launch)                                                    \
    do {                                                                                      \
        std::pair<hipEvent_t, hipEvent_t> *ev_ = timed_slot(ctx, kind);                       \
        if (ev_)                                                                              \
            (void)hipEventRecord(ev_->first, (ctx)->stream);                                  \
        launch;                                                                               \
        if (ev_)                                                                              \
            (void)hipEventRecord(ev_->second, (ctx)->stream);                                 \
    } while (0)

// Work that reads a model's tables must come after their upload: on the
// stream that carried the upload that is the case already, any other stream
// waits for the upload's event (and the model remembers the foreign use).
int model_on_stream(hf_ctx *ctx, const hf_model *model)
{
    if (model->ready && ctx->stream != model->use_stream) {
        HF_HIP(ctx, hipStreamWaitEvent(ctx->stream, model->ready, 0));
        model->foreign = true;
    }
    return HF_OK;
}

} // namespace

extern "C" {

int hf_version(void) { return 1; }

const char *hf_strerror(int status)
{
    switch (status) {
    case HF_OK: return "ok";
    case HF_ERR_BAD_ARG: return "bad argument";
    case HF_ERR_CODELEN: return "code length above 64 (InvalidData)";
    case HF_ERR_EMPTY: return "empty input or model";
    case HF_ERR_CAPACITY: return "output buffer too small";
    case HF_ERR_HIP: return "HIP runtime error";
    case HF_ERR_BAD_CODEBOOK: return "code does not fit its length";
    case HF_ERR_DECODE: return "bit window below the smallest key";
    case HF_ERR_INTERNAL: return "device-side wait expired";
    case HF_ERR_UNSUPPORTED: return "code length or sentinel above 57";
    default: return "unknown status";
    }
}

int hf_ctx_create(int device, hf_ctx **out)
{
    if (!out)
        return HF_ERR_BAD_ARG;
    *out = nullptr;
    hf_ctx *ctx = new (std::nothrow) hf_ctx();
    if (!ctx)
        return HF_ERR_INTERNAL;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        delete ctx;
        return HF_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        ctx->num_cus = prop.multiProcessorCount;
    e = hipMalloc(reinterpret_cast<void **>(&ctx->d_result), HF_RES_WORDS * sizeof(uint64_t));
    if (e != hipSuccess) {
        delete ctx;
        return HF_ERR_HIP;
    }
    *out = ctx;
    return HF_OK;
}

int hf_ctx_destroy(hf_ctx *ctx)
{
    if (!ctx)
        return HF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch)
        (void)hipFree(ctx->scratch);
    if (ctx->arena)
        (void)hipFree(ctx->arena);
    for (auto &b : ctx->blob_cache)
        (void)hipFree(b.d);
    for (auto &t : ctx->timed)
        for (auto &ev : t.ring) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    for (auto &ps : ctx->pinned) {
        if (ps.h)
            (void)hipHostFree(ps.h);
        if (ps.done)
            (void)hipEventDestroy(ps.done);
    }
    if (ctx->d_result)
        (void)hipFree(ctx->d_result);
    delete ctx;
    return HF_OK;
}

int hf_ctx_set_stream(hf_ctx *ctx, void *stream)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    ctx->stream = static_cast<hipStream_t>(stream);
    return HF_OK;
}

int hf_ctx_synchronize(hf_ctx *ctx)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_ctx_kernel_ms(hf_ctx *ctx, int kind, double *sum_ms, uint64_t *launches)
{
    if (!ctx || kind < 0 || kind > 2 || !sum_ms || !launches)
        return HF_ERR_BAD_ARG;
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hf_ctx::Timed &t = ctx->timed[kind];
    double sum = 0;
    for (size_t i = 0; i < t.pending; ++i) {
        float ms = 0;
        HF_HIP(ctx, hipEventElapsedTime(&ms, t.ring[i].first, t.ring[i].second));
        sum += ms;
    }
    *sum_ms = sum;
    *launches = t.pending;
    t.pending = 0;
    return HF_OK;
}

const char *hf_last_error(const hf_ctx *ctx) { return ctx ? ctx->err : "no context"; }

int hf_ctx_set_option(hf_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name)
        return HF_ERR_BAD_ARG;
    if (!std::strcmp(name, "enc_mode") && (value == 0 || value == 1))
        ctx->enc_mode = (int)value;
    else if (!std::strcmp(name, "hist_grid") && value >= 0 && value <= 65536)
        ctx->hist_grid = (int)value;
    else if (!std::strcmp(name, "timeout_ms") && value > 0 && value <= 60000)
        ctx->timeout_ms = (int)value;
    else if (!std::strcmp(name, "kernel_timing"))
        ctx->kernel_timing = value != 0;
    else if (!std::strcmp(name, "dec_trace")) // device address, 128 bytes per segment; 0 = off
        ctx->dec_trace = reinterpret_cast<uint64_t *>((uintptr_t)value);
    else if (!std::strcmp(name, "debug") && value >= 0 && value < 512)
        ctx->debug = (int)value;
    else
        return fail(ctx, HF_ERR_BAD_ARG, "unknown option or value out of range: %s", name);
    return HF_OK;
}

// ---------------------------------------------------------------- histogram

int hf_histogram_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist)
{
    if (!ctx || !d_hist || (!d_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: null pointer");
    if (n >= (1ull << 40))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: n must be below 2^40");
    HF_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
    if (n == 0)
        return HF_OK;
    const uint64_t per_block = (uint64_t)HIST_THREADS * 16 * HIST_UNROLL;
    uint64_t want = (n + per_block - 1) / per_block;
    uint64_t cap = ctx->hist_grid > 0 ? (uint64_t)ctx->hist_grid : (uint64_t)ctx->num_cus * 4;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
    HF_TIMED_LAUNCH(ctx, HF_TIMED_HISTOGRAM,
                    hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream,
                                       d_in, n, reinterpret_cast<unsigned long long *>(d_hist)));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

uint64_t hf_batch_hist_bytes(uint64_t n)
{
    return (n + PLAN_BATCH - 1) / PLAN_BATCH * 256 * sizeof(uint32_t);
}

int hf_histogram_batches_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist,
                               uint32_t *d_batch_hist)
{
    if (!ctx || !d_hist || (!d_in && n) || (!d_batch_hist && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: null pointer");
    if (n >= (1ull << 40))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: n must be below 2^40");
    if ((uintptr_t)d_batch_hist & 15)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: d_batch_hist must be 16-byte aligned");
    HF_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
    if (n == 0)
        return HF_OK;
    const uint64_t n_batches = (n + PLAN_BATCH - 1) / PLAN_BATCH;
    uint64_t cap = ctx->hist_grid > 0 ? (uint64_t)ctx->hist_grid : (uint64_t)ctx->num_cus * 2;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min(n_batches, cap));
    auto kern = ((uintptr_t)d_in & 15) == 0 ? hist_batches_kernel<true> : hist_batches_kernel<false>;
    HF_TIMED_LAUNCH(ctx, HF_TIMED_HISTOGRAM,
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, d_in,
                                       n, reinterpret_cast<unsigned long long *>(d_hist),
                                       d_batch_hist, n_batches));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

int hf_histogram(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t hist[256])
{
    if (!ctx || !hist)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: null pointer");
    int rc = ensure_scratch(ctx, 256 * sizeof(uint64_t));
    if (rc != HF_OK)
        return rc;
    uint64_t *d_hist = static_cast<uint64_t *>(ctx->scratch);
    rc = hf_histogram_async(ctx, d_in, n, d_hist);
    if (rc != HF_OK)
        return rc;
    HF_HIP(ctx, hipMemcpyAsync(hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

// -------------------------------------------------------------------- model

int hf_model_create(hf_ctx *ctx, const uint64_t codes[256], const uint64_t lens[256],
                    hf_model **out)
{
    if (!ctx || !codes || !lens || !out)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_model_create: null pointer");
    *out = nullptr;
    uint64_t maxlen = 0;
    for (int i = 0; i < 256; ++i) {
        if (lens[i] > 64)
            return fail(ctx, HF_ERR_CODELEN, "symbol %d: code length %llu > 64", i,
                        (unsigned long long)lens[i]);
        if (lens[i] > HF_MAX_CODELEN)
            return fail(ctx, HF_ERR_UNSUPPORTED, "symbol %d: code length %llu > %u", i,
                        (unsigned long long)lens[i], HF_MAX_CODELEN);
        if (lens[i] < 64 && (codes[i] >> lens[i]) != 0)
            return fail(ctx, HF_ERR_BAD_CODEBOOK, "symbol %d: code does not fit %llu bits", i,
                        (unsigned long long)lens[i]);
        maxlen = std::max(maxlen, lens[i]);
    }
    if (maxlen == 0)
        return fail(ctx, HF_ERR_EMPTY, "model has no symbol with a code");
    bool speculative = false;
    const int group = hf::pick_merge_width(lens, &speculative);
    uint64_t enc[256];
    uint32_t enc2[512];
    for (int i = 0; i < 256; ++i) {
        enc[i] = (codes[i] << 6) | lens[i];
        const bool fits = lens[i] <= 32;
        enc2[2 * i + 0] = fits ? (uint32_t)codes[i] : 0u;
        enc2[2 * i + 1] = fits ? (uint32_t)lens[i] : (ENC_LONG_FLAG | (uint32_t)lens[i]);
    }
    uint64_t keys[256], m = 0;
    uint8_t syms[256], klens[256];
    int rc = hf::model_table(codes, lens, keys, syms, klens, &m);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_create: cannot build the key table");
    hf::DecodeLut lut;
    rc = hf::build_decode_lut(keys, syms, klens, m, maxlen, &lut);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_create: cannot build the decode table");
    hf_model *mod = new (std::nothrow) hf_model();
    if (!mod)
        return HF_ERR_INTERNAL;
    mod->ctx = ctx;
    mod->maxlen = (uint32_t)maxlen;
    mod->has_encode = true;
    mod->enc_group = group;
    mod->enc_spec = speculative;
    rc = upload_model(ctx, mod, enc, enc2, lut, keys, syms, klens, m);
    if (rc != HF_OK) {
        delete mod;
        return rc;
    }
    *out = mod;
    return HF_OK;
}

int hf_model_from_table(hf_ctx *ctx, const uint64_t *keys, const uint8_t *syms,
                        const uint8_t *klens, uint64_t m, uint64_t sentinel, hf_model **out)
{
    if (!ctx || !keys || !syms || !klens || !out)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_model_from_table: null pointer");
    *out = nullptr;
    if (m == 0 || m > 256)
        return fail(ctx, HF_ERR_EMPTY, "hf_model_from_table: %llu entries", (unsigned long long)m);
    if (sentinel == 0 || sentinel > HF_MAX_SENTINEL)
        return fail(ctx, HF_ERR_UNSUPPORTED, "sentinel %llu outside 1..%u",
                    (unsigned long long)sentinel, HF_MAX_SENTINEL);
    for (uint64_t i = 0; i < m; ++i) {
        if (i && keys[i] <= keys[i - 1])
            return fail(ctx, HF_ERR_BAD_ARG, "keys must be strictly ascending");
        if (sentinel < 64 && (keys[i] >> sentinel) != 0)
            return fail(ctx, HF_ERR_BAD_CODEBOOK, "key %llu wider than the sentinel",
                        (unsigned long long)i);
        if (klens[i] > HF_MAX_CODELEN)
            return fail(ctx, HF_ERR_UNSUPPORTED, "entry %llu: length above %u",
                        (unsigned long long)i, HF_MAX_CODELEN);
    }
    hf::DecodeLut lut;
    int rc = hf::build_decode_lut(keys, syms, klens, m, sentinel, &lut);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_from_table: cannot build the decode table");
    hf_model *mod = new (std::nothrow) hf_model();
    if (!mod)
        return HF_ERR_INTERNAL;
    mod->ctx = ctx;
    mod->maxlen = (uint32_t)sentinel;
    mod->has_encode = false;
    rc = upload_model(ctx, mod, nullptr, nullptr, lut, keys, syms, klens, m);
    if (rc != HF_OK) {
        delete mod;
        return rc;
    }
    *out = mod;
    return HF_OK;
}

int hf_model_destroy(hf_model *model)
{
    if (!model)
        return HF_OK;
    if (model->d_blob) {
        (void)hipSetDevice(model->ctx->device);
        // Kernels that read the tables may still be queued.  The block is kept
        // for the next model, whose upload is ordered behind them on the same
        // stream; only tables that other streams have used need a full wait.
        if (model->foreign)
            (void)hipDeviceSynchronize();
        if (model->ctx->blob_cache.size() < 4) {
            model->ctx->blob_cache.push_back({model->d_blob, model->blob_bytes, model->use_stream});
        } else {
            (void)hipStreamSynchronize(model->use_stream);
            (void)hipFree(model->d_blob);
        }
    }
    delete model;
    return HF_OK;
}

// ------------------------------------------------------------------- encode

uint64_t hf_encode_bound(const hf_model *model, uint64_t n, uint32_t lead_bits)
{
    const uint64_t maxlen = model ? model->maxlen : 64;
    const uint64_t bytes = (lead_bits + n * maxlen + 7) / 8;
    return (bytes + 3) & ~3ull; // whole 32-bit words keep every store on the fast path
}

int hf_gap_layout(uint64_t n, uint32_t chunk_syms, uint64_t *n_chunks, uint64_t *n_segs)
{
    if (!valid_chunk_syms(chunk_syms) || !n_chunks || !n_segs)
        return HF_ERR_BAD_ARG;
    *n_chunks = (n + chunk_syms - 1) / chunk_syms;
    *n_segs = (*n_chunks + HF_SEG_CHUNKS - 1) / HF_SEG_CHUNKS;
    return HF_OK;
}

} // extern "C"

namespace {

// hf_encode_async and hf_encode_planned_async (d_batch_hist != null)
int encode_launch(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                  uint32_t lead_bits, const uint32_t *d_batch_hist, uint8_t *d_out,
                  uint64_t out_cap, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                  uint64_t *d_seg_off, uint64_t *d_result)
{
    if (!ctx || !model || !d_result || (!d_in && n) || (!d_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: null pointer");
    if (!model->has_encode)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: model was built for decoding only");
    if (int rc0 = model_on_stream(ctx, model))
        return rc0;
    if (lead_bits > 7)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: lead_bits must be 0..7");
    if ((uintptr_t)d_out & 3)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: d_out must be 4-byte aligned");
    if ((d_chunk_bits == nullptr) != (d_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: pass both gap arrays or neither");
    if (d_chunk_bits && !valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: chunk_syms must be 64, 128 or 256");
    if (n >= (1ull << 45))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: n too large");
    if (!d_batch_hist || n == 0) // (the planned path's first kernel clears the block itself)
        HF_HIP(ctx, hipMemsetAsync(d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n == 0) {
        const uint64_t bytes = (lead_bits + 7) / 8;
        if (bytes > out_cap)
            return fail(ctx, HF_ERR_CAPACITY, "hf_encode: output buffer too small");
        if (bytes)
            HF_HIP(ctx, hipMemsetAsync(d_out, 0, bytes, ctx->stream));
        return HF_OK;
    }
    const uint64_t n_tiles64 = (n + ENC_TILE - 1) / ENC_TILE;
    const uint32_t n_tiles = (uint32_t)n_tiles64;

    EncParams p{};
    p.in = d_in;
    p.n = n;
    p.tab = model->d_enc;
    p.tab2 = model->d_enc2;
    p.len8 = model->d_len8;
    p.out = d_out;
    p.out_cap = out_cap;
    p.lead_bits = lead_bits;
    p.aligned16 = ((uintptr_t)d_in & 15) == 0;
    p.out_aligned16 = ((uintptr_t)d_out & 15) == 0;
    p.chunk_syms = d_chunk_bits ? chunk_syms : 64;
    p.chunk_bits = d_chunk_bits;
    p.seg_off = d_seg_off;
    p.n_chunks = d_chunk_bits ? (n + chunk_syms - 1) / chunk_syms : 0;
    p.result = d_result;
    p.n_tiles = n_tiles;
    p.timeout_ticks = (uint64_t)ctx->timeout_ms * 100000ull;
    p.debug = (uint32_t)ctx->debug & 31u;
#ifdef HF_ENC_TRACE
    if (out_cap < ((uint64_t)n_tiles / ENC_BATCH + 3) * ENC_TRACE_BYTES)
#endif
        p.debug &= ~16u; // the phase trace overwrites the tail of the output buffer

    const size_t need = ((size_t)n_tiles + 4) * sizeof(uint64_t);
    int rc = ensure_scratch(ctx, need);
    if (rc != HF_OK)
        return rc;
    // codes merged per append, and whether the table flags codes too long for that
    const int g = model->enc_group;
    const bool flagged = model->enc_spec;
    // every merged group stays below 32 bits: the carry sink applies
    const bool carry = !flagged && g > 1 && (uint32_t)g * model->maxlen <= 31u;
    const uint32_t n_batches = (n_tiles + ENC_BATCH - 1) / ENC_BATCH;
    // batches whose tiles are all complete run the lean instantiation; the
    // ragged last batch (and everything, if the input is unaligned) the general one
    const uint32_t n_lean = p.aligned16 ? (uint32_t)(n / ((uint64_t)ENC_TILE * ENC_BATCH)) : 0u;
    const bool planned = d_batch_hist != nullptr;
    const bool chained = !planned && ctx->enc_mode != 0;
    if (planned) {
        // scratch: group offsets (+ total) | start inside the group | bit total, per batch
        const uint32_t n_groups = (n_batches + ENC_WAVES - 1) / ENC_WAVES;
        uint64_t *off = static_cast<uint64_t *>(ctx->scratch);
        uint32_t *local = reinterpret_cast<uint32_t *>(off + n_groups + 1);
        uint32_t *tot = local + n_batches;
        hipLaunchKernelGGL(enc_plan_bits_kernel, dim3(n_groups), dim3(ENC_THREADS), 0, ctx->stream,
                           d_batch_hist, model->d_len8, off, local, tot, n_batches, d_result);
        hipLaunchKernelGGL(enc_scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, off,
                           n_groups, d_result);
        p.plan_off = off;
        p.plan_local = local;
        p.plan_tot = tot;
    } else if (!chained) {
        p.tile_off = static_cast<uint64_t *>(ctx->scratch);
        hipLaunchKernelGGL(enc_tile_bits_kernel, dim3(n_tiles), dim3(ENC_THREADS), 0, ctx->stream,
                           p);
        hipLaunchKernelGGL(enc_scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, p.tile_off,
                           n_tiles, d_result);
    } else {
        HF_HIP(ctx, hipMemsetAsync(ctx->scratch, 0, need, ctx->stream));
        p.desc = static_cast<unsigned long long *>(ctx->scratch);
        p.ticket = reinterpret_cast<uint32_t *>(p.desc + n_batches);
        p.abort_flag = p.ticket + 2;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const bool lean = pass == 0;
        const uint32_t grid = lean ? n_lean : n_batches - n_lean;
        if (grid == 0)
            continue;
        p.first_batch = lean ? 0u : n_lean;
        void (*kern)(EncParams) = nullptr;
#define HF_PICK(G)                                                                            \
    (flagged ? (lean ? enc_pack_kernel<G, true, true, false>                                  \
                     : enc_pack_kernel<G, true, false, false>)                                \
             : carry ? (lean ? enc_pack_kernel<G, false, true, (G > 1)>                       \
                             : enc_pack_kernel<G, false, false, (G > 1)>)                     \
                     : (lean ? enc_pack_kernel<G, false, true, false>                         \
                             : enc_pack_kernel<G, false, false, false>))
        kern = g == 4 ? HF_PICK(4) : (g == 2 ? HF_PICK(2) : HF_PICK(1));
#undef HF_PICK
        // (only the launch over the complete batches is timed: the ragged rest is one workgroup)
        if (lean)
            HF_TIMED_LAUNCH(ctx, HF_TIMED_ENCODE,
                            hipLaunchKernelGGL(kern, dim3(grid), dim3(ENC_THREADS), 0, ctx->stream, p));
        else
            hipLaunchKernelGGL(kern, dim3(grid), dim3(ENC_THREADS), 0, ctx->stream, p);
        if (chained && lean) // the follow-up launch numbers its own workgroups from zero again
            p.ticket += 1;
    }
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

} // namespace

extern "C" {

int hf_encode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                    uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
                    uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint64_t *d_result)
{
    return encode_launch(ctx, model, d_in, n, lead_bits, nullptr, d_out, out_cap, chunk_syms,
                         d_chunk_bits, d_seg_off, d_result);
}

int hf_encode_planned_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                            uint32_t lead_bits, const uint32_t *d_batch_hist, uint8_t *d_out,
                            uint64_t out_cap, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                            uint64_t *d_seg_off, uint64_t *d_result)
{
    if (!d_batch_hist && n)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_planned: null batch histograms");
    if ((uintptr_t)d_batch_hist & 15)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_planned: d_batch_hist must be 16-byte aligned");
    return encode_launch(ctx, model, d_in, n, lead_bits, d_batch_hist, d_out, out_cap, chunk_syms,
                         d_chunk_bits, d_seg_off, d_result);
}

int hf_encode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
              uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
              uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint64_t *total_bits)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    int rc = hf_encode_async(ctx, model, d_in, n, lead_bits, d_out, out_cap, chunk_syms,
                             d_chunk_bits, d_seg_off, ctx->d_result);
    if (rc != HF_OK)
        return rc;
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total_bits)
        *total_bits = res[HF_RES_TOTAL_BITS];
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

int hf_encoder_write_value(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, const uint64_t codes[256],
                           const uint64_t lens[256], uint64_t state[3], uint64_t *ret)
{
    if (!ctx || !codes || !lens || !state || !ret || (!d_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encoder_write_value: null pointer");
    if (n > HF_WRITE_VALUE_MAX)
        return fail(ctx, HF_ERR_UNSUPPORTED, "hf_encoder_write_value: more than %llu symbols",
                    (unsigned long long)HF_WRITE_VALUE_MAX);
    const size_t bytes = (512 + 3 + 2) * sizeof(uint64_t);
    int rc = ensure_scratch(ctx, bytes);
    if (rc != HF_OK)
        return rc;
    uint64_t host[512 + 3 + 2];
    std::memcpy(host, codes, 256 * sizeof(uint64_t));
    std::memcpy(host + 256, lens, 256 * sizeof(uint64_t));
    std::memcpy(host + 512, state, 3 * sizeof(uint64_t));
    host[515] = host[516] = 0;
    uint64_t *d = static_cast<uint64_t *>(ctx->scratch);
    HF_HIP(ctx, hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(enc_automaton_kernel, dim3(1), dim3(64), 0, ctx->stream, d_in, n, d, d + 256,
                       d + 512, d + 515);
    HF_HIP(ctx, hipGetLastError());
    HF_HIP(ctx, hipMemcpyAsync(host + 512, d + 512, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (host[516])
        return fail(ctx, HF_ERR_CODELEN, "Codelen > 64");
    std::memcpy(state, host + 512, 3 * sizeof(uint64_t));
    *ret = host[515];
    return HF_OK;
}

// ------------------------------------------------------------------- decode

static int fill_dec_params(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                           uint64_t in_bytes, uint64_t bit_base, uint64_t n_out, uint8_t *d_out,
                           uint64_t *d_result, DecParams *p)
{
    if (!ctx || !model || !d_result || (!d_out && n_out) || (!d_in && in_bytes))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: null pointer");
    if ((uintptr_t)d_in & 3)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: d_in must be 4-byte aligned");
    if (int rc0 = model_on_stream(ctx, model))
        return rc0;
    p->in = d_in;
    p->in_bytes = in_bytes;
    p->bit_base = bit_base;
    p->n_out = n_out;
    p->out = d_out;
    p->result = d_result;
    p->lut = model->d_lut;
    p->lut_entries = model->lut_entries;
    p->w1 = model->w1;
    p->sentinel = model->sentinel;
    p->keys = model->d_keys;
    p->ksyms = model->d_ksyms;
    p->klens = model->d_klens;
    p->m = model->m;
    p->out_aligned16 = ((uintptr_t)d_out & 15) == 0;
    p->trace = ctx->dec_trace;
    p->in_aligned16 = ((uintptr_t)d_in & 15) == 0;
    return HF_OK;
}

int hf_decode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                    uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
                    const uint16_t *d_chunk_bits, const uint64_t *d_seg_off, uint8_t *d_out,
                    uint64_t *d_result)
{
    DecParams p{};
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, bit_base, n_out, d_out, d_result, &p);
    if (rc != HF_OK)
        return rc;
    if (!valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: chunk_syms must be 64, 128 or 256");
    if (n_out && (!d_chunk_bits || !d_seg_off))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: gap arrays missing (use hf_decode_serial)");
    HF_HIP(ctx, hipMemsetAsync(d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n_out == 0)
        return HF_OK;
    p.chunk_syms = chunk_syms;
    p.chunk_bits = d_chunk_bits;
    p.seg_off = d_seg_off;
    p.n_chunks = (n_out + chunk_syms - 1) / chunk_syms;
    const uint64_t n_segs = (p.n_chunks + HF_SEG_CHUNKS - 1) / HF_SEG_CHUNKS;
    // LDS: table + compressed image (sized for 8 bits per symbol; denser
    // segments read from global memory instead) + padded output image.
    const uint32_t lut_bytes = (model->lut_entries * 2u + 15u) & ~15u;
    // symbols decoded per refill test: the reader guarantees 33 valid bits
    const int spr = model->sentinel <= 8 ? 4 : (model->sentinel <= 16 ? 2 : 1);
    void (*kern)(DecParams) = nullptr;
    size_t lds = 0;
    const bool reg128 = chunk_syms == 128 && !model->lut_escapes;
    const bool reg256 = chunk_syms == 256 && !model->lut_escapes;
    if ((chunk_syms == 64 || reg128 || reg256) && !(ctx->debug & 256)) {
        // register-output kernel: one LDS image serves as input, then as output
        const uint32_t in_raw = HF_SEG_CHUNKS * chunk_syms / 4 + 16;
        const uint32_t in_words = in_raw + in_raw / 32 + 1; // padded slots, see SwzWords
        const uint32_t out_words = HF_SEG_CHUNKS * (chunk_syms / 4 + 1);
        p.istg_words = std::max(in_words, out_words);
        lds = lut_bytes + (size_t)(p.istg_words + 4) * 4;
        // symbols per window rebuild: K * sentinel <= 64, or the model's speculative K
        const uint32_t k = 64u / std::max<uint32_t>(model->sentinel, 1u);
        if (reg256) // 64 output registers per lane, two work-groups per CU
            kern = model->regular_chunks
                       ? (k >= 6 ? dec_kernel_reg<256, 6, false, true, false>
                                 : dec_kernel_reg<256, 5, false, true, false>)
                       : (k >= 6 ? dec_kernel_reg<256, 6, false, false, false>
                                 : dec_kernel_reg<256, 5, false, false, false>);
        else if (reg128)
            kern = model->regular_chunks
                       ? (k >= 6 ? dec_kernel_reg<128, 6, false, true, false>
                                 : dec_kernel_reg<128, 5, false, true, false>)
                       : (k >= 6 ? dec_kernel_reg<128, 6, false, false, false>
                                 : dec_kernel_reg<128, 5, false, false, false>);
        else if (model->dec_spec_k > k) {
            const uint32_t sk = model->dec_spec_k;
            if (model->lut_escapes)
                kern = sk >= 16 ? dec_kernel_reg<64, 16, true, false, true>
                                : (sk >= 8 ? dec_kernel_reg<64, 8, true, false, true>
                                           : (sk == 7 ? dec_kernel_reg<64, 7, true, false, true>
                                                      : dec_kernel_reg<64, 6, true, false, true>));
            else
                kern = sk >= 16 ? dec_kernel_reg<64, 16, false, false, true>
                                : (sk >= 8 ? dec_kernel_reg<64, 8, false, false, true>
                                           : (sk == 7 ? dec_kernel_reg<64, 7, false, false, true>
                                                      : dec_kernel_reg<64, 6, false, false, true>));
        } else if (model->lut_escapes)
            kern = k >= 4 ? dec_kernel_reg<64, 4, true, false, false>
                          : (k == 3 ? dec_kernel_reg<64, 3, true, false, false>
                                    : (k == 2 ? dec_kernel_reg<64, 2, true, false, false>
                                              : dec_kernel_reg<64, 1, true, false, false>));
        else if (model->regular_chunks)
            kern = k >= 8 ? dec_kernel_reg<64, 8, false, true, false>
                          : (k >= 6 ? dec_kernel_reg<64, 6, false, true, false>
                                    : dec_kernel_reg<64, 5, false, true, false>);
        else
            kern = k >= 8 ? dec_kernel_reg<64, 8, false, false, false>
                          : (k >= 6 ? dec_kernel_reg<64, 6, false, false, false>
                                    : dec_kernel_reg<64, 5, false, false, false>);
    } else {
        p.istg_words = HF_SEG_CHUNKS * chunk_syms / 4 + 16;
        const uint32_t ostg_words = HF_SEG_CHUNKS * (chunk_syms / 4 + 1);
        lds = lut_bytes + (size_t)(p.istg_words + ostg_words + 4) * 4;
        if (model->lut_escapes)
            kern = spr == 4 ? dec_kernel<4, true>
                            : (spr == 2 ? dec_kernel<2, true> : dec_kernel<1, true>);
        else
            kern = spr == 4 ? dec_kernel<4, false>
                            : (spr == 2 ? dec_kernel<2, false> : dec_kernel<1, false>);
    }
    if (lds > 160 * 1024 - 64)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: chunk_syms %u needs %zu bytes of LDS",
                    chunk_syms, lds);
    HF_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HF_TIMED_LAUNCH(ctx, HF_TIMED_DECODE,
                    hipLaunchKernelGGL(kern, dim3((uint32_t)n_segs), dim3(DEC_THREADS), lds,
                                       ctx->stream, p));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

int hf_decode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
              uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
              const uint16_t *d_chunk_bits, const uint64_t *d_seg_off, uint8_t *d_out)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    int rc = hf_decode_async(ctx, model, d_in, in_bytes, bit_base, n_out, chunk_syms, d_chunk_bits,
                             d_seg_off, d_out, ctx->d_result);
    if (rc != HF_OK)
        return rc;
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

int hf_decode_serial(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                     uint64_t bit_base, uint64_t n_out, uint8_t *d_out)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    DecParams p{};
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, bit_base, n_out, d_out, ctx->d_result, &p);
    if (rc != HF_OK)
        return rc;
    if (bit_base != 0)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_serial: bit_base must be 0");
    HF_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n_out) {
        hipLaunchKernelGGL(dec_serial_kernel, dim3(1), dim3(64), 0, ctx->stream, p);
        HF_HIP(ctx, hipGetLastError());
    }
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

int hf_build_gap_array(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                       uint64_t n_out, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                       uint64_t *d_seg_off)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    if (!valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: chunk_syms must be 64, 128 or 256");
    if (n_out && (!d_chunk_bits || !d_seg_off))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: null output");
    GapParams g{};
    // d.out is not written; any non-null value passes the shared argument checks
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, 0, n_out,
                             reinterpret_cast<uint8_t *>(d_chunk_bits), ctx->d_result, &g.d);
    if (rc != HF_OK)
        return rc;
    if (n_out == 0)
        return HF_OK;
    if (in_bytes == 0 || in_bytes >= (1ull << 58))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: empty or oversized stream");
    g.d.chunk_syms = chunk_syms;
    g.per_window = std::max(1u, 64u / std::max(1u, g.d.sentinel));
    g.total_bits = in_bytes * 8;
    g.n_sub = (g.total_bits + GAP_SUB_BITS - 1) / GAP_SUB_BITS;
    g.n_wg = (g.n_sub + GAP_THREADS - 1) / GAP_THREADS;
    g.n_chunks = (n_out + chunk_syms - 1) / chunk_syms;
    g.chunk_bits = d_chunk_bits;
    g.seg_off = d_seg_off;
    if (g.n_wg >= (1ull << 31))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: stream too long");

    // scratch: start_off | exit_off | cnt | wg_in | wg_used | changed | wg_base | start32
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t o_start = take(g.n_sub), o_exit = take(g.n_sub), o_cnt = take(g.n_sub * 4);
    const size_t o_in = take(g.n_wg * 4), o_used = take(g.n_wg * 4), o_chg = take(4);
    const size_t o_base = take((g.n_wg + 1) * 8), o_s32 = take((g.n_chunks + 1) * 4);
    rc = ensure_scratch(ctx, off);
    if (rc != HF_OK)
        return rc;
    uint8_t *b = static_cast<uint8_t *>(ctx->scratch);
    g.start_off = b + o_start;
    g.exit_off = b + o_exit;
    g.cnt = reinterpret_cast<uint32_t *>(b + o_cnt);
    g.wg_in = reinterpret_cast<uint32_t *>(b + o_in);
    g.wg_used = reinterpret_cast<uint32_t *>(b + o_used);
    g.changed = reinterpret_cast<uint32_t *>(b + o_chg);
    g.wg_base = reinterpret_cast<uint64_t *>(b + o_base);
    g.start32 = reinterpret_cast<uint32_t *>(b + o_s32);
    HF_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    HF_HIP(ctx, hipMemsetAsync(g.exit_off, 0, g.n_sub, ctx->stream));
    HF_HIP(ctx, hipMemsetAsync(g.wg_used, 0xff, g.n_wg * 4, ctx->stream));

    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    const size_t lds = lut_bytes + (size_t)GAP_IMG_SLOTS * 4 + GAP_THREADS * 4;
    const dim3 wgs((uint32_t)g.n_wg), one_per_wg((uint32_t)((g.n_wg + GAP_THREADS - 1) / GAP_THREADS));
    // every round fixes at least the first workgroup whose first start was
    // wrong, so n_wg + 1 rounds always suffice; real streams need two or three
    uint64_t round = 0;
    for (;; ++round) {
        if (round > g.n_wg + 1)
            return fail(ctx, HF_ERR_INTERNAL, "hf_build_gap_array: no fixed point");
        HF_HIP(ctx, hipMemsetAsync(g.changed, 0, 4, ctx->stream));
        hipLaunchKernelGGL(gap_snapshot_kernel, one_per_wg, dim3(GAP_THREADS), 0, ctx->stream, g);
        uint32_t changed = 0;
        HF_HIP(ctx, hipMemcpyAsync(&changed, g.changed, 4, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!changed)
            break;
        g.first = round == 0;
        hipLaunchKernelGGL(gap_sync_kernel, wgs, dim3(GAP_THREADS), lds, ctx->stream, g);
        HF_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(gap_wgsum_kernel, wgs, dim3(GAP_THREADS), 0, ctx->stream, g);
    hipLaunchKernelGGL(gap_scan_kernel, dim3(1), dim3(GAP_THREADS), 0, ctx->stream, g);
    uint64_t total_syms = 0;
    HF_HIP(ctx, hipMemcpyAsync(&total_syms, g.wg_base + g.n_wg, 8, hipMemcpyDeviceToHost,
                               ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total_syms < n_out)
        return fail(ctx, HF_ERR_DECODE,
                    "hf_build_gap_array: the stream holds %llu symbols, fewer than n_out = %llu",
                    (unsigned long long)total_syms, (unsigned long long)n_out);
    hipLaunchKernelGGL(gap_emit_kernel, wgs, dim3(GAP_THREADS), lds, ctx->stream, g);
    hipLaunchKernelGGL(gap_diff_kernel, dim3((uint32_t)((g.n_chunks + GAP_THREADS - 1) / GAP_THREADS)),
                       dim3(GAP_THREADS), 0, ctx->stream, g);
    HF_HIP(ctx, hipGetLastError());
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

// ------------------------------------------------------- host-buffer forms

int hf_histogram_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t hist[256])
{
    if (!ctx || !hist || (!h_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_host: null pointer");
    int rc = ensure_arena(ctx, n + 16);
    if (rc != HF_OK)
        return rc;
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(ctx->arena, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    return hf_histogram(ctx, static_cast<const uint8_t *>(ctx->arena), n, hist);
}

int hf_encode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t n,
                   uint32_t lead_bits, uint8_t *h_out, uint64_t out_cap, uint32_t chunk_syms,
                   uint16_t *h_chunk_bits, uint64_t *h_seg_off, uint64_t *total_bits)
{
    if (!ctx || !model || (!h_in && n) || (!h_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: pass both gap arrays or neither");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: chunk_syms must be 64, 128 or 256");
    // arena layout: input | stream | chunk_bits | seg_off, each 256-byte aligned
    const uint64_t bound = std::min<uint64_t>(hf_encode_bound(model, n, lead_bits),
                                              (out_cap + 3) & ~3ull);
    const size_t off_out = align_up(n + 16, 256);
    const size_t off_cb = align_up(off_out + bound + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t total = off_so + n_segs * 8 + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(a, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t bits = 0;
    rc = hf_encode(ctx, model, a, n, lead_bits, a + off_out, std::min<uint64_t>(bound, out_cap),
                   chunk_syms, h_chunk_bits ? reinterpret_cast<uint16_t *>(a + off_cb) : nullptr,
                   h_seg_off ? reinterpret_cast<uint64_t *>(a + off_so) : nullptr, &bits);
    if (total_bits)
        *total_bits = bits;
    if (rc != HF_OK)
        return rc;
    const uint64_t nbytes = (lead_bits + bits + 7) / 8;
    if (nbytes > out_cap)
        return fail(ctx, HF_ERR_CAPACITY, "hf_encode_host: output buffer too small");
    if (nbytes)
        HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_chunk_bits && n_chunks) {
        HF_HIP(ctx, hipMemcpyAsync(h_chunk_bits, a + off_cb, n_chunks * 2, hipMemcpyDeviceToHost,
                                   ctx->stream));
        HF_HIP(ctx, hipMemcpyAsync(h_seg_off, a + off_so, n_segs * 8, hipMemcpyDeviceToHost,
                                   ctx->stream));
    }
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_compress_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t codes[256],
                     uint64_t lens[256], uint8_t *h_out, uint64_t out_cap, uint32_t chunk_syms,
                     uint16_t *h_chunk_bits, uint64_t *h_seg_off, uint64_t *total_bits)
{
    if (!ctx || !codes || !lens || !total_bits || (!h_in && n) || (!h_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: pass both gap arrays or neither");
    if (n == 0)
        return fail(ctx, HF_ERR_EMPTY, "hf_compress_host: empty input");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: chunk_syms must be 64, 128 or 256");
    // A Huffman code never needs more than the 8 bits per symbol of the plain
    // bytes, so n bytes (rounded up to the store width) always hold the stream.
    // arena: input | histogram | batch histograms | stream | chunk_bits | seg_off
    const uint64_t bound = std::min<uint64_t>((n + 15) & ~15ull, (out_cap + 3) & ~3ull);
    const size_t off_hist = align_up(n + 16, 256);
    const size_t off_bh = align_up(off_hist + 256 * 8, 256);
    const size_t off_out = align_up(off_bh + hf_batch_hist_bytes(n), 256);
    const size_t off_cb = align_up(off_out + bound + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t total = off_so + n_segs * 8 + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    uint64_t *d_hist = reinterpret_cast<uint64_t *>(a + off_hist);
    uint32_t *d_bh = reinterpret_cast<uint32_t *>(a + off_bh);
    HF_HIP(ctx, hipMemcpyAsync(a, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    rc = hf_histogram_batches_async(ctx, a, n, d_hist, d_bh);
    if (rc != HF_OK)
        return rc;
    uint64_t hist[256];
    HF_HIP(ctx, hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = hf_build_codebook(hist, 256, codes, lens);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_compress_host: cannot build the codebook");
    hf_model *model = nullptr;
    rc = hf_model_create(ctx, codes, lens, &model);
    if (rc != HF_OK)
        return rc;
    rc = hf_encode_planned_async(ctx, model, a, n, 0, d_bh, a + off_out,
                                 std::min<uint64_t>(bound, out_cap), chunk_syms,
                                 h_chunk_bits ? reinterpret_cast<uint16_t *>(a + off_cb) : nullptr,
                                 h_seg_off ? reinterpret_cast<uint64_t *>(a + off_so) : nullptr,
                                 ctx->d_result);
    uint64_t res[HF_RES_WORDS] = {0};
    if (rc == HF_OK) {
        // the stream's length is known from the histogram: its download is
        // queued behind the encoder without waiting for the result block
        uint64_t bits = 0;
        for (int i = 0; i < 256; ++i)
            bits += hist[i] * lens[i];
        const uint64_t nbytes = (bits + 7) / 8;
        if (nbytes > out_cap) {
            rc = fail(ctx, HF_ERR_CAPACITY, "hf_compress_host: output buffer too small");
        } else {
            hipError_t e = hipMemcpyAsync(h_out, a + off_out, nbytes, hipMemcpyDeviceToHost,
                                          ctx->stream);
            if (e == hipSuccess && h_chunk_bits && n_chunks) {
                e = hipMemcpyAsync(h_chunk_bits, a + off_cb, n_chunks * 2, hipMemcpyDeviceToHost,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(h_seg_off, a + off_so, n_segs * 8, hipMemcpyDeviceToHost,
                                       ctx->stream);
            }
            if (e == hipSuccess)
                e = hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost,
                                   ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess)
                rc = fail(ctx, HF_ERR_HIP, "hf_compress_host: %s", hipGetErrorString(e));
            else if ((rc = flags_to_status(ctx, res[HF_RES_FLAGS])) == HF_OK &&
                     res[HF_RES_TOTAL_BITS] != bits)
                rc = fail(ctx, HF_ERR_INTERNAL, "hf_compress_host: stream length mismatch");
            *total_bits = bits;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    hf_model_destroy(model);
    return rc;
}

int hf_decode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t in_bytes,
                   uint64_t n_out, uint32_t chunk_syms, const uint16_t *h_chunk_bits,
                   const uint64_t *h_seg_off, uint8_t *h_out)
{
    if (!ctx || !model || (!h_in && in_bytes) || (!h_out && n_out))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: pass both gap arrays or neither");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n_out, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: chunk_syms must be 64, 128 or 256");
    const bool rebuild = !h_chunk_bits && n_out > 0;
    if (rebuild) { // no side metadata: the device reconstructs it from the stream
        chunk_syms = 64;
        hf_gap_layout(n_out, chunk_syms, &n_chunks, &n_segs);
    }
    const size_t off_out = align_up(in_bytes + 16, 256);
    const size_t off_cb = align_up(off_out + n_out + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t total = off_so + n_segs * 8 + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (in_bytes)
        HF_HIP(ctx, hipMemcpyAsync(a, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint16_t *d_cb = reinterpret_cast<uint16_t *>(a + off_cb);
    uint64_t *d_so = reinterpret_cast<uint64_t *>(a + off_so);
    bool serial = false;
    if (rebuild) {
        rc = in_bytes ? hf_build_gap_array(ctx, model, a, in_bytes, n_out, chunk_syms, d_cb, d_so)
                      : HF_ERR_DECODE;
        // a stream that ends early is continued with zero bits, as the reference's
        // drain loop does (src/huffman/decode/mod.rs:219-229): the sequential walk
        if (rc == HF_ERR_DECODE && n_out <= HF_WRITE_VALUE_MAX)
            serial = true;
        else if (rc != HF_OK)
            return rc;
    } else if (n_chunks) {
        HF_HIP(ctx, hipMemcpyAsync(d_cb, h_chunk_bits, n_chunks * 2, hipMemcpyHostToDevice,
                                   ctx->stream));
        HF_HIP(ctx, hipMemcpyAsync(d_so, h_seg_off, n_segs * 8, hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    if (serial)
        rc = hf_decode_serial(ctx, model, a, in_bytes, 0, n_out, a + off_out);
    else if (n_out)
        rc = hf_decode(ctx, model, a, in_bytes, 0, n_out, chunk_syms, d_cb, d_so, a + off_out);
    if (rc != HF_OK)
        return rc;
    if (n_out)
        HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, n_out, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_encoder_write_value_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n,
                                const uint64_t codes[256], const uint64_t lens[256],
                                uint64_t state[3], uint64_t *ret)
{
    if (!ctx || (!h_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encoder_write_value_host: null pointer");
    if (n > HF_WRITE_VALUE_MAX)
        return fail(ctx, HF_ERR_UNSUPPORTED, "hf_encoder_write_value_host: more than %llu symbols",
                    (unsigned long long)HF_WRITE_VALUE_MAX);
    int rc = ensure_arena(ctx, n + 16);
    if (rc != HF_OK)
        return rc;
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(ctx->arena, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    return hf_encoder_write_value(ctx, static_cast<const uint8_t *>(ctx->arena), n, codes, lens,
                                  state, ret);
}

} // extern "C"
