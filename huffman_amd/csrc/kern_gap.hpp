// Gap array rebuilt from a bare stream (no counterpart in the reference: its files carry no side metadata).
// Part of the single translation unit huffman_amd/csrc/hfgpu.hip (device code
// for gfx950, wave64); included there, not compiled on its own.
#ifndef HFGPU_KERN_GAP_HPP
#define HFGPU_KERN_GAP_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "hfgpu.h"
#include "dev_common.hpp"
#include "kern_decode.hpp"

namespace {

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
    uint32_t first_start; // bit of the stream's first sub-sequence at which its first code starts
    uint64_t bit_base;   // stream bit of bit 0 of d.in (added to the segment offsets)
    uint32_t per_window; // symbols decoded per 64-bit window: per_window * sentinel <= 64
    uint16_t *chunk_bits;
    uint64_t *seg_off;
};

// Entry (sym << 8 | len) of the code at the head of a 64-bit window; 0 with
// `bad` set when the window lies below the smallest key.
__device__ __forceinline__ uint32_t gap_entry(uint64_t win, const uint16_t *s_lut,
                                              const DecParams &p, bool &bad)
{
    uint32_t e = s_lut[(uint32_t)(win >> 32) >> (32 - p.w1)];
    if (__builtin_expect((e & HF_LUT_ESC) != 0, 0))
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
            uint32_t len = HF_LUT_LEN(e);
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
    const uint32_t v = w ? g.exit_off[w * GAP_THREADS - 1] : g.first_start;
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
                    g.seg_off[c / HF_SEG_CHUNKS] = g.bit_base + bit0 + pos;
                until = cs;
            }
            const uint32_t e = gap_entry(win, s_lut, g.d, bad);
            uint32_t len = HF_LUT_LEN(e);
            len = len ? len : 1u;
            win <<= len;
            pos += len;
            ++s;
            --until;
            if (s == g.d.n_out) {
                g.start32[g.n_chunks] = (uint32_t)(bit0 + pos); // end of the last chunk
                g.d.result[HF_RES_END_BIT] = bit0 + pos;        // (counted from bit 0 of d.in)
            }
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

} // namespace

#endif // HFGPU_KERN_GAP_HPP
