// Decoding a bare stream (files the reference wrote carry no side metadata):
// the gap array rebuilt on the device, and the stream decoded in the same pass.
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

// ------------------------------------------------ a stream without gap array
//
// The stream is cut into sub-sequences of GAP_SUB_BITS bits, one per lane.
// Where the first code of a sub-sequence starts is only known once its
// predecessor has been walked, so the lanes guess, walk to the end of their
// sub-sequence, and hand the offset at which they leave it to their successor;
// a lane whose start changes walks again.  Huffman codes resynchronise after a
// few symbols, so a lane leaves its sub-sequence at the right offset nearly
// whatever its start was.  Two things follow from that:
//  * the first attempt of a lane only walks the last GAP_PROBE_BITS bits of
//    its sub-sequence (a quarter of the work): all it is good for is the exit
//    offset;
//  * a work-group also walks the last sub-sequence of its predecessor (lane 0,
//    from offset 0) and takes that lane's exit as the start of its own first
//    sub-sequence, instead of waiting for the predecessor's result.
// Neither is trusted: inside a work-group the walks are repeated until every
// lane starts where its predecessor ends, and after the launch every
// work-group's assumed first start is compared with what its predecessor
// really computed (gap_verify_kernel); work-groups that assumed wrong run again
// with the right value.  The fixed point -- every lane starts where its
// predecessor ends, the first at the stream's first code -- is exactly the
// sequential decode, whatever the model (also one that is not prefix free); a
// stream that never resynchronises merely needs more rounds.
// Then the symbols of every sub-sequence are counted and scanned, and a last
// walk either records the bit position of every chunk_syms-th symbol -- the
// gap array (gap_emit_kernel) -- or writes the symbols themselves
// (gap_decode_kernel): symbol number and bit position of every lane's first
// code are known by then, so the stream is decoded in that walk, without gap
// array or second kernel.

constexpr int GAP_THREADS = 256;
constexpr uint32_t GAP_OWN = GAP_THREADS - 1; // sub-sequences a work-group owns (lane 0 walks the predecessor's last)
#ifndef HF_GAP_SUB_BITS // (timing experiments build other sizes: tools/build_variant.py)
#define HF_GAP_SUB_BITS 512
#endif
#ifndef HF_GAP_PROBE_BITS
#define HF_GAP_PROBE_BITS 192
#endif
constexpr uint32_t GAP_SUB_BITS = HF_GAP_SUB_BITS;
constexpr uint32_t GAP_PROBE_BITS = HF_GAP_PROBE_BITS;
static_assert(GAP_SUB_BITS % 128 == 0 && GAP_SUB_BITS >= 256 && GAP_PROBE_BITS <= GAP_SUB_BITS,
              "sub-sequences are whole 16-byte units, longer than the longest code's exit offset");
// LDS image of a work-group's bit range: a row per lane with the sub-sequence's
// words, the two words behind it (a code that starts in the sub-sequence's
// last bit ends within them) and one unused slot, which makes the row length odd:
// the lanes' reads of equal word numbers then fall on different banks.  A lane
// addresses its window by row base + word number, nothing else.
constexpr uint32_t GAP_SUB_WORDS = GAP_SUB_BITS / 32;
constexpr uint32_t GAP_ROW = GAP_SUB_WORDS + 3;
static_assert(GAP_ROW % 2 == 1, "odd row length");
constexpr uint32_t GAP_IMG_SLOTS = GAP_THREADS * GAP_ROW + 16; // (+ what unchecked steps read past the last row)

struct GapParams {
    DecParams d;         // stream and model; d.out = where gap_decode_kernel writes the symbols
    uint64_t total_bits; // 8 * in_bytes
    uint64_t n_sub;      // sub-sequences
    uint64_t n_wg;       // workgroups = ceil(n_sub / GAP_OWN)
    uint8_t *start_off;  // [n_sub] offset of the first code inside each sub-sequence
    uint8_t *exit_off;   // [n_sub] offset at which the lane leaves into the next one
    uint32_t *cnt;       // [n_sub] codes that start inside the sub-sequence
    uint32_t *wg_in;     // [n_wg] first start of each workgroup as its predecessor computed it
    uint32_t *wg_used;   // [n_wg] first start its results were computed with
    uint32_t *changed;   // set by gap_verify_kernel when some wg_in != wg_used
    uint64_t *wg_base;   // [n_wg + 1] symbols before each workgroup (after the scan)
    uint32_t *blk_total; // [ceil(n_wg / GAP_SCAN_BLOCK)] symbols per block of work-groups (the scan's second level)
    uint32_t *start32;   // [n_chunks + 1] low 32 bits of the chunk start bits
    uint64_t n_chunks;
    uint32_t first;      // first launch: no earlier results, first starts assumed from the overlap lane
    uint32_t first_start; // bit of the stream's first sub-sequence at which its first code starts
    uint32_t safe_w;     // codes a 64-bit window always holds (64 / sentinel): the checked steps' window width
    uint64_t bit_base;   // stream bit of bit 0 of d.in (added to the segment offsets)
    uint16_t *chunk_bits;
    uint64_t *seg_off;
};

// Entry (sym << 8 | len) of the code at the head of a 64-bit window.  A window
// below the smallest key sets `bad`; its entry has length 1, so a walk that
// guessed its start wrong (and may land anywhere) keeps moving.  ESC = the
// table has entries with the escape bit (codes behind the first level, key
// search, invalid windows); tables without them are not even tested.
template <bool ESC>
__device__ __forceinline__ uint32_t gap_entry(uint64_t win, const uint16_t *s_lut,
                                              const DecParams &p, bool &bad)
{
    uint32_t e = s_lut[(uint32_t)(win >> 32) >> (32 - p.w1)];
    if (ESC && __builtin_expect((e & HF_LUT_ESC) != 0, 0)) {
        e = dec_escape_window(win, e, s_lut, p, bad);
        e = HF_LUT_LEN(e) ? e : 1u;
    }
    return e;
}

// 64-bit window at bit `pos` of the lane's row (three words).
__device__ __forceinline__ uint64_t gap_window(const uint32_t *row, uint32_t pos)
{
    const uint32_t *w = row + (pos >> 5);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    const uint32_t sh = 32u - (pos & 31u); // 1..32
    const uint32_t hi = (uint32_t)((((uint64_t)w0 << 32) | w1) >> sh);
    const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w2) >> sh);
    return ((uint64_t)hi << 32) | lo;
}

// Sub-sequence of lane `lane` of work-group `wg` (-1 for lane 0 of work-group 0).
__device__ __forceinline__ int64_t gap_sub_of(uint64_t wg, uint32_t lane)
{
    return (int64_t)(wg * GAP_OWN) + (int64_t)lane - 1;
}

// Walks the lane's row from bit `start` to `end` (the sub-sequence's end):
// returns the bit behind the last code that starts before `end` and adds the
// number of those codes to `count`.  W codes are taken from one window without
// looking at the bounds.  W may be more than a window always holds (the host
// picks it for the model's typical code lengths, W * sentinel < 256): a group
// that took more than 64 bits, like one whose last code starts behind `end`,
// is done again with a look at every code and a window per `safe_w` codes.
template <int W, bool ESC>
__device__ __forceinline__ uint32_t gap_walk(const uint32_t *row, uint32_t start, uint32_t end,
                                             const uint16_t *s_lut, const DecParams &p, uint32_t safe_w,
                                             uint32_t &count, bool &bad)
{
    uint32_t pos = start, c = 0;
    // Windows of W codes, unchecked, for as long as they stand; a window that does not
    // (the lane's last one as a rule, rarely codes that exceed the 64 bits) is done again
    // code by code -- behind the inner loop, where the lanes of a wave meet again: inside
    // it, every lane that reaches its end would take the whole wave through that path.
    while (pos < end) {
        bool ok = true;
        uint64_t win = 0;
        while (ok && pos < end) {
            win = gap_window(row, pos);
            uint32_t esum = 0, before_last = 0; // low byte: bits taken (at most 64)
            bool bad_w = false; // (only counts when the window stands: it may look past the bounds)
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const uint32_t e = gap_entry<ESC>(win, s_lut, p, bad_w);
                win <<= (e & 63u);
                before_last = esum;
                esum += e;
            }
            ok = pos + (before_last & 0xffu) < end && (esum & 0xffu) <= 64u;
            if (ok) {
                pos += esum & 0xffu;
                c += W;
                bad = bad || bad_w;
            }
        }
        if (!ok) {
            uint32_t since = 0; // codes taken from the current window (the same in all lanes)
#pragma unroll
            for (int k = 0; k < W; ++k) {
                if (since == 0)
                    win = gap_window(row, pos);
                since = since + 1 == safe_w ? 0 : since + 1;
                if (pos < end) {
                    const uint32_t e = gap_entry<ESC>(win, s_lut, p, bad);
                    win <<= (e & 63u);
                    pos += HF_LUT_LEN(e);
                    ++c;
                }
            }
        }
    }
    count += c;
    return pos;
}

// After a launch of gap_sync_kernel: the first start every work-group should
// have used (its predecessor's last exit, as that work-group computed it from
// a start of its own that was checked the same way), against the one it used.
__global__ __launch_bounds__(GAP_THREADS) void gap_verify_kernel(GapParams g)
{
    const uint64_t w = (uint64_t)blockIdx.x * GAP_THREADS + threadIdx.x;
    if (w >= g.n_wg)
        return;
    const uint32_t v = w ? g.exit_off[w * GAP_OWN - 1] : g.first_start;
    g.wg_in[w] = v;
    if (v != g.wg_used[w])
        atomicOr(g.changed, 1u);
}

// Stages the 32-bit big-endian words of workgroup `wg`'s bit range into the
// rows.  The range starts one sub-sequence before the work-group's own ones;
// what lies before the stream (work-group 0) or behind it reads as zero.
__device__ __forceinline__ void gap_stage(const GapParams &g, uint64_t wg, uint32_t *img)
{
    constexpr uint32_t SUB_UNITS = GAP_SUB_BITS / 128; // 16-byte units per sub-sequence
    const int64_t ubase = ((int64_t)(wg * GAP_OWN) - 1) * SUB_UNITS;
    const int64_t whole = (int64_t)(g.d.in_bytes >> 4);
    const bool aligned = g.d.in_aligned16 != 0;
    // one unit more than the rows hold: its first two words end the last row
    for (uint32_t u = threadIdx.x; u < GAP_THREADS * SUB_UNITS + 1; u += GAP_THREADS) {
        const int64_t gu = ubase + u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gu >= 0) {
            if (aligned && gu < whole) {
                v = reinterpret_cast<const uint4 *>(g.d.in)[gu];
                v.x = bswap32(v.x), v.y = bswap32(v.y), v.z = bswap32(v.z), v.w = bswap32(v.w);
            } else {
                v.x = dec_load_word(g.d.in, g.d.in_bytes, (uint64_t)gu * 4 + 0);
                v.y = dec_load_word(g.d.in, g.d.in_bytes, (uint64_t)gu * 4 + 1);
                v.z = dec_load_word(g.d.in, g.d.in_bytes, (uint64_t)gu * 4 + 2);
                v.w = dec_load_word(g.d.in, g.d.in_bytes, (uint64_t)gu * 4 + 3);
            }
        }
        const uint32_t r = u / SUB_UNITS, part = u % SUB_UNITS;
        if (r < GAP_THREADS) {
            uint32_t *dst = img + r * GAP_ROW + 4 * part;
            dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
        }
        if (part == 0 && r > 0) { // the two words behind the previous row's sub-sequence
            uint32_t *la = img + (r - 1) * GAP_ROW + GAP_SUB_WORDS;
            la[0] = v.x, la[1] = v.y;
        }
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

// Bits of sub-sequence j that belong to the stream (GAP_SUB_BITS except for the last one).
__device__ __forceinline__ uint32_t gap_sub_end(const GapParams &g, uint64_t j)
{
    const uint64_t left = g.total_bits - j * GAP_SUB_BITS;
    return left < GAP_SUB_BITS ? (uint32_t)left : GAP_SUB_BITS;
}

template <int W, bool ESC>
__global__ __launch_bounds__(GAP_THREADS) void gap_sync_kernel(GapParams g)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    uint32_t *s_exit = img + GAP_IMG_SLOTS;

    if (W == 5 && !ESC)
        HF_TOUCH_VGPR("v24"); // (24 registers otherwise, a shift amount in the last one)
    const uint64_t wg = blockIdx.x;
    const uint32_t st_in = g.first ? 0u : g.wg_in[wg];
    if (!g.first && st_in == g.wg_used[wg])
        return; // the first start it used was right: every result of this workgroup stands
    const uint32_t lane = threadIdx.x;
    const int64_t js = gap_sub_of(wg, lane);
    const bool exists = js >= 0 && (uint64_t)js < g.n_sub;
    const uint64_t j = exists ? (uint64_t)js : 0;
    const bool own = exists && lane > 0;
    gap_stage_lut(g.d, smem);
    gap_stage(g, wg, img);
    __syncthreads();

    const uint32_t *row = img + lane * GAP_ROW;
    const uint32_t end = exists ? gap_sub_end(g, j) : 0u;
    bool bad = false; // (walks from guessed starts may meet anything: not reported here)
    uint32_t start = 0, ex = 0, c = 0;
    bool need;
    if (g.first) {
        // first attempts: the overlap lane walks its whole sub-sequence from offset 0,
        // the others only the end of theirs; none of the counts is kept
        if (exists) {
            const uint32_t from = (lane == 0 || end <= GAP_PROBE_BITS) ? 0u : end - GAP_PROBE_BITS;
            uint32_t unused = 0;
            const uint32_t pos = gap_walk<W, ESC>(row, from, end, s_lut, g.d, g.safe_w, unused, bad);
            ex = pos >= GAP_SUB_BITS ? pos - GAP_SUB_BITS : 0u;
        }
        s_exit[lane] = ex;
        __syncthreads();
        // the stream's first sub-sequence starts where the caller says; everybody
        // else where the predecessor's first attempt left
        start = (wg == 0 && lane == 1) ? g.first_start : (lane ? s_exit[lane - 1] : 0u);
        need = own;
        __syncthreads();
    } else {
        start = lane == 1 ? st_in : (own ? g.start_off[j] : 0u);
        ex = own ? g.exit_off[j] : 0u;
        c = own ? g.cnt[j] : 0u;
        need = own && lane == 1;
    }
    const uint32_t first_used = __shfl(start, 1); // (lane 1 is in wave 0; read below by lane 0 of that wave)
    while (true) {
        if (need) {
            c = 0;
            const uint32_t pos = gap_walk<W, ESC>(row, start, end, s_lut, g.d, g.safe_w, c, bad);
            ex = pos >= GAP_SUB_BITS ? pos - GAP_SUB_BITS : 0u;
        }
        s_exit[lane] = ex;
        __syncthreads();
        // (lane 1's start is this launch's assumption about the predecessor work-group:
        // it is checked after the launch, not here)
        const uint32_t ns = lane <= 1 ? start : s_exit[lane - 1];
        need = own && ns != start;
        start = ns;
        if (!__syncthreads_or(need))
            break;
    }
    if (own) {
        g.start_off[j] = (uint8_t)start;
        g.exit_off[j] = (uint8_t)ex;
        g.cnt[j] = c;
    }
    if (lane == 0)
        g.wg_used[wg] = first_used;
}

// Symbols per workgroup, then (one workgroup) their exclusive scan.
__global__ __launch_bounds__(GAP_THREADS) void gap_wgsum_kernel(GapParams g)
{
    __shared__ uint32_t s_wave[GAP_THREADS / 64];
    const int64_t js = gap_sub_of(blockIdx.x, threadIdx.x);
    const bool own = threadIdx.x > 0 && js >= 0 && (uint64_t)js < g.n_sub;
    uint32_t total;
    block_excl_scan256(own ? g.cnt[js] : 0u, s_wave, total);
    if (threadIdx.x == 0)
        g.wg_base[blockIdx.x] = total;
}

// Exclusive scan of the work-groups' symbol counts (wg_base, in place; the total
// goes behind the last entry) in two launches of many work-groups: each scans a
// block of GAP_SCAN_BLOCK counts on its own and leaves the block's total, then
// each adds what the blocks before it hold.  (A block's counts sum to less than
// 2^32: 4096 work-groups of at most 255 x 512 codes.)
constexpr uint32_t GAP_SCAN_PER = 16; // counts per lane
constexpr uint32_t GAP_SCAN_BLOCK = GAP_THREADS * GAP_SCAN_PER;

__global__ __launch_bounds__(GAP_THREADS) void gap_scan_local_kernel(GapParams g)
{
    __shared__ uint32_t s_wave[GAP_THREADS / 64];
    const uint64_t lo = (uint64_t)blockIdx.x * GAP_SCAN_BLOCK + threadIdx.x * GAP_SCAN_PER;
    uint32_t v[GAP_SCAN_PER];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < GAP_SCAN_PER; ++k) {
        v[k] = lo + k < g.n_wg ? (uint32_t)g.wg_base[lo + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan256(sum, s_wave, total);
#pragma unroll
    for (uint32_t k = 0; k < GAP_SCAN_PER; ++k) {
        if (lo + k < g.n_wg)
            g.wg_base[lo + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0)
        g.blk_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(GAP_THREADS) void gap_scan_add_kernel(GapParams g)
{
    __shared__ uint64_t s_part[GAP_THREADS];
    uint64_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += GAP_THREADS)
        part += g.blk_total[b];
    s_part[threadIdx.x] = part;
    __syncthreads();
    for (uint32_t step = GAP_THREADS / 2; step > 0; step >>= 1) {
        if (threadIdx.x < step)
            s_part[threadIdx.x] += s_part[threadIdx.x + step];
        __syncthreads();
    }
    const uint64_t base = s_part[0];
    const uint64_t lo = (uint64_t)blockIdx.x * GAP_SCAN_BLOCK + threadIdx.x * GAP_SCAN_PER;
    if (base) {
#pragma unroll
        for (uint32_t k = 0; k < GAP_SCAN_PER; ++k)
            if (lo + k < g.n_wg)
                g.wg_base[lo + k] += base;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        g.wg_base[g.n_wg] = base + g.blk_total[blockIdx.x];
}

// What the last walk of a lane starts from: its sub-sequence, the number of
// its first symbol (from the scan) and the offset of its first code.  Lanes
// without a sub-sequence of their own, or whose first symbol lies behind the
// output, have nothing to do (`active` false).
struct GapLane {
    bool active;
    uint64_t s;    // index of the lane's first symbol
    uint64_t bit0; // bit of d.in at which the sub-sequence starts
    uint32_t pos, end;
    const uint32_t *row; // the lane's words in the LDS image
};

// Stages table and stream image and works out GapLane (all lanes of the
// work-group must call it: it holds the barriers).
__device__ __forceinline__ GapLane gap_last_walk_setup(const GapParams &g, uint8_t *smem, uint32_t *img,
                                                       uint32_t *s_wave)
{
    const uint64_t wg = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const int64_t js = gap_sub_of(wg, lane);
    const bool own = lane > 0 && js >= 0 && (uint64_t)js < g.n_sub;
    const uint64_t j = own ? (uint64_t)js : 0;
    gap_stage_lut(g.d, smem);
    gap_stage(g, wg, img);
    uint32_t total;
    const uint32_t excl = block_excl_scan256(own ? g.cnt[j] : 0u, s_wave, total);
    __syncthreads();
    GapLane L;
    L.s = g.wg_base[wg] + excl;
    L.active = own && L.s < g.d.n_out;
    L.bit0 = j * GAP_SUB_BITS;
    L.end = gap_sub_end(g, j);
    L.row = img + lane * GAP_ROW;
    L.pos = own ? g.start_off[j] : 0u;
    return L;
}

// Last walk, decoding: every lane walks its sub-sequence from its final start
// and stores the symbols; their place in the output is known from the scan.
// Sixteen codes are taken per step (W of them from one window, see gap_walk)
// and leave as one 16-byte store at the symbols' own
// address, whatever its alignment -- the lanes of a wave write 64 different
// places, which costs a store instruction sixteen times its aligned price, but
// there is one per sixteen symbols.  A step that would reach the end of the
// sub-sequence or of the output is done one code at a time.
// GAP_HOLD = steps whose stores leave together (see below): 1, or 8 for steep codes,
// whose lanes take a dozen steps and more.
template <int W, bool ESC, int GAP_HOLD = 1>
__global__ __launch_bounds__(GAP_THREADS) void gap_decode_kernel(GapParams g)
{
    if (W == 6 && !ESC && GAP_HOLD == 1) // (this instantiation allocates exactly 32 VGPRs, see dev_common.hpp)
        HF_TOUCH_VGPR("v32");
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    const GapLane L = gap_last_walk_setup(g, smem, img, img + GAP_IMG_SLOTS);
    if (!L.active)
        return;
    const uint32_t *row = L.row;
    const uint64_t n_out = g.d.n_out;
    uint8_t *out = g.d.out;
    uint64_t s = L.s;
    uint32_t pos = L.pos;
    const uint32_t end = L.end;
    bool bad = false;
    while (pos < end && s < n_out) {
        // Up to GAP_HOLD steps of sixteen codes each are taken before anything is stored,
        // and their 16-byte stores then leave back to back: with eight, 128 contiguous
        // bytes reach the L2 together and are written to memory once.  (A store per step,
        // a microsecond apart, with every lane of the chip writing somewhere else in
        // between, has most lines written back several times, partly filled: 3.4 bytes
        // written per symbol of the 4 GiB geometric stream, whose last walk that bounds.
        // Streams of longer codes have five or six steps per lane; for them the grouping
        // costs more than it saves, and they keep one step per round.)  A lane's steps
        // stand from the first on, so `held` counts them.
        // The rounds go on for as long as their steps stand; the step that does not (a
        // lane's last, as a rule) is done again code by code behind the inner loop, where
        // the lanes of a wave meet again.
        bool failed = false;
        uint32_t pack[4] = {0, 0, 0, 0};
        while (!failed && pos < end && s < n_out) {
        uint4 hold[GAP_HOLD];
        uint32_t held = 0;
#pragma unroll
        for (int q = 0; q < GAP_HOLD; ++q) {
            if (!failed && pos < end && s + 16u * held < n_out) {
                uint32_t used = 0, start_last = 0, wstart = 0;
                uint64_t win = 0;
                bool bad16 = false; // (only counts when the step stands: it may read past the bounds)
                bool over = false;  // some window gave more than its 64 bits
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k % W == 0) {
                        over = over || used - wstart > 64u;
                        wstart = used;
                        win = gap_window(row, pos + used);
                    }
                    const uint32_t e = gap_entry<ESC>(win, s_lut, g.d, bad16);
                    win <<= (e & 63u);
                    start_last = used;
                    used += HF_LUT_LEN(e);
                    // byte k % 4 of the word takes byte 1 of the entry
                    constexpr uint32_t keep = 0x03020100u;
                    const uint32_t sel = (keep & ~(0xffu << (8 * (k & 3)))) | (0x05u << (8 * (k & 3)));
                    pack[k >> 2] = __builtin_amdgcn_perm(e, pack[k >> 2], sel);
                }
                over = over || used - wstart > 64u;
                if (pos + start_last < end && s + 16u * held + 16u <= n_out && !over) {
                    hold[q] = make_uint4(pack[0], pack[1], pack[2], pack[3]);
                    pos += used;
                    ++held;
                    bad = bad || bad16;
                } else {
                    failed = true;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < GAP_HOLD; ++q)
            if ((uint32_t)q < held)
                __builtin_memcpy(out + s + 16u * q, &hold[q], 16);
        s += 16u * held;
        }
        if (failed) {
            // the same sixteen codes, each behind a look at the bounds (every lane comes
            // here once, at its end -- but its wave with it, in several of its steps)
            uint32_t nv = 0;    // codes that stand
            uint32_t since = 0; // codes taken from the current window (the same in all lanes)
            uint64_t win = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (since == 0)
                    win = gap_window(row, pos);
                since = since + 1 == g.safe_w ? 0 : since + 1;
                if (pos < end && s + nv < n_out) {
                    const uint32_t e = gap_entry<ESC>(win, s_lut, g.d, bad);
                    win <<= (e & 63u);
                    pos += HF_LUT_LEN(e);
                    constexpr uint32_t keep = 0x03020100u;
                    const uint32_t sel = (keep & ~(0xffu << (8 * (k & 3)))) | (0x05u << (8 * (k & 3)));
                    pack[k >> 2] = __builtin_amdgcn_perm(e, pack[k >> 2], sel);
                    ++nv;
                }
            }
            // they leave as pieces of 16 (a step that was only done again because a window
            // overflowed), 8, 4, 2 and 1 bytes (a byte store per symbol would keep the store
            // unit busier than all the full steps together)
            if (nv & 16u) {
                const uint4 v = make_uint4(pack[0], pack[1], pack[2], pack[3]);
                __builtin_memcpy(out + s, &v, 16);
                s += 16;
            }
            uint32_t p0 = pack[0], p1 = pack[1];
            if (nv & 8u) {
                const uint2 v = make_uint2(p0, p1);
                __builtin_memcpy(out + s, &v, 8);
                p0 = pack[2], p1 = pack[3];
                s += 8;
            }
            if (nv & 4u) {
                __builtin_memcpy(out + s, &p0, 4);
                p0 = p1;
                s += 4;
            }
            if (nv & 2u) {
                const uint16_t h = (uint16_t)p0;
                __builtin_memcpy(out + s, &h, 2);
                p0 >>= 16;
                s += 2;
            }
            if (nv & 1u) {
                out[s] = (uint8_t)p0;
                s += 1;
            }
        }
        if (s == n_out)
            g.d.result[HF_RES_END_BIT] = L.bit0 + pos; // (counted from bit 0 of d.in)
    }
    if (bad)
        atomicOr((unsigned long long *)&g.d.result[HF_RES_FLAGS], (unsigned long long)HF_FLAG_DECODE);
}

// Last walk, gap array: the bit position of every symbol that opens a chunk.
// Four codes per step, unchecked; at most one of them opens a chunk (a chunk
// has 64 symbols or more), and its position is picked from the four starts.
template <int W, bool ESC>
__global__ __launch_bounds__(GAP_THREADS) void gap_emit_kernel(GapParams g)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + lut_bytes);
    const GapLane L = gap_last_walk_setup(g, smem, img, img + GAP_IMG_SLOTS);
    if (!L.active)
        return;
    const uint32_t *row = L.row;
    const uint64_t n_out = g.d.n_out;
    const uint32_t cs = g.d.chunk_syms;
    uint64_t s = L.s;
    uint32_t pos = L.pos;
    const uint32_t end = L.end;
    bool bad = false;
    auto open_chunk = [&](uint64_t sym, uint32_t at) {
        const uint64_t c = sym / cs;
        g.start32[c] = (uint32_t)(L.bit0 + at);
        if ((c % HF_SEG_CHUNKS) == 0)
            g.seg_off[c / HF_SEG_CHUNKS] = g.bit_base + L.bit0 + at;
    };
    while (pos < end && s < n_out) {
        // unchecked steps for as long as they stand; the one that does not is done again
        // code by code behind the inner loop, where the lanes of a wave meet again
        bool ok = true;
        uint64_t win = 0;
        while (ok && pos < end && s < n_out) {
            uint32_t st[4], used = 0, wstart = 0;
            bool bad4 = false; // (only counts when the step stands: it may read past the bounds)
            bool over = false; // some window gave more than its 64 bits
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k % W == 0) {
                    over = over || used - wstart > 64u;
                    wstart = used;
                    win = gap_window(row, pos + used);
                }
                const uint32_t e = gap_entry<ESC>(win, s_lut, g.d, bad4);
                win <<= (e & 63u);
                st[k] = used;
                used += HF_LUT_LEN(e);
            }
            over = over || used - wstart > 64u;
            ok = pos + st[3] < end && s + 4 <= n_out && !over;
            if (ok) {
                const uint32_t k_open = (cs - ((uint32_t)s & (cs - 1u))) & (cs - 1u); // 0..3: s + k_open opens a chunk
                if (k_open < 4u)
                    open_chunk(s + k_open, pos + (k_open == 0 ? st[0] : k_open == 1 ? st[1] : k_open == 2 ? st[2] : st[3]));
                pos += used;
                s += 4;
                bad = bad || bad4;
            }
        }
        if (!ok) {
            uint32_t since = 0; // codes taken from the current window (the same in all lanes)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (since == 0)
                    win = gap_window(row, pos);
                since = since + 1 == g.safe_w ? 0 : since + 1;
                if (pos < end && s < n_out) {
                    if (((uint32_t)s & (cs - 1u)) == 0)
                        open_chunk(s, pos);
                    const uint32_t e = gap_entry<ESC>(win, s_lut, g.d, bad);
                    win <<= (e & 63u);
                    pos += HF_LUT_LEN(e);
                    ++s;
                }
            }
        }
        if (s == n_out) {
            g.start32[g.n_chunks] = (uint32_t)(L.bit0 + pos); // end of the last chunk
            g.d.result[HF_RES_END_BIT] = L.bit0 + pos;        // (counted from bit 0 of d.in)
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
