// Histogram kernels (src/stats.rs:13-28, src/huffman/mod.rs:59-65).
// Part of the single translation unit huffman_amd/csrc/hfgpu.hip (device code
// for gfx950, wave64); included there, not compiled on its own.
#ifndef HFGPU_KERN_HIST_HPP
#define HFGPU_KERN_HIST_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "hfgpu.h"
#include "dev_common.hpp"

namespace {

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
constexpr int HIST_VECS = 8; // 16-byte loads in flight per lane
constexpr uint64_t HIST_STREAM_MIN_BYTES = 256ull << 20; // inputs the last-level cache cannot hold

// 16 bytes of input.  STREAM marks the load non-temporal: an input far larger
// than the caches is read at a higher rate that way (1 GiB: 5.6 instead of
// 5.2 TB/s); a small one is loaded normally so that the encoder, which reads
// it next, can find it in the last-level cache.
typedef uint32_t hist_u32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ uint4 hist_load(const uint4 *p)
{
    if (STREAM) {
        const hist_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const hist_u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

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

template <bool STREAM>
__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const uint8_t *__restrict__ in,
                                                            uint64_t n,
                                                            unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t bins[256 * HIST_COLS];
    for (uint32_t i = threadIdx.x; i < 256 * HIST_COLS; i += HIST_THREADS)
        bins[i] = 0;
    __syncthreads();
    uint32_t *col = bins + (threadIdx.x & (HIST_COLS - 1));

    // bytes before the first 16-byte boundary, then spans of HIST_VECS x
    // HIST_THREADS vectors, then single vectors and the bytes after the last
    const uint64_t mis = (16 - ((uintptr_t)in & 15)) & 15;
    const uint64_t head = mis < n ? mis : n;
    const uint64_t nvec = (n - head) >> 4;
    const uint64_t tail = head + (nvec << 4);
    constexpr uint64_t SPAN = (uint64_t)HIST_VECS * HIST_THREADS;
    const uint64_t n_spans = nvec / SPAN;
    const uint4 *vin = reinterpret_cast<const uint4 *>(in + head);
    if (blockIdx.x == 0) {
        for (uint64_t i = threadIdx.x; i < head; i += HIST_THREADS)
            __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        for (uint64_t v = n_spans * SPAN + threadIdx.x; v < nvec; v += HIST_THREADS) {
            const uint4 x = vin[v];
            hist_add4(col, x.x);
            hist_add4(col, x.y);
            hist_add4(col, x.z);
            hist_add4(col, x.w);
        }
        for (uint64_t i = tail + threadIdx.x; i < n; i += HIST_THREADS)
            __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    uint64_t sp = blockIdx.x;
    if (sp < n_spans) {
        // x[k] is refilled for the workgroup's next span as soon as it has been
        // counted, so HIST_VECS loads per lane are in flight at any time
        uint4 x[HIST_VECS];
        const uint4 *src = vin + sp * SPAN + threadIdx.x;
#pragma unroll
        for (int k = 0; k < HIST_VECS; ++k) {
            x[k] = hist_load<STREAM>(src + k * HIST_THREADS);
            __builtin_amdgcn_sched_barrier(0); // in the order in which the loop refills them
        }
#pragma nounroll
        for (; sp < n_spans; sp += gridDim.x) {
            // after the last span every refill reads the span's first 16 bytes
            // again (unused): no branch in the loop, exact wait counts
            const bool more = sp + gridDim.x < n_spans;
            const uint32_t step = more ? HIST_THREADS : 0;
            src = vin + (more ? (sp + gridDim.x) * SPAN + threadIdx.x : sp * SPAN);
#pragma unroll
            for (int k = 0; k < HIST_VECS; ++k) {
                hist_add4(col, x[k].x);
                hist_add4(col, x[k].y);
                hist_add4(col, x[k].z);
                hist_add4(col, x[k].w);
                __builtin_amdgcn_sched_barrier(0); // the refill goes into the registers just counted
                x[k] = hist_load<STREAM>(src + k * step);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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
static_assert(PLAN_VECS == HIST_VECS, "a batch is one span of loads");

// Column sums of one batch: lanes 2b and 2b + 1 take 16 columns of bin b each,
// as four 16-byte pieces in an order that keeps 16 neighbouring lanes on 16
// different bank groups; the columns are cleared for the next batch.  The 256
// sums wait in `s_sums` (HIST_PEND batches of them) and are written out
// together, 16 bytes per lane.
constexpr int HIST_PEND = 8;

__device__ __forceinline__ void hist_batch_sum(uint32_t *bins, uint32_t *s_sums, uint32_t slot,
                                               unsigned long long &running)
{
    const uint32_t bin = threadIdx.x >> 1, half = threadIdx.x & 1u;
    uint4 *mine = reinterpret_cast<uint4 *>(bins + bin * HIST_COLS + half * 16);
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
        s_sums[slot * 256 + bin] = s;
        running += s;
    }
    __syncthreads();
}

// Writes the pending sums of batches first, first + stride, ... (count of them).
// STREAM: non-temporal stores, like the loads of such an input -- plain stores
// between the non-temporal loads cost the pass 6 % (1 GiB: 0.186 -> 0.174 ms),
// far more than their 1.6 % of the bytes.
template <bool STREAM>
__device__ __forceinline__ void hist_flush(const uint32_t *s_sums, uint32_t *__restrict__ batch_hist,
                                           uint64_t first, uint64_t stride, uint32_t count)
{
    const uint32_t slot = threadIdx.x >> 6, lane = threadIdx.x & 63u; // one wave per pending batch
    if (slot < count) {
        const uint4 v = reinterpret_cast<const uint4 *>(s_sums + slot * 256)[lane];
        uint4 *dst = reinterpret_cast<uint4 *>(batch_hist + (first + slot * stride) * 256) + lane;
        if (STREAM) {
            const hist_u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<hist_u32x4 *>(dst));
        } else {
            *dst = v;
        }
    }
}

template <bool ALIGNED, bool STREAM>
__global__ __launch_bounds__(HIST_THREADS) void hist_batches_kernel(
    const uint8_t *__restrict__ in, uint64_t n, unsigned long long *__restrict__ hist,
    uint32_t *__restrict__ batch_hist, uint64_t n_batches)
{
    static_assert(HIST_PEND * 64 == HIST_THREADS, "one wave per pending batch");
    __shared__ __align__(16) uint32_t bins[256 * HIST_COLS];
    __shared__ __align__(16) uint32_t s_sums[HIST_PEND * 256];
    for (uint32_t i = threadIdx.x; i < 256 * HIST_COLS; i += HIST_THREADS)
        bins[i] = 0;
    __syncthreads();
    uint32_t *col = bins + (threadIdx.x & (HIST_COLS - 1));
    unsigned long long running = 0;
    uint32_t pend = 0;       // sums waiting in s_sums
    uint64_t pend_first = 0; // the batch of the first of them

    // whole batches of a 16-byte aligned input, 16 bytes per lane and load
    const uint64_t n_whole = ALIGNED ? n / PLAN_BATCH : 0;
    uint64_t b = blockIdx.x;
    if (ALIGNED && b < n_whole) {
        // x[k] is refilled for the workgroup's next batch as soon as it has
        // been counted: eight 16-byte loads per lane stay in flight through the
        // count, both barriers and the column sums
        uint4 x[PLAN_VECS];
        const uint4 *vin = reinterpret_cast<const uint4 *>(in + b * PLAN_BATCH) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < PLAN_VECS; ++k) {
            x[k] = hist_load<STREAM>(vin + k * HIST_THREADS);
            __builtin_amdgcn_sched_barrier(0); // in the order in which the loop refills them
        }
#pragma nounroll
        for (; b < n_whole; b += gridDim.x) {
            // after the workgroup's last whole batch every refill reads that
            // batch's first 16 bytes again (unused): no branch in the loop,
            // exact wait counts
            const bool more = b + gridDim.x < n_whole;
            const uint32_t step = more ? HIST_THREADS : 0;
            vin = reinterpret_cast<const uint4 *>(in + (more ? b + gridDim.x : b) * PLAN_BATCH) +
                  (more ? threadIdx.x : 0u);
#pragma unroll
            for (int k = 0; k < PLAN_VECS; ++k) {
                hist_add4(col, x[k].x);
                hist_add4(col, x[k].y);
                hist_add4(col, x[k].z);
                hist_add4(col, x[k].w);
                __builtin_amdgcn_sched_barrier(0); // the refill goes into the registers just counted
                x[k] = hist_load<STREAM>(vin + k * step);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pend == 0)
                pend_first = b;
            hist_batch_sum(bins, s_sums, pend, running);
            if (++pend == HIST_PEND) {
                hist_flush<STREAM>(s_sums, batch_hist, pend_first, gridDim.x, pend);
                pend = 0; // (the next sums are written behind the next batch's first barrier)
            }
        }
    }
    // the ragged last batch, and every batch of an input that is not aligned
    for (; b < n_batches; b += gridDim.x) {
        const uint64_t base = b * PLAN_BATCH;
        const uint64_t end = base + PLAN_BATCH < n ? base + PLAN_BATCH : n;
        for (uint64_t i = base + threadIdx.x; i < end; i += HIST_THREADS)
            __hip_atomic_fetch_add(col + ((uint32_t)in[i] << 5), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pend == 0)
            pend_first = b;
        hist_batch_sum(bins, s_sums, pend, running);
        if (++pend == HIST_PEND) {
            hist_flush<STREAM>(s_sums, batch_hist, pend_first, gridDim.x, pend);
            pend = 0;
        }
    }
    if (pend)
        hist_flush<STREAM>(s_sums, batch_hist, pend_first, gridDim.x, pend);
    if ((threadIdx.x & 1u) == 0 && running)
        atomicAdd(&hist[threadIdx.x >> 1], running);
}

} // namespace

#endif // HFGPU_KERN_HIST_HPP
