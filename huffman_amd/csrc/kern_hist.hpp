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

} // namespace

#endif // HFGPU_KERN_HIST_HPP
