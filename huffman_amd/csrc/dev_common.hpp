// Helpers shared by the kernels: wave and workgroup scans, byte swap, wall clock.
// Part of the single translation unit huffman_amd/csrc/hfgpu.hip (device code
// for gfx950, wave64); included there, not compiled on its own.
#ifndef HFGPU_DEV_COMMON_HPP
#define HFGPU_DEV_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "hfgpu.h"

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

} // namespace

#endif // HFGPU_DEV_COMMON_HPP
