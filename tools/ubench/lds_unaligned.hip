// Micro-test: 32-bit LDS stores at byte addresses that are not multiples of four.
// (1) Correctness: every lane stores a known dword at row + k for k = 0..7 into
//     a zeroed row of its own and the rows are compared with what byte-wise
//     stores would have left.
// (2) Cost: cycles per wave-instruction for ds_write_b32 at aligned addresses, at
//     addresses of every alignment (row + lane-dependent offset), and for the
//     same with ds_read_b64 table reads between them -- the access mix of a
//     decoder that appends up to four symbols per table look-up.
//
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int ROW = 84; // bytes per lane (21 words: consecutive lanes start on different banks)

__global__ __launch_bounds__(256) void store_check(uint8_t *out, uint32_t k)
{
    __shared__ __align__(16) uint8_t rows[256 * ROW];
    for (uint32_t i = threadIdx.x; i < 256 * ROW / 4; i += 256)
        reinterpret_cast<uint32_t *>(rows)[i] = 0;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(rows + threadIdx.x * ROW + k + 4 * (threadIdx.x & 7));
    const uint32_t val = 0xA1B2C3D4u ^ (threadIdx.x * 0x01010101u);
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(val) : "memory");
    // a second store that overlaps the first by cnt bytes, as the decoder's appends do
    const uint32_t val2 = ~val;
    asm volatile("ds_write_b32 %0, %1 offset:3\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(val2) : "memory");
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256 * ROW; i += 256)
        out[i] = rows[i];
}

// mode 0: aligned stores; 1: stores at row + (it + lane) % 4 (all alignments);
// 2: as 1 with a ds_read_b64 from a 16 KiB table in front of each store
__global__ __launch_bounds__(256) void store_rate(uint64_t *cycles, uint32_t *sink, int mode, int iters)
{
    __shared__ __align__(16) uint8_t rows[256 * ROW];
    __shared__ __align__(16) uint64_t table[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += 256)
        table[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)(rows + threadIdx.x * ROW);
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint32_t a = base + 4u * u;
            if (mode >= 1)
                a += (x >> (2 * u)) & 3u;
            uint32_t v = x;
            if (mode == 2) {
                uint64_t e;
                const uint32_t ta = tbase + ((x >> 7) & 0x3ff8u);
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(ta) : "memory");
                v = (uint32_t)e;
                x = x * 1664525u + (uint32_t)(e >> 32);
            }
            asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
        }
        x = x * 1664525u + 1013904223u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    acc = reinterpret_cast<uint32_t *>(rows)[threadIdx.x];
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u)
        sink[0] = x;
}

#define CHECK(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                     \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)

int main()
{
    uint8_t *d;
    CHECK(hipMalloc(&d, 256 * ROW));
    std::vector<uint8_t> got(256 * ROW), want(256 * ROW);
    int failures = 0;
    for (uint32_t k = 0; k < 8; ++k) {
        store_check<<<1, 256>>>(d, k);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), d, got.size(), hipMemcpyDeviceToHost));
        std::fill(want.begin(), want.end(), 0);
        for (uint32_t t = 0; t < 256; ++t) {
            const uint32_t val = 0xA1B2C3D4u ^ (t * 0x01010101u), val2 = ~val;
            uint8_t *row = want.data() + t * ROW + k + 4 * (t & 7);
            memcpy(row, &val, 4);
            memcpy(row + 3, &val2, 4);
        }
        const bool ok = got == want;
        failures += !ok;
        printf("ds_write_b32 at row + %u (+ 4 * (lane %% 8)): %s\n", k, ok ? "bytes as expected" : "WRONG");
        if (!ok) {
            for (uint32_t t = 0; t < 2; ++t) {
                printf("  lane %u got :", t);
                for (int i = 0; i < 16; ++i)
                    printf(" %02x", got[t * ROW + i]);
                printf("\n  lane %u want:", t);
                for (int i = 0; i < 16; ++i)
                    printf(" %02x", want[t * ROW + i]);
                printf("\n");
            }
        }
    }
    uint64_t *cyc;
    uint32_t *sink;
    const int groups = 1024, iters = 2000;
    CHECK(hipMalloc(&cyc, groups * 8));
    CHECK(hipMalloc(&sink, 4));
    const char *names[] = {"aligned ds_write_b32", "ds_write_b32 at any byte alignment",
                           "ds_read_b64 (random, 16 KiB table) + unaligned ds_write_b32"};
    for (int mode = 0; mode < 3; ++mode) {
        store_rate<<<groups, 256>>>(cyc, sink, mode, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h(groups);
        CHECK(hipMemcpy(h.data(), cyc, groups * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto c : h)
            sum += (double)c;
        // (s_memtime counts shader cycles; the figure is the time a wave sees per store with
        // all its CU's resident waves doing the same)
        printf("%-62s %.2f cycles per store as one wave sees it (4-wave work-groups, at most %d resident per CU)\n",
               names[mode], sum / groups / (16.0 * iters), 160 * 1024 / (256 * ROW + 16384));
    }
    return failures ? 1 : 0;
}
