// Micro-benchmark: what the memory system delivers to a streaming kernel that keeps
// several 16-byte accesses in flight per lane -- the ceiling that the codec's
// kernels (read N, write C or the reverse) are compared with.
//
// Every work-group of 512 lanes moves tiles of DEPTH x 512 x 16 bytes: DEPTH
// loads per lane are issued back to back, then the DEPTH stores.  Variants:
// plain / non-temporal loads, plain / non-temporal stores, a tile per
// work-group or a persistent grid that walks tiles, and read-only / write-only
// forms.  Buffers are 1 GiB (and 4 GiB for the copy), well past the 256 MiB
// Infinity Cache.  Reported: best and median of 7 launches, bytes read + bytes
// written per second.
//
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4 *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u32x4 *p, u32x4 v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <int DEPTH, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(512) void copy_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out,
                                                   size_t tiles)
{
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = t * (size_t)(DEPTH * 512) + threadIdx.x;
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            v[u] = ld16<NT_LD>(in + base + u * 512);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            st16<NT_ST>(out + base + u * 512, v[u]);
    }
}

template <int DEPTH, bool NT_LD>
__global__ __launch_bounds__(512) void read_kernel(const u32x4 *__restrict__ in, uint32_t *sink, size_t tiles)
{
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = t * (size_t)(DEPTH * 512) + threadIdx.x;
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            v[u] = ld16<NT_LD>(in + base + u * 512);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

template <int DEPTH, bool NT_ST>
__global__ __launch_bounds__(512) void write_kernel(u32x4 *__restrict__ out, size_t tiles)
{
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = t * (size_t)(DEPTH * 512) + threadIdx.x;
        u32x4 v = {(uint32_t)t, threadIdx.x, 1u, 2u};
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            st16<NT_ST>(out + base + u * 512, v);
    }
}

#define CHECK(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                     \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)

template <typename F>
static void timeit(const char *what, double bytes, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    launch();
    launch();
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-62s best %.3f ms %.2f TB/s   median %.3f ms %.2f TB/s\n", what, ms[0], bytes / ms[0] / 1e9,
           ms[3], bytes / ms[3] / 1e9);
    fflush(stdout);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    const size_t big = (size_t)4 << 30;
    u32x4 *a, *b;
    uint32_t *sink;
    CHECK(hipMalloc(&a, big));
    CHECK(hipMalloc(&b, big));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 1, big));
    CHECK(hipMemset(b, 2, big));
    for (size_t n : {(size_t)1 << 30, big}) {
        const double gib = (double)n / (1 << 30);
        char what[128];
        const size_t t8 = n / (8 * 512 * 16), t4 = n / (4 * 512 * 16);
#define RUN(LABEL, BYTES, ...)                                                                 \
    snprintf(what, sizeof what, "%g GiB %s", gib, LABEL);                                      \
    timeit(what, BYTES, [&] { __VA_ARGS__; })
        RUN("read  8x16B in flight, tile per work-group", (double)n,
            (read_kernel<8, false><<<(unsigned)t8, 512>>>(a, sink, t8)));
        RUN("read  8x16B in flight, tile per work-group, nt", (double)n,
            (read_kernel<8, true><<<(unsigned)t8, 512>>>(a, sink, t8)));
        RUN("read  8x16B in flight, 2048 persistent work-groups, nt", (double)n,
            (read_kernel<8, true><<<2048, 512>>>(a, sink, t8)));
        RUN("write 8x16B, tile per work-group", (double)n, (write_kernel<8, false><<<(unsigned)t8, 512>>>(b, t8)));
        RUN("write 8x16B, tile per work-group, nt", (double)n,
            (write_kernel<8, true><<<(unsigned)t8, 512>>>(b, t8)));
        RUN("copy  4x16B in flight, tile per work-group", 2.0 * n,
            (copy_kernel<4, false, false><<<(unsigned)t4, 512>>>(a, b, t4)));
        RUN("copy  8x16B in flight, tile per work-group", 2.0 * n,
            (copy_kernel<8, false, false><<<(unsigned)t8, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, tile per work-group, nt loads", 2.0 * n,
            (copy_kernel<8, true, false><<<(unsigned)t8, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, tile per work-group, nt stores", 2.0 * n,
            (copy_kernel<8, false, true><<<(unsigned)t8, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, tile per work-group, nt loads+stores", 2.0 * n,
            (copy_kernel<8, true, true><<<(unsigned)t8, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, 1024 persistent work-groups", 2.0 * n,
            (copy_kernel<8, false, false><<<1024, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, 2048 persistent work-groups", 2.0 * n,
            (copy_kernel<8, false, false><<<2048, 512>>>(a, b, t8)));
        RUN("copy  8x16B in flight, 2048 persistent work-groups, nt both", 2.0 * n,
            (copy_kernel<8, true, true><<<2048, 512>>>(a, b, t8)));
        // the codec's shape: read N, write 0.725 N (encode) -- approximated by a copy
        // of 0.725 N beside a read of the remaining 0.275 N
        {
            const size_t tc = (size_t)(t8 * 0.725), tr = t8 - tc;
            RUN("read N + write 0.725 N (copy part + read part, two launches)", (double)n * 1.725,
                (copy_kernel<8, false, false><<<(unsigned)tc, 512>>>(a, b, tc));
                (read_kernel<8, false><<<(unsigned)tr, 512>>>(a + tc * 4096, sink, tr)));
        }
    }
    return 0;
}
