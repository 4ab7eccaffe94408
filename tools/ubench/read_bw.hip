// Micro-benchmark: achievable HBM read and copy bandwidth with plain 16-byte loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int UNROLL>
__global__ __launch_bounds__(512) void rd(const uint4 *in, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(512) void cp(const uint4 *in, uint4 *o, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) o[i] = in[i];
}
int main()
{
    const size_t n = (size_t)1 << 30;
    uint4 *d, *o; uint32_t *r;
    hipMalloc(&d, n); hipMalloc(&o, n); hipMalloc(&r, 4);
    hipMemset(d, 1, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            rd<4><<<grid, 512>>>(d, n / 16, r);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read  unroll4 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, n / ms / 1e9);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            rd<8><<<grid, 512>>>(d, n / 16, r);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("read  unroll8 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, n / ms / 1e9);
    }
    for (int grid : {1024, 2048, 4096, 16384}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            cp<<<grid, 512>>>(d, o, n / 16);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy          grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, ms, 2.0 * n / ms / 1e9);
    }
    return 0;
}
