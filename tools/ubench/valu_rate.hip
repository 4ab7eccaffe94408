// Micro-benchmark: issue rate of a few VALU / LDS instructions on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * seed;
    __syncthreads();
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t s = (seed & 7) + 1;
    uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) { // 32-bit shift
            asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
        } else if (OP == 1) { // 64-bit shift
            asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
        } else if (OP == 2) { // alignbit
            asm volatile("v_alignbit_b32 %0, %0, %1, %4\n v_alignbit_b32 %1, %1, %2, %4\n v_alignbit_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %0, %4"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
        } else if (OP == 3) { // add3
            asm volatile("v_add3_u32 %0, %0, %1, %4\n v_add3_u32 %1, %1, %2, %4\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %0, %4"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
        } else if (OP == 4) { // ds_read_u16 x4 (random-ish addresses)
            uint32_t x0, x1, x2, x3;
            asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %5\n ds_read_u16 %2, %6\n ds_read_u16 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(b0 & 0x3ffe), "v"(b1 & 0x3ffe), "v"(b2 & 0x3ffe), "v"(b3 & 0x3ffe));
            b0 += x0; b1 += x1; b2 += x2; b3 += x3;
        } else if (OP == 5) { // ds_read_b64 x4
            uint64_t x0, x1, x2, x3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(b0 & 0x3ff8), "v"(b1 & 0x3ff8), "v"(b2 & 0x3ff8), "v"(b3 & 0x3ff8));
            b0 += (uint32_t)x0; b1 += (uint32_t)x1; b2 += (uint32_t)x2; b3 += (uint32_t)x3;
        } else if (OP == 6) { // ds_or_b32 x4 (no return)
            asm volatile("ds_or_b32 %0, %4\n ds_or_b32 %1, %4\n ds_or_b32 %2, %4\n ds_or_b32 %3, %4\n s_waitcnt lgkmcnt(0)"
                         :: "v"(b0 & 0x3ffc), "v"(b1 & 0x3ffc), "v"(b2 & 0x3ffc), "v"(b3 & 0x3ffc), "v"(s));
            b0 += 4 * s; b1 += 12; b2 += 20; b3 += 28;
        } else if (OP == 7) { // lshrrev_b64
            asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
        } else if (OP == 8) { // v_perm
            asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(0x00010203u + s));
        } else if (OP == 10 || OP == 11 || OP == 14) { // ds_read_b64, few distinct addresses
            uint64_t x0, x1, x2, x3;
            const uint32_t base = OP == 10 ? 0u : (OP == 11 ? (threadIdx.x & 3u) * 8u : ((threadIdx.x * 2654435761u >> 28) & 7u) * 32u + (threadIdx.x & 3u) * 8u);
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(base + (b0 & 0)), "v"(base + 256 + (b1 & 0)), "v"(base + 512 + (b2 & 0)), "v"(base + 1024 + (b3 & 0)));
            b0 += (uint32_t)x0; b1 += (uint32_t)x1; b2 += (uint32_t)x2; b3 += (uint32_t)x3;
        } else if (OP == 12) { // ds_read_u8 same address
            uint32_t x0, x1, x2, x3;
            asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %5\n ds_read_u8 %2, %6\n ds_read_u8 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(b0 & 1), "v"(b1 & 1), "v"(b2 & 1), "v"(b3 & 1));
            b0 += x0 + 1; b1 += x1 + 3; b2 += x2 + 5; b3 += x3 + 7;
        } else if (OP == 13) { // ds_or_b32 to a lane-private slot
            asm volatile("ds_or_b32 %0, %1\n ds_or_b32 %0, %1\n ds_or_b32 %0, %1\n ds_or_b32 %0, %1\n s_waitcnt lgkmcnt(0)"
                         :: "v"(threadIdx.x * 4u), "v"(s));
        } else if (OP == 9) { // ds_read_u8 x4
            uint32_t x0, x1, x2, x3;
            asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %5\n ds_read_u8 %2, %6\n ds_read_u8 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(b0 & 0xff), "v"(b1 & 0xff), "v"(b2 & 0xff), "v"(b3 & 0xff));
            b0 += x0 + 1; b1 += x1 + 3; b2 += x2 + 5; b3 += x3 + 7;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 + b1 + b2 + b3 + (uint32_t)(a0 + a1 + a2 + a3);
}
template <int OP>
void run(const char *name, uint32_t *d)
{
    const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * ITER * 4
    const double winst = (double)blocks * 4 / 1024.0 * ITER * 4;
    printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("lshl_b32", d); run<1>("lshl_b64", d); run<7>("lshr_b64", d); run<2>("alignbit", d);
    run<3>("add3", d); run<8>("perm", d); run<4>("ds_read_u16", d); run<9>("ds_read_u8", d);
    run<5>("ds_read_b64", d); run<6>("ds_or_b32", d);
    run<10>("b64 same addr", d); run<11>("b64 4 addrs", d); run<14>("b64 32 addrs", d); run<12>("u8 2 addrs", d); run<13>("ds_or private", d);
    return 0;
}
