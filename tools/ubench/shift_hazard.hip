// Micro-test: does a 64-bit shift mis-read a 32-bit shift amount that sits in the
// last VGPR of the wave's allocation?
//
//     v_lshlrev_b64 v[16:17], v63, v[14:15]      (kernel allocates exactly 64 VGPRs)
//
// Three kernels contain nothing but that pattern: the amount in v63 with 64
// registers allocated (the suspected case), the same instruction with 72
// allocated (v63 is then inside the allocation), and the amount in v71 with 72
// allocated (the last register of a larger allocation).  Each is launched with
// enough work-groups that several waves share every SIMD, and with long enough
// a body (a dependent chain of the same shift) that they overlap in time.
// Every lane checks its own result; the program prints, per kernel, how many
// lanes and how many work-groups saw a wrong value and what the first wrong
// value looked like (the amount the hardware must have used).
//
// build: hipcc --offload-arch=gfx950 -O2 -o shift_hazard shift_hazard.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

struct Report {
    unsigned long long wrong_lanes;
    unsigned long long wrong_groups;
    unsigned long long first_got;
    unsigned long long first_want;
    unsigned int first_block;
    unsigned int first_lane;
};

__device__ void note(Report *r, uint64_t got, uint64_t want)
{
    if (got == want)
        return;
    if (atomicAdd(&r->wrong_lanes, 1ull) == 0) {
        r->first_got = got;
        r->first_want = want;
        r->first_block = blockIdx.x;
        r->first_lane = threadIdx.x;
    }
}

// AMOUNT_REG holds the amount; LAST_REG is clobbered so that the allocation
// reaches it.  `iters` dependent shifts by `amt` and back keep the wave busy.
#define SHIFT_KERNEL(NAME, AMOUNT_REG, LAST_REG)                                               \
    __global__ __launch_bounds__(256) void NAME(Report *rep, uint32_t amt, uint32_t iters,     \
                                                uint32_t *group_flags)                        \
    {                                                                                          \
        const uint64_t x = 0x0000000100000001ull * (threadIdx.x + 3u) + blockIdx.x;            \
        uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), olo, ohi;                         \
        uint32_t bad = 0;                                                                      \
        for (uint32_t it = 0; it < iters; ++it) {                                              \
            asm volatile("v_mov_b32 v14, %2\n\t"                                               \
                         "v_mov_b32 v15, %3\n\t"                                               \
                         "v_mov_b32 " AMOUNT_REG ", %4\n\t"                                    \
                         "s_nop 4\n\t"                                                         \
                         "v_lshlrev_b64 v[16:17], " AMOUNT_REG ", v[14:15]\n\t"                \
                         "s_nop 4\n\t"                                                         \
                         "v_mov_b32 %0, v16\n\t"                                               \
                         "v_mov_b32 %1, v17\n\t"                                               \
                         : "=v"(olo), "=v"(ohi)                                                \
                         : "v"(lo), "v"(hi), "v"(amt)                                          \
                         : "v14", "v15", "v16", "v17", AMOUNT_REG, LAST_REG);                  \
            const uint64_t got = ((uint64_t)ohi << 32) | olo;                                  \
            const uint64_t want = (((uint64_t)hi << 32) | lo) << (amt & 63u);                  \
            if (got != want && !bad) {                                                         \
                bad = 1;                                                                       \
                note(rep, got, want);                                                          \
            }                                                                                  \
            lo = lo * 1664525u + 1013904223u;                                                  \
            hi = hi * 22695477u + 1u;                                                          \
        }                                                                                      \
        if (bad)                                                                               \
            group_flags[blockIdx.x] = 1;                                                       \
    }

// The same for other instructions with a 64-bit result and a 32-bit VGPR source:
// OP is the instruction with v63 as that source (64 VGPRs allocated, v63 the last).
#define OTHER_KERNEL(NAME, ASM_TEXT, WANT_EXPR)                                                \
    __global__ __launch_bounds__(256) void NAME(Report *rep, uint32_t amt, uint32_t iters,     \
                                                uint32_t *group_flags)                        \
    {                                                                                          \
        const uint64_t x = 0x0000000100000001ull * (threadIdx.x + 3u) + blockIdx.x;            \
        uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), olo, ohi;                         \
        uint32_t bad = 0;                                                                      \
        for (uint32_t it = 0; it < iters; ++it) {                                              \
            asm volatile("v_mov_b32 v14, %2\n\t"                                               \
                         "v_mov_b32 v15, %3\n\t"                                               \
                         "v_mov_b32 v63, %4\n\t"                                               \
                         "s_nop 4\n\t" ASM_TEXT "\n\t"                                        \
                         "s_nop 4\n\t"                                                         \
                         "v_mov_b32 %0, v16\n\t"                                               \
                         "v_mov_b32 %1, v17\n\t"                                               \
                         : "=v"(olo), "=v"(ohi)                                                \
                         : "v"(lo), "v"(hi), "v"(amt)                                          \
                         : "v14", "v15", "v16", "v17", "v63");                                 \
            const uint64_t got = ((uint64_t)ohi << 32) | olo;                                  \
            const uint64_t v = ((uint64_t)hi << 32) | lo;                                      \
            const uint64_t want = (WANT_EXPR);                                                 \
            if (got != want && !bad) {                                                         \
                bad = 1;                                                                       \
                note(rep, got, want);                                                          \
            }                                                                                  \
            lo = lo * 1664525u + 1013904223u;                                                  \
            hi = hi * 22695477u + 1u;                                                          \
        }                                                                                      \
        if (bad)                                                                               \
            group_flags[blockIdx.x] = 1;                                                       \
    }

OTHER_KERNEL(lshrrev_b64_v63, "v_lshrrev_b64 v[16:17], v63, v[14:15]", v >> (amt & 63u))
OTHER_KERNEL(ashrrev_i64_v63, "v_ashrrev_i64 v[16:17], v63, v[14:15]", (uint64_t)((int64_t)v >> (amt & 63u)))
OTHER_KERNEL(mad_u64_u32_src0_v63, "v_mad_u64_u32 v[16:17], vcc, v63, v14, v[14:15]", (uint64_t)amt * lo + v)
OTHER_KERNEL(mad_u64_u32_src1_v63, "v_mad_u64_u32 v[16:17], vcc, v14, v63, v[14:15]", (uint64_t)amt * lo + v)
OTHER_KERNEL(lshl_add_u64_v63, "v_lshl_add_u64 v[16:17], v[14:15], v63, v[14:15]", (v << (amt & 7u)) + v)
OTHER_KERNEL(cvt_f64_u32_v63, "v_cvt_f64_u32 v[16:17], v63", (uint64_t)__double_as_longlong((double)amt))
OTHER_KERNEL(lshlrev_b32_v63, "v_lshlrev_b32 v16, v63, v14\n\tv_mov_b32 v17, v15", ((uint64_t)hi << 32) | (uint32_t)(lo << (amt & 31u)))

SHIFT_KERNEL(amount_v63_alloc64, "v63", "v63")
SHIFT_KERNEL(amount_v63_alloc72, "v63", "v64")
SHIFT_KERNEL(amount_v71_alloc72, "v71", "v71")
SHIFT_KERNEL(amount_v62_alloc64, "v62", "v63")

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
    const int groups = 8192; // 32 per CU: eight resident work-groups per CU, four rounds
    Report *rep;
    uint32_t *flags;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMalloc(&flags, groups * sizeof(uint32_t)));
    using Kern = void (*)(Report *, uint32_t, uint32_t, uint32_t *);
    struct {
        const char *name;
        Kern k;
    } kernels[] = {{"amount in v63, 64 VGPRs allocated", amount_v63_alloc64},
                   {"amount in v63, 72 VGPRs allocated", amount_v63_alloc72},
                   {"amount in v71, 72 VGPRs allocated", amount_v71_alloc72},
                   {"amount in v62, 64 VGPRs allocated", amount_v62_alloc64},
                   {"v_lshrrev_b64, amount in v63 of 64", lshrrev_b64_v63},
                   {"v_ashrrev_i64, amount in v63 of 64", ashrrev_i64_v63},
                   {"v_mad_u64_u32, src0 in v63 of 64", mad_u64_u32_src0_v63},
                   {"v_mad_u64_u32, src1 in v63 of 64", mad_u64_u32_src1_v63},
                   {"v_lshl_add_u64, amount in v63 of 64", lshl_add_u64_v63},
                   {"v_cvt_f64_u32, source in v63 of 64", cvt_f64_u32_v63},
                   {"v_lshlrev_b32 (32-bit), amount in v63", lshlrev_b32_v63}};
    for (auto &kk : kernels) {
        hipFuncAttributes attr;
        CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kk.k)));
        for (uint32_t amt : {5u, 2u}) {
            CHECK(hipMemset(rep, 0, sizeof(Report)));
            CHECK(hipMemset(flags, 0, groups * sizeof(uint32_t)));
            kk.k<<<groups, 256>>>(rep, amt, 2000, flags);
            CHECK(hipDeviceSynchronize());
            Report h;
            std::vector<uint32_t> hf(groups);
            CHECK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hf.data(), flags, groups * sizeof(uint32_t), hipMemcpyDeviceToHost));
            unsigned wg = 0, first_wg = groups;
            for (int i = 0; i < groups; ++i)
                if (hf[i]) {
                    ++wg;
                    if (first_wg == (unsigned)groups)
                        first_wg = i;
                }
            printf("%-36s (numRegs %d) amount %2u: wrong lanes %llu, wrong work-groups %u of %d", kk.name,
                   attr.numRegs, amt, h.wrong_lanes, wg, groups);
            if (h.wrong_lanes)
                printf(", lowest wrong group %u; first seen: block %u lane %u got %016llx want %016llx",
                       first_wg, h.first_block, h.first_lane, h.first_got, h.first_want);
            printf("\n");
        }
    }
    return 0;
}
