// probe_blocks.hip — bare keccak-f[1600] loops (no loads, no LDS) at 1, 2 and 4 waves per SIMD, launched as 64-thread
// or 256-thread workgroups, for the one-lane and two-lane forms, unrolled and rolled.  Question (VERDICT r1 item 2,
// profiles/r02_second_issue_slot.txt): what stops a second wave per SIMD from adding throughput in the sponge kernels
// when single-instruction loops (tools/microbench*.hip) gain 1.3-2x?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I capycrypt_amd/csrc -o tools/probe_blocks tools/probe_blocks.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "sponge_kernels_k2.h"
using namespace capy;

template <int FORM, int BS>
__global__ __launch_bounds__(BS) void probe(uint32_t iters, uint64_t *sink)
{
    const uint64_t id = (uint64_t)blockIdx.x * BS + threadIdx.x;
    uint32_t x = 0;
    if constexpr (FORM < 2) {
        KState a;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            a.lo[i] = (uint32_t)(id * 25 + i);
            a.hi[i] = (uint32_t)((id * 25 + i) * 0x9E3779B9u);
        }
        for (uint32_t it = 0; it < iters; it++) {
            if constexpr (FORM == 0)
                keccakf1600_unrolled(a);
            else
                keccakf1600_pipelined(a);
        }
#pragma unroll
        for (int i = 0; i < 25; i++) x ^= a.lo[i] ^ a.hi[i];
    } else {
        KHalf a;
        const uint32_t hmask = 0u - (threadIdx.x & 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a.a[i] = (uint32_t)((id * 25 + i) * 0x9E3779B9u);
        for (uint32_t it = 0; it < iters; it++) {
            if constexpr (FORM == 2)
                keccakf1600_k2_unrolled(a, hmask);
            else
                keccakf1600_k2_pipelined(a, hmask);
        }
#pragma unroll
        for (int i = 0; i < 25; i++) x ^= a.a[i];
    }
    if (x == 0x12345678u) atomicXor((unsigned long long *)sink, (unsigned long long)x);
}

template <int FORM, int BS>
static void run(const char *name, uint64_t *sink)
{
    const uint32_t iters = 2000;
    printf("%-22s block=%3d:", name, BS);
    for (int W : {1, 2, 4, 8}) {
        const unsigned waves = 1024u * W;
        const dim3 grid(waves * 64 / BS), block(BS);
        hipLaunchKernelGGL((probe<FORM, BS>), grid, block, 0, 0, 50u, sink);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<FORM, BS>), grid, block, 0, 0, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double sponges = (double)waves * (FORM < 2 ? 64 : 32);
        printf("  W=%d %7.2f ms %6.2f Gperm/s", W, ms, sponges * iters / (ms * 1e-3) / 1e9);
    }
    printf("\n");
}

int main()
{
    uint64_t *sink;
    (void)hipMalloc(&sink, 8);
    (void)hipMemset(sink, 0, 8);
    run<0, 64>("k1 unrolled", sink);
    run<0, 256>("k1 unrolled", sink);
    run<1, 64>("k1 rolled+prefetch", sink);
    run<1, 256>("k1 rolled+prefetch", sink);
    run<2, 64>("k2 unrolled", sink);
    run<2, 256>("k2 unrolled", sink);
    run<3, 64>("k2 rolled+prefetch", sink);
    run<3, 256>("k2 rolled+prefetch", sink);
    return 0;
}
