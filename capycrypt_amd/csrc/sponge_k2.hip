// sponge_k2.hip — instances of sponge_kernel_k2<RW, MODE> (see sponge_kernels_k2.h)
#include "sponge_kernels_k2.h"
#include "sponge_launch.h"

namespace capy {

#define CAPY_CASE(RW, MODE) \
    case RW * 2 + MODE: hipLaunchKernelGGL((sponge_kernel_k2<RW, MODE>), grid, block, 0, s, p); break;

hipError_t launch_sponge_k2(int rw, int mode, const SpongeParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.n + 31) / 32)), block(64);
    if ((p.debug_flags & 8) && rw == 17 && mode == 0) {  // A/B: rolled two-round body (profiles/r02_second_issue_slot.txt)
        hipLaunchKernelGGL((sponge_kernel_k2<17, 0, 1>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    if ((p.debug_flags & 512) && rw == 17 && mode == 0) {
        // A/B (debug bit 9): the blocked round with priority, for forced two-lane launches with two waves per SIMD
        hipLaunchKernelGGL((sponge_kernel_k2<17, 0, 2>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    switch (rw * 2 + mode) {
        CAPY_CASE(9, 0)
        CAPY_CASE(13, 0)
        CAPY_CASE(17, 0)
        CAPY_CASE(18, 0)
        CAPY_CASE(19, 0)
        CAPY_CASE(21, 0)
        CAPY_CASE(17, 1)
        CAPY_CASE(19, 1)
        CAPY_CASE(21, 1)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace capy
