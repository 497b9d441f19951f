// sponge_mixed.hip — instances of sponge_mixed_kernel<RW> and its LDS-staged A/B twin (see sponge_mixed.h)
#include "sponge_mixed.h"
#include "sponge_launch.h"

namespace capy {

hipError_t launch_sponge_mixed(int rw, const MixedParams &q, unsigned waves, hipStream_t s)
{
    const dim3 grid(waves), block(64);
    if (q.staged) {
        switch (rw) {
        case 9: hipLaunchKernelGGL(sponge_mixed_staged_kernel<9>, grid, block, 0, s, q); break;
        case 13: hipLaunchKernelGGL(sponge_mixed_staged_kernel<13>, grid, block, 0, s, q); break;
        case 17: hipLaunchKernelGGL(sponge_mixed_staged_kernel<17>, grid, block, 0, s, q); break;
        case 18: hipLaunchKernelGGL(sponge_mixed_staged_kernel<18>, grid, block, 0, s, q); break;
        case 19: hipLaunchKernelGGL(sponge_mixed_staged_kernel<19>, grid, block, 0, s, q); break;
        case 21: hipLaunchKernelGGL(sponge_mixed_staged_kernel<21>, grid, block, 0, s, q); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (rw) {
    case 9: hipLaunchKernelGGL(sponge_mixed_kernel<9>, grid, block, 0, s, q); break;
    case 13: hipLaunchKernelGGL(sponge_mixed_kernel<13>, grid, block, 0, s, q); break;
    case 17: hipLaunchKernelGGL(sponge_mixed_kernel<17>, grid, block, 0, s, q); break;
    case 18: hipLaunchKernelGGL(sponge_mixed_kernel<18>, grid, block, 0, s, q); break;
    case 19: hipLaunchKernelGGL(sponge_mixed_kernel<19>, grid, block, 0, s, q); break;
    case 21: hipLaunchKernelGGL(sponge_mixed_kernel<21>, grid, block, 0, s, q); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace capy
