// sponge_fused.hip — instances of sponge_fused_crypt_kernel<RW> (see sponge_fused.h)
#include "sponge_fused.h"
#include "sponge_launch.h"

namespace capy {

hipError_t launch_sponge_fused(int rw, const FusedParams &fp, hipStream_t s)
{
    if (fp.wide) return launch_sponge_il_crypt(rw, fp, s);  // very small batches: two waves per item (sponge_wide_il.h)
    const dim3 grid(fp.sl_groups ? fp.sl_grid : (unsigned)((fp.n + 15) / 16)), block(64);
    if (fp.staged) {
        switch (rw) {
        case 17: hipLaunchKernelGGL((sponge_fused_crypt_kernel<17, true>), grid, block, 0, s, fp); break;
        case 19: hipLaunchKernelGGL((sponge_fused_crypt_kernel<19, true>), grid, block, 0, s, fp); break;
        case 21: hipLaunchKernelGGL((sponge_fused_crypt_kernel<21, true>), grid, block, 0, s, fp); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#define CAPY_FUSED_PAIRED(P)                                                                                             \
    if (fp.paired == P) {                                                                                                \
        switch (rw) {                                                                                                    \
        case 17: hipLaunchKernelGGL((sponge_fused_crypt_kernel<17, false, P>), grid, block, 0, s, fp); break;            \
        case 19: hipLaunchKernelGGL((sponge_fused_crypt_kernel<19, false, P>), grid, block, 0, s, fp); break;            \
        case 21: hipLaunchKernelGGL((sponge_fused_crypt_kernel<21, false, P>), grid, block, 0, s, fp); break;            \
        default: return hipErrorInvalidValue;                                                                            \
        }                                                                                                                \
        return hipGetLastError();                                                                                        \
    }
    CAPY_FUSED_PAIRED(1)  // more than one wave per SIMD
#undef CAPY_FUSED_PAIRED
    switch (rw) {
    case 17: hipLaunchKernelGGL(sponge_fused_crypt_kernel<17>, grid, block, 0, s, fp); break;
    case 19: hipLaunchKernelGGL(sponge_fused_crypt_kernel<19>, grid, block, 0, s, fp); break;
    case 21: hipLaunchKernelGGL(sponge_fused_crypt_kernel<21>, grid, block, 0, s, fp); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace capy
