// sponge_wide_il.hip — instances of the bit-interleaved one-wave-per-sponge kernels (see sponge_wide_il.h)
#include "sponge_wide_il.h"
#include "sponge_launch.h"

namespace capy {

hipError_t launch_sponge_il_digest(int rw, const SpongeParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)p.n), block(64);
    switch (rw) {
    case 9: hipLaunchKernelGGL(sponge_il_digest_kernel<9>, grid, block, 0, s, p); break;
    case 13: hipLaunchKernelGGL(sponge_il_digest_kernel<13>, grid, block, 0, s, p); break;
    case 17: hipLaunchKernelGGL(sponge_il_digest_kernel<17>, grid, block, 0, s, p); break;
    case 18: hipLaunchKernelGGL(sponge_il_digest_kernel<18>, grid, block, 0, s, p); break;
    case 19: hipLaunchKernelGGL(sponge_il_digest_kernel<19>, grid, block, 0, s, p); break;
    case 21: hipLaunchKernelGGL(sponge_il_digest_kernel<21>, grid, block, 0, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sponge_il_crypt(int rw, const FusedParams &fp, hipStream_t s)
{
    const dim3 grid((unsigned)fp.n), block(128);
#define CAPY_IL_CRYPT(RW)                                                                       \
    case RW:                                                                                    \
        if (fp.decrypt)                                                                         \
            hipLaunchKernelGGL((sponge_il_crypt_kernel<RW, true>), grid, block, 0, s, fp);      \
        else                                                                                    \
            hipLaunchKernelGGL((sponge_il_crypt_kernel<RW, false>), grid, block, 0, s, fp);     \
        break;
    switch (rw) {
        CAPY_IL_CRYPT(17)
        CAPY_IL_CRYPT(19)
        CAPY_IL_CRYPT(21)
    default: return hipErrorInvalidValue;
    }
#undef CAPY_IL_CRYPT
    return hipGetLastError();
}

}  // namespace capy
