// sponge_wide_il.hip — instances of the bit-interleaved one-wave-per-sponge kernels (see sponge_wide_il.h)
#include "sponge_wide_il.h"
#include "sponge_launch.h"
#include "sponge_internal.h"

namespace capy {

hipError_t launch_sponge_il_digest(int rw, const SpongeParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)p.n), block(64);
    const bool lone = p.n <= device_simds();
#define CAPY_IL_DIGEST(RW)                                                                 \
    case RW:                                                                               \
        if (lone)                                                                          \
            hipLaunchKernelGGL((sponge_il_digest_kernel<RW, true>), grid, block, 0, s, p);  \
        else                                                                               \
            hipLaunchKernelGGL((sponge_il_digest_kernel<RW, false>), grid, block, 0, s, p); \
        break;
    switch (rw) {
        CAPY_IL_DIGEST(9)
        CAPY_IL_DIGEST(13)
        CAPY_IL_DIGEST(17)
        CAPY_IL_DIGEST(18)
        CAPY_IL_DIGEST(19)
        CAPY_IL_DIGEST(21)
    default: return hipErrorInvalidValue;
    }
#undef CAPY_IL_DIGEST
    return hipGetLastError();
}

hipError_t launch_sponge_il_crypt(int rw, const FusedParams &fp, hipStream_t s)
{
    const dim3 grid((unsigned)fp.n), block(128);
    const bool lone = 2 * fp.n <= device_simds();
#define CAPY_IL_CRYPT1(RW, DEC)                                                                  \
    if (lone)                                                                                    \
        hipLaunchKernelGGL((sponge_il_crypt_kernel<RW, DEC, true>), grid, block, 0, s, fp);       \
    else                                                                                         \
        hipLaunchKernelGGL((sponge_il_crypt_kernel<RW, DEC, false>), grid, block, 0, s, fp);
#define CAPY_IL_CRYPT(RW)              \
    case RW:                           \
        if (fp.decrypt) {              \
            CAPY_IL_CRYPT1(RW, true)   \
        } else {                       \
            CAPY_IL_CRYPT1(RW, false)  \
        }                              \
        break;
    switch (rw) {
        CAPY_IL_CRYPT(17)
        CAPY_IL_CRYPT(19)
        CAPY_IL_CRYPT(21)
    default: return hipErrorInvalidValue;
    }
#undef CAPY_IL_CRYPT
#undef CAPY_IL_CRYPT1
    return hipGetLastError();
}

}  // namespace capy
