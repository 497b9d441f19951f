// sponge_k1_full.hip — instances of sponge_kernel<RW, FULLCHIP=true, MODE> (see sponge_kernels.h)
#include "sponge_kernels.h"
#include "sponge_launch.h"

namespace capy {

#define CAPY_CASE(RW, MODE) \
    case RW * 2 + MODE: hipLaunchKernelGGL((sponge_kernel<RW, true, MODE>), grid, block, 0, s, p); break;

hipError_t launch_sponge_k1_full(int rw, int mode, const SpongeParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.n + 63) / 64)), block(64);
    switch (rw * 2 + mode) {
        CAPY_CASE(9, 0)
        CAPY_CASE(13, 0)
        CAPY_CASE(17, 0)
        CAPY_CASE(18, 0)
        CAPY_CASE(19, 0)
        CAPY_CASE(21, 0)
        CAPY_CASE(17, 1)  // keystream XOR exists only for cSHAKE/KMAC rates
        CAPY_CASE(19, 1)
        CAPY_CASE(21, 1)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace capy
