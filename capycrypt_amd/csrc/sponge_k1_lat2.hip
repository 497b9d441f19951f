// sponge_k1_lat2.hip — instances of sponge_kernel<RW, FULLCHIP=false, MODE, 2, PAIRED=true>: the latency-tuned kernel
// for launches that put two waves on a SIMD (keccak_dev.h: keccak_round_blocked)
#include "sponge_kernels.h"
#include "sponge_launch.h"

namespace capy {

#define CAPY_CASE(RW, MODE) \
    case RW * 2 + MODE: hipLaunchKernelGGL((sponge_kernel<RW, false, MODE, 2, true>), grid, block, 0, s, p); break;

hipError_t launch_sponge_k1_lat_paired(int rw, int mode, const SpongeParams &p, hipStream_t s)
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
