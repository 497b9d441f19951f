// sponge_uniform.hip — instances of sponge_uniform_kernel<RW> (see sponge_uniform.h)
#include "sponge_uniform.h"
#include "sponge_launch.h"

namespace capy {

#define CAPY_CASE(RW) \
    case RW: hipLaunchKernelGGL((sponge_uniform_kernel<RW>), grid, block, pad, s, p); break;

// waves: occupancy cap in waves per SIMD (0 = whatever fits: 4).  The kernels need ~100 VGPRs, so the register file
// admits four waves; a lower cap is imposed with unused dynamic LDS (160 KB per CU, 4 SIMDs).
hipError_t launch_sponge_uniform(int rw, const SpongeParams &p, int waves, hipStream_t s, unsigned sliced_grid)
{
    const dim3 grid(sliced_grid ? sliced_grid : (unsigned)((p.n + 63) / 64)), block(64);
    const size_t stat = rw >= 16 ? (size_t)64 * rw * 8 : 0;
    size_t pad = 0;
    if (waves >= 1 && waves <= 3) {
        const size_t need = 163840 / (4 * (size_t)waves + 1) + 64;  // one workgroup too many would not fit
        pad = need > stat ? need - stat : 0;
    }
    if (sliced_grid) {
#define CAPY_SCASE(RW) \
    case RW: hipLaunchKernelGGL((sponge_uniform_kernel<RW, true>), grid, block, pad, s, p); break;
        switch (rw) {
            CAPY_SCASE(9)
            CAPY_SCASE(13)
            CAPY_SCASE(17)
            CAPY_SCASE(18)
            CAPY_SCASE(19)
            CAPY_SCASE(21)
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (rw) {
        CAPY_CASE(9)
        CAPY_CASE(13)
        CAPY_CASE(17)
        CAPY_CASE(18)
        CAPY_CASE(19)
        CAPY_CASE(21)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace capy
