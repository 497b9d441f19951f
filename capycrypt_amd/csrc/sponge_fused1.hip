// sponge_fused1.hip — instances of sponge_fused1_kernel<RW, FORM, DECRYPT> (see sponge_fused1.h)
#include "sponge_fused1.h"
#include "sponge_launch.h"

namespace capy {

// fp.one_lane = FORM (1, 2, 4); fp.cap_waves = 1..3: occupancy cap in waves per SIMD for the FORM-4 instance (unused dynamic
// LDS: one workgroup too many would not fit; 160 KB per compute unit, 4 SIMDs), 0: whatever fits
hipError_t launch_sponge_fused1(int rw, const FusedParams &fp, hipStream_t s)
{
    const dim3 grid(fp.sl_groups ? fp.sl_grid : (unsigned)((fp.n + FUSED1_ITEMS - 1) / FUSED1_ITEMS)), block(64);
    size_t pad = 0;
    if (fp.one_lane == 4 && fp.cap_waves >= 1 && fp.cap_waves <= 3) {
        const size_t need = 163840 / (4 * (size_t)fp.cap_waves + 1) + 64;
        pad = need > FUSED1_LDS_WAVE ? need - FUSED1_LDS_WAVE : 0;
    }
#define CAPY_FUSED1_RW(F, D)                                                                              \
    switch (rw) {                                                                                         \
    case 17: hipLaunchKernelGGL((sponge_fused1_kernel<17, F, D>), grid, block, pad, s, fp); break;        \
    case 19: hipLaunchKernelGGL((sponge_fused1_kernel<19, F, D>), grid, block, pad, s, fp); break;        \
    case 21: hipLaunchKernelGGL((sponge_fused1_kernel<21, F, D>), grid, block, pad, s, fp); break;        \
    default: return hipErrorInvalidValue;                                                                 \
    }                                                                                                     \
    return hipGetLastError();
#define CAPY_FUSED1(F)                          \
    if (fp.one_lane == F) {                     \
        if (fp.decrypt) {                       \
            CAPY_FUSED1_RW(F, true)             \
        } else {                                \
            CAPY_FUSED1_RW(F, false)            \
        }                                       \
    }
    CAPY_FUSED1(1)
    CAPY_FUSED1(2)
    CAPY_FUSED1(4)
#undef CAPY_FUSED1
#undef CAPY_FUSED1_RW
    return hipErrorInvalidValue;
}

// one phase of the rotating-occupancy schedule: `cus` workgroups of 512 lanes
hipError_t launch_sponge_fused1_rot(int rw, const FusedParams &fp, unsigned cus, hipStream_t s)
{
    const dim3 grid(cus), block(512);
#define CAPY_F1ROT(D)                                                                                  \
    switch (rw) {                                                                                      \
    case 17: hipLaunchKernelGGL((sponge_fused1_rot_kernel<17, D>), grid, block, 0, s, fp); break;      \
    case 19: hipLaunchKernelGGL((sponge_fused1_rot_kernel<19, D>), grid, block, 0, s, fp); break;      \
    case 21: hipLaunchKernelGGL((sponge_fused1_rot_kernel<21, D>), grid, block, 0, s, fp); break;      \
    default: return hipErrorInvalidValue;                                                              \
    }                                                                                                  \
    return hipGetLastError();
    if (fp.decrypt) {
        CAPY_F1ROT(true)
    } else {
        CAPY_F1ROT(false)
    }
#undef CAPY_F1ROT
}

}  // namespace capy
