// sponge_rot.hip — instances of sponge_rot_kernel<RW> (see sponge_rot.h)
#include "sponge_rot.h"
#include "sponge_launch.h"

namespace capy {

#define CAPY_CASE(RW) \
    case RW: hipLaunchKernelGGL((sponge_rot_kernel<RW>), grid, block, lds, s, q); break;

// cus workgroups of 512 lanes; lds_bytes of (unused) dynamic LDS make a compute unit hold exactly one of them
hipError_t launch_sponge_rot(int rw, const RotParams &q, unsigned cus, size_t lds_bytes, hipStream_t s)
{
    const dim3 grid(cus), block(512);
    const size_t lds = lds_bytes;
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
