// ed448_vb2.hip — vb2_kernel, the variable-base kernel of chip-filling batches (two items per lane, one shared inversion:
// BASELINE config 4, the Ed448 half of bench.py), in a translation unit of its own because it is the ONE kernel that gains
// from pinned multiply-add chains (CAPY_ED448_ASM_MAD, ed448_dev.h: +1.1 % at 2^18 items) -- every other kernel loses with
// them (one item per lane at 65 536 items -2.7 %, verify's double multiplication -4.5 %, the constant-address fixed base
// -7 %, the constant-address variable base 2.1x: profiles/r04_ed448_pinned_chains.txt), so ed448.hip is compiled without.
// Device code is linked per translation unit, so the two forms of fe_mul / fe_sqr never meet; the host forms are identical.
#ifndef CAPY_ED448_ASM_MAD
#define CAPY_ED448_ASM_MAD 1
#endif
// ... and the one kernel that gains from raised priority around its 4-cycle instructions (two waves per SIMD; needs the
// pinned chains; ed448_dev.h: CAPY_ED448_SETPRIO): 9.63 -> 9.27 ms with form 1, -5 % with form 2 (profiles/r04_ed448_setprio.txt);
// the one-wave-per-SIMD kernels of ed448.hip lose 1-3 % with it.
#ifndef CAPY_ED448_SETPRIO
#define CAPY_ED448_SETPRIO 2
#endif
#include "common.h"
#include "ed448_algo.h"

namespace capy {

#ifndef CAPY_ED448_WAVES
#define CAPY_ED448_WAVES 2
#endif

__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void vb2_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                 const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
                                                 uint32_t *table_ws)
{
    __shared__ PtXYZ parked;
    const uint64_t base = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    if (base >= n) return;
    Pt r = pt_identity();
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
        // the second item of a ragged last wave repeats the first (its result is not written)
        const uint64_t i = (j == 1 && base + 64 < n) ? base + 64 : base;
        const Pt P = pt_from_affine_bytes(points_xy + i * point_stride);
        r = vb_scalarmul(scalars_be + i * scalar_stride, P, table_ws + i * VB_TABLE_DWORDS);
        if (j == 0) park_xyz(parked, r);
    }
    const Pt r0 = unpark_xyz(parked);
    if (base + 64 < n) {
        pt_pair_to_affine_bytes(out_xy + base * 112, out_xy + (base + 64) * 112, r0, r);
    } else {
        pt_to_affine_bytes(out_xy + base * 112, r0);
    }
}

int vb2_launch(size_t n, const uint8_t *scalars, uint64_t scalar_stride, const uint8_t *points, uint64_t point_stride, uint8_t *out,
               uint32_t *table_ws, hipStream_t s)
{
    hipLaunchKernelGGL(vb2_kernel, dim3((unsigned)((n + 127) / 128)), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points,
                       point_stride, out, table_ws);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

}  // namespace capy
